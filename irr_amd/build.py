"""Builds irr_amd/lib/libirr_hip.so from irr_amd/csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU).

    python -m irr_amd.build [--force]

One object per source, keyed on a CONTENT hash of (source, every header, compiler flags) written next to the object -- never on
modification times, which do not survive the copy to the GPU box -- linked into ONE C-ABI shared library whose exported symbols
are exactly the functions declared in include/irr_hip.h.  ``source.hash`` next to the library is written only after a link that
contains exactly the objects of that hash.

Ablation / trace builds (the IRR_*_ABL, IRR_X3S_TRACE ... macro switches below) never overwrite the product library: they
require ``IRR_BUILD_TAG=<name>`` and go to irr_amd/lib_<name>/; load one with ``IRR_HIP_LIB=<path to that .so>``.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
TAG = os.environ.get("IRR_BUILD_TAG", "")
LIBDIR = os.path.join(PKG, "lib_" + TAG if TAG else "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libirr_hip.so")
ARCH = "gfx950"

# -fno-slp-vectorize -fno-vectorize for EVERY file (round 4, profiles/NOTES.md C.3): the vectorisers turn adjacent scalar fp32 operations
# into packed ones (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), and the one kernel in which we could pin round 4's lane deviation down
# -- conv_smallco_dgrad4_kernel beside the dilation-16 weight gradient on the second stream: 30-39 of 40 launches off by 4e-3 -- is
# bit-stable (0 of 40) when built without them (tools/pair_probe.py).  The instruction at fault: a v_pk_fma_f32 whose accumulator halves
# are swapped (op_sel:[0,0,1] op_sel_hi:[0,1,0] -- what SLP makes of a lane swap); beside MFMA waves of another kernel its low result
# is computed with a zero accumulator in lanes 48..63 (stand-alone: tools/pkfma_swap.py).  Nothing hand-written needs that form, the
# vectorisers cannot be told to avoid it, hence the flags for everything; tools/scan_pk_swap.py checks the machine code.  (Beside MFMAs of the SAME wave a
# packed fp32 instruction also costs far more than its issue slot: x3_split.h.)
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
          "-Wno-unused-function", "-fno-slp-vectorize", "-fno-vectorize"]
_BASE_FLAGS = len(COMMON)  # everything appended below is an ablation / trace macro switch
# per-file extras.  warp.hip reproduces ATen's fp32 rounding sequence: no contraction there.
if os.environ.get("IRR_WG_ABL"):
    COMMON = COMMON + ["-DWG_ABL=" + os.environ["IRR_WG_ABL"]]
if os.environ.get("IRR_WG_ISSUE"):
    COMMON = COMMON + ["-DWG_ISSUE_MODE=" + os.environ["IRR_WG_ISSUE"]]
if os.environ.get("IRR_CONV_D"):
    COMMON = COMMON + ["-DCONV_PREFETCH_D=" + os.environ["IRR_CONV_D"]]
if os.environ.get("IRR_CONV_ABL"):
    COMMON = COMMON + ["-DCONV_ABL=" + os.environ["IRR_CONV_ABL"]]
if os.environ.get("IRR_CONV_ORDER"):
    COMMON = COMMON + ["-DCONV_ORDER=" + os.environ["IRR_CONV_ORDER"]]
if os.environ.get("IRR_WX3_ABL"):
    COMMON = COMMON + ["-DWX3_ABL=" + os.environ["IRR_WX3_ABL"]]
if os.environ.get("IRR_X3_SPLIT_SCALAR"):
    COMMON = COMMON + ["-DX3_SPLIT_SCALAR=" + os.environ["IRR_X3_SPLIT_SCALAR"]]
if os.environ.get("IRR_X3S_TRACE"):
    COMMON = COMMON + ["-DX3S_TRACE=1"]
if os.environ.get("IRR_X3_ABL"):
    COMMON = COMMON + ["-DX3_ABL=" + os.environ["IRR_X3_ABL"]]
if os.environ.get("IRR_WX3_STAGE_OLD"):
    COMMON = COMMON + ["-DWX3_STAGE_OLD=" + os.environ["IRR_WX3_STAGE_OLD"]]
if os.environ.get("IRR_X3S_PRODUCERS_OLD"):
    COMMON = COMMON + ["-DX3S_PRODUCERS_OLD=" + os.environ["IRR_X3S_PRODUCERS_OLD"]]
if os.environ.get("IRR_WX3_TRACE"):
    COMMON = COMMON + ["-DWX3_TRACE=1"]
if os.environ.get("IRR_WX3_STAGGER"):
    COMMON = COMMON + ["-DWX3_STAGGER=" + os.environ["IRR_WX3_STAGGER"]]
if os.environ.get("IRR_CORR_ABL"):
    COMMON = COMMON + ["-DCORR_ABL=" + os.environ["IRR_CORR_ABL"]]
if os.environ.get("IRR_DEFS"):              # generic form: IRR_DEFS="-DWX3_MINI=0 -DWX3_TSTAGE=0"
    COMMON = COMMON + os.environ["IRR_DEFS"].split()
if os.environ.get("IRR_WG_TR4"):
    COMMON = COMMON + ["-DWG_TR4=1"]
if len(COMMON) != _BASE_FLAGS and not TAG:
    raise RuntimeError("ablation / trace macros change the kernels: set IRR_BUILD_TAG=<name> so the build goes to "
                       "irr_amd/lib_<name>/ instead of replacing the product library")
EXTRA = {"warp.hip": ["-ffp-contract=off"], "resize.hip": ["-ffp-contract=off"],
         "augment.hip": ["-ffp-contract=off"]}
# Round 5: the vectorisers are back ON for the files below -- HBM-bound elementwise / stencil kernels without an
# MFMA in them; what the global flag cost was measured on the cost-volume gradients (263 -> 266 us).  The condition is checked on the
# MACHINE CODE, not assumed: tools/scan_pk_swap.py (tests/test_pk_swap_scan.py, CPU suite) must find no packed fp32 instruction whose
# low result reads the high half of its own destination pair in the linked library -- the one form that deviated beside foreign MFMA
# waves (profiles/NOTES.md C.3); a file whose vectorised build contains that form goes back on the global flags.  Files whose kernels
# run on the lane or in the backward pass beside the lane's MFMA kernels (conv_*.hip, wgrad_reduce.hip, warp.hip, misc.hip,
# refine.hip, pack_batch.hip) keep the flags regardless.
# (tried and put back: loss.hip -- f1_bwd_kernel / f1_multi_bwd_kernel come out with v_pk_add_f32 / v_pk_mul_f32 of that form -- and
# augment.hip -- affine_warp_kernel / affine_flow_occ_kernel: 9 hits in the scan)
VECTORISE = ("corr.hip", "adam.hip", "resize.hip")
_NOVEC = ("-fno-slp-vectorize", "-fno-vectorize")


def flags_for(src: str):
    base = [f for f in COMMON if not (src in VECTORISE and f in _NOVEC)] if not os.environ.get("IRR_NO_VECTORISE") else list(COMMON)
    return base + EXTRA.get(src, [])


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra: str = "") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for f in paths:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _read(path: str) -> str:
    try:
        return open(path).read().strip()
    except OSError:
        return ""


def sources():
    """everything the library is built from: csrc/*.hip, csrc/*.h, the public header and this recipe"""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + \
        [os.path.join(ROOT, "include", "irr_hip.h"), os.path.abspath(__file__)]


def source_hash() -> str:
    """sha256 over the library's sources (16 hex digits): profiles/hbm_traffic.json stores the hash of the sources its PMC
    passes were measured on, bench.py refuses to report those bytes for a library built from other sources"""
    import hashlib
    h = hashlib.sha256()
    for f in sources():
        if f.endswith("build.py"):
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def stale(lib: str = LIB) -> bool:
    """library missing, or built from other sources than the ones in the tree (content hash written next to it by build():
    modification times do not survive the copy to the GPU box)"""
    stamp = os.path.join(os.path.dirname(lib), "source.hash")
    return (not os.path.exists(lib)) or (not os.path.exists(stamp)) or open(stamp).read().strip() != source_hash()


def built_hash(lib: str = LIB) -> str:
    """the source hash the library at ``lib`` was linked from ("" if unknown) -- bench.py / hip.py compare it with the tree"""
    return _read(os.path.join(os.path.dirname(lib), "source.hash"))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
        [os.path.join(ROOT, "include", "irr_hip.h")]
    hdr_digest = _digest(hdrs)
    jobs, stamps = [], {}
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s[:-4] + ".o")
        flags = flags_for(s)
        want = _digest([src], hdr_digest + " ".join(flags))
        stamps[obj] = want
        if force or not os.path.exists(obj) or _read(obj + ".hash") != want:
            jobs.append((obj, [hipcc] + flags + ["-c", src, "-o", obj]))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        return r

    def compile_one(job):
        obj, cmd = job
        if os.path.exists(obj + ".hash"):
            os.remove(obj + ".hash")               # (an interrupted compile must not leave a valid stamp behind)
        run(cmd)
        with open(obj + ".hash", "w") as f:
            f.write(stamps[obj])

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJDIR, s[:-4] + ".o") for s in srcs]
    link_id = _digest([], " ".join(stamps[o] for o in objs))
    link_stamp = os.path.join(LIBDIR, "link.hash")
    if force or jobs or not os.path.exists(LIB) or _read(link_stamp) != link_id:
        for st in (link_stamp, os.path.join(LIBDIR, "source.hash")):
            if os.path.exists(st):
                os.remove(st)
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
        with open(link_stamp, "w") as f:
            f.write(link_id)
    with open(os.path.join(LIBDIR, "source.hash"), "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
