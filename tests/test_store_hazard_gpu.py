"""Round 5 (profiles/NOTES.md D.5, profiles/r5_store_hazard_standalone.txt): the wait states a 16-byte buffer store needs on gfx950 before a
VALU instruction rewrites its data registers, measured with the stand-alone victim of tools/store_hazard.hip: TWO behind the literal-soffset
form (what hipcc inserts), ONE behind the SGPR-soffset form (hipcc inserts none: with none, lanes 12-15 of every 16-lane row store the
rewritten register -- the corruption of conv_x3s_kernel).  The library pins eight (irr_buffer_store_b128_guarded); this test keeps the
measured rule on record: with the required wait states no slot is wrong."""
import ctypes
import os
import shutil
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    src, so = os.path.join(ROOT, "tools", "store_hazard.hip"), os.path.join(ROOT, "tools", "_store_hazard.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("no hipcc to build tools/store_hazard.hip")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.launch_store_victim.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def _wrong(lib, out, nblk, iters, form, ws):
    out.zero_()
    assert lib.launch_store_victim(out.data_ptr(), nblk, iters, nblk * 256 * 16, form, ws, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    v = out.view(-1, 4)
    assert bool((v[:, 1:] == 1.0).all())                   # components 1..3 are never rewritten
    return int((v[:, 0] == 2.0).sum()), int(((v[:, 0] != 2.0) & (v[:, 0] != 1.0)).sum())


def test_wait_states_behind_a_wide_buffer_store():
    lib = _lib()
    nblk, iters = 1024, 64
    out = torch.empty(iters * nblk * 256 * 4, device="cuda")
    report = {}
    for form, need in ((0, 1), (1, 2)):                    # (SGPR soffset, literal soffset 0): wait states that suffice
        for ws in (0, 1, 2, 8):
            wrong = other = 0
            for _ in range(3):
                w, o = _wrong(lib, out, nblk, iters, form, ws)
                wrong += w
                other += o
            report[(form, ws)] = wrong
            assert other == 0, (form, ws, other)
            if ws >= need:
                assert wrong == 0, (form, ws, wrong)
    print("wrong slots of 3 x 16.8 M stores by (form, wait states):", report)
    # (not asserted: that fewer wait states DO fail -- (0, 0): ~1.5 % of the slots, lanes 12-15 of each row; (1, 0): nearly all; (1, 1): ~3 %)
