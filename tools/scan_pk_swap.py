"""Static guard for round 4's finding (profiles/NOTES.md C.3, tools/pkfma_swap.py): on gfx950 a packed fp32 instruction whose LOW result
reads the HIGH half of ITS OWN DESTINATION register pair (op_sel bit of that source set: v_pk_fma_f32 ... op_sel:[0,0,1] op_sel_hi:[0,1,0], the
accumulator-swap form the SLP vectoriser emits, and the v_pk_add / v_pk_mul equivalents) returned a wrong low result in lanes 48..63
-- the accumulator read as zero -- while waves of another kernel that streams MFMAs ran beside it; never alone, never with the halves
in place.  This disassembles every gfx950 code object inside a built library and lists the packed fp32 instructions of that form.

    python tools/scan_pk_swap.py [path to libirr_hip.so]          exit status 1 if any is found
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_objects(lib, tmp):
    """-> paths of the gfx950 code objects bundled into ``lib`` (one bundle per translation unit in .hip_fatbin)"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for n, s in enumerate(starts):
        piece = os.path.join(tmp, f"bundle{n}.bin")
        open(piece, "wb").write(blob[s:starts[n + 1] if n + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, f"co{n}.elf")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + piece, "--targets=" + TARGET,
                        "--output=" + co], check=True, capture_output=True)
        out.append(co)
    return out


PACKED = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")


def low_reads_high(line):
    """sources (0-based) of a disassembled packed instruction that ARE THE DESTINATION register pair and whose high half feeds the low
    result (the forms that deviated in tools/pkfma_swap.py; a high half of ANOTHER pair -- form 5 there, the broadcasts of
    corr81_fwd4_kernel -- did not)"""
    ops = [o.strip() for o in line.split(None, 1)[1].split("op_sel")[0].split(",") if o.strip()]
    srcs = ops[1:]                                       # (ops[0] is the destination)
    sel = re.search(r"op_sel:\[([01,]+)\]", line)
    if not sel:
        return []
    bits = [int(v) for v in sel.group(1).split(",")]
    return [i for i, b in enumerate(bits) if b == 1 and i < len(srcs) and srcs[i].startswith("v[") and srcs[i] == ops[0]]


def scan(lib):
    """-> (number of packed fp32 instructions, [(kernel, instruction)] whose low result reads the high half of its destination pair)"""
    hits, total = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            kernel = "?"
            for line in dis.splitlines():
                m = re.match(r"[0-9a-f]+ <(.+)>:", line)
                if m:
                    kernel = m.group(1)
                    continue
                line = line.split("//")[0].strip()
                if not line.startswith(PACKED):
                    continue
                total += 1
                if low_reads_high(line):
                    hits.append((kernel, line))
    return total, hits


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irr_amd", "lib", "libirr_hip.so")
    total, hits = scan(lib)
    print(f"{lib}: {total} packed fp32 instructions, {len(hits)} whose low result reads the high half of its destination pair")
    for k, ins in hits[:40]:
        print("  ", k, "|", ins)
    sys.exit(1 if hits else 0)
