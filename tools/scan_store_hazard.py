"""Static guard for round 5's finding (profiles/NOTES.md D.5): on gfx950 a vector-memory STORE of more than 64 bits whose
data registers are overwritten by the very next VALU instructions can write the NEW register contents to memory -- seen with
`buffer_store_dwordx4 v[8:11], v12, s[80:83], s68 offen` (an SGPR in the soffset field) followed at once by `v_add_f32 v8, ...`
in the epilogue of conv_x3s_kernel: component 0 of the quad, lanes 12-15 of every 16-lane row, under load from another kernel or
process.  hipcc's hazard recognizer inserts wait states behind such a store only when soffset is NOT a register (the documented
rule); with 8 wait states pinned behind the store (`asm volatile("s_nop 7" : "+v"(data))`) the fault is gone.

This disassembles every gfx950 code object in a built library and lists wide stores (buffer / global / flat / scratch, x3 / x4) whose
data registers are written again within WINDOW wait states (VALU / s_nop counted; any other instruction counts one).

    python tools/scan_store_hazard.py [path to libirr_hip.so] [window]      exit status 1 if any is found
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scan_pk_swap import LLVM, code_objects  # noqa: E402

WIDE = re.compile(r"^(buffer_store_dwordx[34]|buffer_store_format_xyzw?|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\b")
WINDOW = 5          # wait states that must separate the store from the first overwrite of its data


def regs(tok):
    """'v[8:11]' / 'v12' -> set of VGPR numbers"""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def data_regs(line):
    ops = [o.strip() for o in line.split(None, 1)[1].split(",")]
    if line.startswith("buffer_store"):
        return regs(ops[0])                       # vdata, vaddr, srsrc, soffset
    return regs(ops[1])                           # global / flat / scratch: vaddr, vdata, ...


def dest_regs(line):
    """VGPRs an instruction writes (first operand of v_* instructions; loads: their destination)"""
    head = line.split(None, 1)
    if len(head) < 2:
        return set()
    op, rest = head
    first = rest.split(",")[0].strip()
    if op.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
        return set()
    if op.startswith("v_") or op.startswith(("buffer_load", "global_load", "flat_load", "ds_read", "ds_bpermute", "ds_permute", "scratch_load")):
        return regs(first)
    return set()


def scan(lib, window=WINDOW):
    hits, total = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            kernel, body = "?", []
            for raw in dis.splitlines() + ["0 <end>:"]:
                m = re.match(r"[0-9a-f]+ <(.+)>:", raw)
                if m:
                    for i, line in enumerate(body):
                        if not WIDE.match(line):
                            continue
                        total += 1
                        data, ws = data_regs(line), 0
                        for nxt in body[i + 1:i + 1 + window + 2]:
                            if nxt.startswith("s_nop"):
                                ws += int(nxt.split()[1]) + 1
                                continue
                            if nxt.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                                break
                            if ws < window and dest_regs(nxt) & data:
                                hits.append((kernel, line, nxt, ws))
                                break
                            ws += 1
                            if ws >= window:
                                break
                    kernel, body = m.group(1), []
                    continue
                line = raw.split("//")[0].strip()
                if line:
                    body.append(line)
    return total, hits


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irr_amd", "lib", "libirr_hip.so")
    window = int(sys.argv[-1]) if sys.argv[-1].isdigit() else WINDOW
    total, hits = scan(lib, window)
    print(f"{lib}: {total} wide vector-memory stores, {len(hits)} whose data registers are rewritten within {window} wait states")
    by = {}
    for k, st, nx, ws in hits:
        by.setdefault(k, []).append((st, nx, ws))
    for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print(f"  {len(v):4d}  {k[:110]}")
        for st, nx, ws in v[:2]:
            print(f"          {st}   ->   {nx}   (after {ws} wait states)")
    sys.exit(1 if hits else 0)
