"""Audit of every gfx950 kernel inside libirr_hip.so: what the code object DECLARES against what the machine code TOUCHES
("a kernel whose descriptor under-declares what it uses explains an illegal instruction under multi-process
sharing, a register that reads 0 beside foreign waves, and a timing-dependent NaN at once").

Per kernel, from the metadata note (llvm-readelf --notes) and the kernel descriptor bytes (the .kd symbol in .rodata):
  vgpr_count / agpr_count / accum_offset / granulated VGPR blocks / LDS bytes / scratch bytes / dynamic stack / spills,
and from the disassembly (llvm-objdump -d): the highest v / a / s register any instruction names, every scratch_ / buffer ... offen
private access, every ds_ offset (lower bound of the LDS touched with static addressing), s_setreg / s_sethalt / inline-asm
oddities.  Flags a kernel when
  * an instruction names v[N] with N >= accum_offset (unified register file: arch VGPRs end at accum_offset) or N >= vgpr_count,
  * an instruction names a[N] with N >= agpr_count, or accum_offset + agpr_count > 512, or the granule count in the descriptor does
    not cover accum_offset + agpr_count,
  * scratch instructions exist but private_segment_fixed_size == 0 and no dynamic stack,
  * a ds_ instruction's immediate offset alone is beyond group_segment_fixed_size for a kernel with STATIC LDS only.

    python tools/codeobj_audit.py [libirr_hip.so] [--all]        exit status 1 if anything is flagged
"""
import os, re, struct, subprocess, sys, tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scan_pk_swap import LLVM, code_objects  # noqa: E402


def metadata(co):
    """-> {kernel name: dict of the scalar .keys of its metadata entry}"""
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:
            cur = {"agpr_count": int(m.group(1))}
            continue
        m = re.match(r"\s+\.(\w+):\s+(\S+)\s*$", line)
        if m and cur is not None and not line.startswith("      "):
            k, v = m.group(1), m.group(2)
            cur[k] = int(v) if v.isdigit() else v
            if k == "name":
                out[v] = cur
    return out


def descriptors(co):
    """-> {kernel name: dict decoded from the 64-byte kernel descriptor}"""
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", co], check=True, capture_output=True, text=True).stdout
    secs = subprocess.run([f"{LLVM}/llvm-readelf", "-SW", co], check=True, capture_output=True, text=True).stdout
    ro = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
    addr, off = int(ro.group(1), 16), int(ro.group(2), 16)
    blob = open(co, "rb").read()
    out = {}
    for line in syms.splitlines():
        p = line.split()
        if len(p) >= 8 and p[-1].endswith(".kd"):
            a = int(p[1], 16)
            kd = blob[off + a - addr: off + a - addr + 64]
            group, private = struct.unpack_from("<II", kd, 0)
            rsrc3, rsrc1, rsrc2 = struct.unpack_from("<III", kd, 44)
            gran_vgpr = rsrc1 & 0x3f                      # GRANULATED_WORKITEM_VGPR_COUNT (units of 8, minus 1, gfx90a+ unified)
            gran_sgpr = (rsrc1 >> 6) & 0xf
            accum_offset = ((rsrc3 & 0x3f) + 1) * 4       # COMPUTE_PGM_RSRC3.ACCUM_OFFSET (gfx90a+)
            tg_split = (rsrc3 >> 16) & 1
            out[p[-1][:-3]] = {"kd_group": group, "kd_private": private, "kd_total_vgpr": (gran_vgpr + 1) * 8,
                              "kd_accum_offset": accum_offset, "kd_tg_split": tg_split, "kd_gran_sgpr": gran_sgpr,
                              "kd_scratch_en": rsrc2 & 1, "kd_dx10_ieee": (rsrc1 >> 21) & 3, "kd_float_mode": (rsrc1 >> 12) & 0xff}
    return out


REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]")


def isa_use(co):
    """-> {kernel: dict(max_v, max_a, max_s, scratch, ds_max_off, mfma, pk_f32, setreg, lines)}"""
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    out, k = {}, None
    for line in dis.splitlines():
        m = re.match(r"[0-9a-f]+ <(.+)>:", line)
        if m:
            name = m.group(1)
            k = out.setdefault(name, {"max_v": -1, "max_a": -1, "max_s": -1, "scratch": 0, "ds_max_off": -1, "mfma": 0, "pk_f32": 0,
                                      "setreg": [], "lines": 0, "lds_direct": 0}) if not name.startswith("$") and ".kd" not in name else None
            continue
        if k is None:
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        k["lines"] += 1
        op = ins.split()[0]
        for m in REG.finditer(ins.split(None, 1)[1] if " " in ins else ""):
            if m.group(1):
                kind, hi = m.group(1), int(m.group(2))
            else:
                kind, hi = m.group(3), int(m.group(5))
            key = "max_" + kind
            if hi > k[key]:
                k[key] = hi
        if op.startswith("scratch_") or (op.startswith("buffer_") and " s[0:3]" in ins and "offen" in ins and False):
            k["scratch"] += 1
        if op.startswith("ds_"):
            for m in re.finditer(r"offset\d?:(\d+)", ins):
                mult = 1
                if "read2st64" in op or "write2st64" in op:
                    mult = 64 * (8 if "b64" in op else 4)
                elif "read2" in op or "write2" in op:
                    mult = 8 if "b64" in op else 4
                k["ds_max_off"] = max(k["ds_max_off"], int(m.group(1)) * mult)
        if "mfma" in op:
            k["mfma"] += 1
        if op in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
            k["pk_f32"] += 1
        if op.startswith("s_setreg") or op in ("s_sethalt", "s_trap", "s_setprio", "s_sleep"):
            k["setreg"].append(ins)
        if " lds" in ins and op.startswith(("buffer_load", "global_load")):
            k["lds_direct"] += 1
    return out


def audit(lib, show_all=False):
    flagged, rows = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            md, kd, isa = metadata(co), descriptors(co), isa_use(co)
            for name, m in md.items():
                d, u = kd.get(name, {}), isa.get(name)
                if u is None:
                    flagged.append((name, "no disassembly found"))
                    continue
                why = []
                acc_off = d.get("kd_accum_offset", 0)
                vg, ag = m.get("vgpr_count", 0), m.get("agpr_count", 0)
                arch = vg - ag if ag else vg                      # metadata vgpr_count is the unified total on gfx90a+
                if ag:
                    if u["max_v"] >= acc_off:
                        why.append(f"names v{u['max_v']} but arch VGPRs end at accum_offset {acc_off}")
                    if u["max_a"] >= ag:
                        why.append(f"names a{u['max_a']} but agpr_count is {ag}")
                    if acc_off + ag > d.get("kd_total_vgpr", 0):
                        why.append(f"accum_offset {acc_off} + agprs {ag} > allocated {d.get('kd_total_vgpr')}")
                else:
                    if u["max_a"] >= 0:
                        why.append(f"names a{u['max_a']} with agpr_count 0")
                    if u["max_v"] >= d.get("kd_total_vgpr", 0):
                        why.append(f"names v{u['max_v']} but only {d.get('kd_total_vgpr')} VGPRs are allocated")
                if u["max_v"] + 1 > (acc_off if ag else vg):
                    why.append(f"names v{u['max_v']}, metadata vgpr_count {vg} (arch {arch})")
                if d.get("kd_total_vgpr", 0) > 512:
                    why.append("more than 512 unified registers")
                if u["scratch"] and not m.get("private_segment_fixed_size") and m.get("uses_dynamic_stack") != "true":
                    why.append(f"{u['scratch']} scratch instructions, no private segment")
                if d.get("kd_private") != m.get("private_segment_fixed_size") or d.get("kd_group") != m.get("group_segment_fixed_size"):
                    why.append(f"descriptor group/private {d.get('kd_group')}/{d.get('kd_private')} != metadata "
                               f"{m.get('group_segment_fixed_size')}/{m.get('private_segment_fixed_size')}")
                if m.get("group_segment_fixed_size", 0) and u["ds_max_off"] >= m["group_segment_fixed_size"] and "x3s" not in name:
                    why.append(f"ds offset {u['ds_max_off']} >= static LDS {m['group_segment_fixed_size']}")
                if m.get("max_flat_workgroup_size", 0) and (d.get("kd_total_vgpr", 0)) * ((m["max_flat_workgroup_size"] + 255) // 256) > 512:
                    why.append(f"{d.get('kd_total_vgpr')} registers x {(m['max_flat_workgroup_size'] + 63) // 64} waves do not fit a CU "
                               f"(4 SIMDs x 512)")
                row = (name, vg, ag, acc_off, d.get("kd_total_vgpr"), u["max_v"], u["max_a"], m.get("sgpr_count"), u["max_s"],
                       m.get("group_segment_fixed_size"), u["ds_max_off"], m.get("private_segment_fixed_size"), u["scratch"],
                       m.get("vgpr_spill_count"), m.get("max_flat_workgroup_size"), u["mfma"], u["pk_f32"], u["lds_direct"],
                       ";".join(sorted(set(x.split()[0] for x in u["setreg"]))))
                rows.append(row)
                if why:
                    flagged.append((name, "; ".join(why)))
    return rows, flagged


def short(name):
    try:
        return subprocess.run([f"{LLVM}/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:70]
    except Exception:
        return name[:70]


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irr_amd", "lib", "libirr_hip.so")
    rows, flagged = audit(lib)
    print(f"{lib}: {len(rows)} kernels audited, {len(flagged)} flagged")
    hdr = ("kernel", "vgpr", "agpr", "acc_off", "alloc", "max_v", "max_a", "sgpr", "max_s", "lds", "ds_off", "priv", "scr_ins", "spill",
           "wg", "mfma", "pk32", "ldsdir", "special")
    if "--all" in sys.argv or True:
        print(" | ".join(hdr))
        for r in sorted(rows, key=lambda r: -(r[1] or 0)):
            if "--all" in sys.argv or (r[1] or 0) >= 128 or r[11] or r[13]:
                print(" | ".join([short(r[0])] + [str(x) for x in r[1:]]))
    for n, w in flagged:
        print("FLAG", short(n), "::", w)
    sys.exit(1 if flagged else 0)
