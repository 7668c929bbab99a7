"""Joins the per-group outputs of tools/rocpd_pmc.py (tools/r6_pmc.sh) into one table per kernel: matrix-pipe busy share, effective
clock, issue-stall / LDS-stall shares, VALU and LDS instructions per MFMA.   usage: r6_pmc_table.py mfma.txt lds.txt wait.txt
Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's SIMDs... (= 32 x N_mfma for 32x32x16);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs."""
import re, sys
K = {}
for f in sys.argv[1:]:
    cur = None
    try:
        lines = open(f).read().splitlines()
    except OSError:
        continue
    for ln in lines:
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+) avg_us=([\d.]+)", ln)
        if m:
            cur = K.setdefault(m.group(1), {"us": {}})
            cur["us"][f] = float(m.group(3))
            continue
        m = re.match(r"^\s+(\S+)\s+(\d+) per dispatch", ln)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
SIMDS = 256 * 4
for name, c in K.items():
    us = list(c["us"].values())
    print(f"{name}   avg_us per pass: {', '.join(f'{u:.1f}' for u in us)}")
    gui = c.get("GRBM_GUI_ACTIVE")
    if gui:
        clk = gui / 8.0                      # cycles of the launch
        t = us[0] * 1e-6
        print(f"    clock (GRBM_GUI_ACTIVE / 8 / duration)      {clk / t / 1e9:6.2f} GHz   ({clk:.3e} cycles)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (clk * SIMDS)
            print(f"    matrix pipe busy                            {busy * 100:6.1f} %   x clock/2.4 = {busy * clk / t / 2.4e9:5.3f} of nominal")
            print(f"    MFMAs (busy / 32)                           {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 32:.4e}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for k_, lab in (("SQ_WAIT_INST_ANY", "issue stall (WAIT_INST_ANY)"), ("SQ_WAIT_ANY", "parked: waitcnt / barrier (WAIT_ANY)"),
                        ("SQ_ACTIVE_INST_ANY", "issuing (ACTIVE_INST_ANY)"), ("SQ_WAIT_INST_LDS", "LDS issue stall (WAIT_INST_LDS)")):
            if k_ in c:
                print(f"    {lab:43s} {c[k_] / wc * 100:6.1f} % of wave cycles")
    nm = c.get("SQ_INSTS_MFMA") or (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 32 if "SQ_VALU_MFMA_BUSY_CYCLES" in c else None)
    if nm:
        for k_ in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
            if k_ in c:
                print(f"    {k_ + ' per MFMA':43s} {c[k_] / nm:6.2f}")
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        print(f"    LDS bank-conflict cycles / LDS active cycles {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] * 100:6.1f} %")
        if gui:
            print(f"    LDS array active                            {c['SQ_LDS_IDX_ACTIVE'] / (gui / 8.0 * 256) * 100:6.1f} % of CU cycles")
