#!/bin/bash
# kernel tables (serial: --no-async-wgrad) with and without the per-channel scale of the weight gradient's gy-role operand
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --no-async-wgrad --steps 4 --warmup 2"
IRR_WGRAD_CHANNEL_SCALE=0 rocprofv3 --kernel-trace --stats -d $OUT/k0 -o k0 -- python3 bench.py $F > /dev/null 2> $OUT/chs0.err
python tools/rocpd_stats.py $(find $OUT/k0 -name "*.db" | head -1) 60 > $OUT/r6_chs_off_kernel_stats.txt; rm -rf $OUT/k0
rocprofv3 --kernel-trace --stats -d $OUT/k1 -o k1 -- python3 bench.py $F > /dev/null 2> $OUT/chs1.err
python tools/rocpd_stats.py $(find $OUT/k1 -name "*.db" | head -1) 60 > $OUT/r6_chs_on_kernel_stats.txt; rm -rf $OUT/k1
python - <<'PY'
import re
def load(f):
    d = {}
    for ln in open(f):
        m = re.match(r"^(\S.*?)\s{2,}(\d+)\s+([\d.]+)\s+([\d.]+)", ln)
        if m: d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load("gpurun_out/prof/r6_chs_off_kernel_stats.txt"), load("gpurun_out/prof/r6_chs_on_kernel_stats.txt")
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort(reverse=True)
print("largest differences (on - off), ms over the profiled steps:")
for d, k, ca, ta, cb, tb in rows[:14] + rows[-5:]:
    print(f"{d:+9.2f}  {k[:90]:90s} off {ca:5d} x {ta:8.2f}   on {cb:5d} x {tb:8.2f}")
PY
