#!/bin/bash
# the lane decided by measurement at the current main-stream speed, one box, alternating:
#   (a) --no-async-wgrad (weight gradients in line), (b) the lane (default), (c) the lane for the MFMA-bound levels only
#   (IRR_LANE_OCCUP_INLINE=1: the upsampler nodes of levels 5-6 keep their weight gradients on the issuing stream).   bash tools/r6_lane.sh [rounds]
R=${1:-3}
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3"
mkdir -p gpurun_out/lane
for i in $(seq 1 $R); do
  python bench.py $F --no-async-wgrad 2>/dev/null > gpurun_out/lane/a_$i.json
  python bench.py $F 2>/dev/null > gpurun_out/lane/b_$i.json
  IRR_LANE_OCCUP_INLINE=1 python bench.py $F 2>/dev/null > gpurun_out/lane/c_$i.json
done
python - <<'PY'
import json, glob
names = {"a": "(a) weight gradients in line (--no-async-wgrad)", "b": "(b) lane for every level (default)", "c": "(c) lane for levels 0-4, levels 5-6 in line (IRR_LANE_OCCUP_INLINE=1)"}
for tag in "abc":
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/lane/{tag}_*.json")):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1]); vals.append((d["value"], d["ms_per_step"]))
        except Exception as e:
            vals.append(("?", str(e)[:40]))
    ok = [v for v in vals if v[0] != "?"]
    mean = sum(v[0] for v in ok) / max(1, len(ok))
    print(f"{names[tag]:75s} pairs/s {[round(v[0], 1) for v in ok]}  mean {mean:.1f}  ms/step {[round(v[1], 2) for v in ok]}")
PY
