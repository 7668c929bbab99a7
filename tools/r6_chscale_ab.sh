#!/bin/bash
# cost of the per-channel scale of the weight gradient's gy-role operand (one channel-maxima pass per fp16x2 weight gradient, on the lane):
# same box, alternating.   bash tools/r6_chscale_ab.sh [rounds]
R=${1:-3}
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3"
mkdir -p gpurun_out/chs
for i in $(seq 1 $R); do
  IRR_WGRAD_CHANNEL_SCALE=0 python bench.py $F 2>/dev/null > gpurun_out/chs/off_$i.json
  python bench.py $F 2>/dev/null > gpurun_out/chs/on_$i.json
done
python - <<'PY'
import json, glob
for tag in ("off", "on"):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/chs/{tag}_*.json"))]
    print(f"IRR_WGRAD_CHANNEL_SCALE {tag:3s}: pairs/s {[round(d['value'], 1) for d in v]}  ms/step {[round(d['ms_per_step'], 2) for d in v]}  launches amax_channels {v[0]['launches_per_step'].get('amax_channels')}")
PY
