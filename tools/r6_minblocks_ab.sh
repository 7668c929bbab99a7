#!/bin/bash
# routing threshold of the fp16x2 3x3 kernels (irr_conv_x3_set_min_blocks, default 384): launches with fewer blocks stay on the
# fp32-MFMA kernels.  Same box, alternating.   bash tools/r6_minblocks_ab.sh [rounds]
R=${1:-2}
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3"
mkdir -p gpurun_out/minb
for i in $(seq 1 $R); do
  for n in 384 96 192 256 768; do
    IRR_X3_MIN_BLOCKS=$n python bench.py $F 2>/dev/null > gpurun_out/minb/n${n}_$i.json
  done
done
python - <<'PY'
import json, glob
for n in (96, 192, 256, 384, 768):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/minb/n{n}_*.json"))]
    print(f"IRR_X3_MIN_BLOCKS {n:4d}: pairs/s {[round(d['value'], 1) for d in v]}  ms/step {[round(d['ms_per_step'], 2) for d in v]}")
PY
