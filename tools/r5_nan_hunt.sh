#!/bin/bash
# Round 5, NOTES C.5: experiments on the two-rank shared-GPU NaN of conv_x3s_kernel<EPI,2>.  Run on the GPU box from the repo root:
#   bash tools/r5_nan_hunt.sh "<label>:<runs>:<env assignments separated by spaces>" ...
# Each spec runs `bench.py --gpus 2` (gloo, both ranks on cuda:0) <runs> times; prints rc, NaN lines, grad-log lines.
OUT=gpurun_out/nan_hunt.txt
mkdir -p gpurun_out
for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}; runs=${rest%%:*}; envs=${rest#*:}
  echo "=== $label ($envs)" | tee -a $OUT
  for i in $(seq 1 $runs); do
    ( export IRR_DDP_BACKEND=gloo IRR_X3S_H2=1; for kv in $envs; do export "$kv"; done
      timeout 300 python bench.py --gpus 2 --steps 2 --warmup 2 --batch 2 --no-cpu-baseline > /tmp/nh_o.txt 2> /tmp/nh_e.txt
      echo "run $i: rc=$? NaN=$(grep -c 'is NaN' /tmp/nh_e.txt) gradlog=$(grep -c 'grad log' /tmp/nh_e.txt)" ) | tee -a $OUT
    grep -h "grad log\|slot log\|finite log\|  #" /tmp/nh_e.txt | cut -c1-4000 | head -4 | tee -a $OUT
  done
done
