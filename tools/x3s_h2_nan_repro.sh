#!/bin/bash
# profiles/NOTES.md C.5: the open NaN of the streaming kernel's fp16x2 form when two ranks share one GPU.
#   bash tools/x3s_h2_nan_repro.sh [runs] [extra env assignments ...]      (on the GPU box, from the repository root)
# e.g.  bash tools/x3s_h2_nan_repro.sh 5 IRR_CONV_CHECK_FINITE=slots
#       bash tools/x3s_h2_nan_repro.sh 5 AMD_SERIALIZE_KERNEL=3
#       bash tools/x3s_h2_nan_repro.sh 5 IRR_X3S_NO_FUSED_AMAX=1
# Prints, per run, the exit status, the number of "training_loss is NaN" lines and (with IRR_CONV_CHECK_FINITE=slots) the slot log.
R=${1:-5}; shift
export IRR_DDP_BACKEND=gloo IRR_X3S_H2=1
for kv in "$@"; do export "$kv"; done
for i in $(seq 1 $R); do
  python bench.py --gpus 2 --steps 2 --warmup 2 --batch 2 --no-cpu-baseline > /tmp/x3s_nan_o_$i.txt 2> /tmp/x3s_nan_e_$i.txt
  echo "run $i: rc=$? NaN lines $(grep -c 'is NaN' /tmp/x3s_nan_e_$i.txt)"
  grep -h -A40 "slot log\|finite log" /tmp/x3s_nan_e_$i.txt | cut -c1-220 | head -44
done
