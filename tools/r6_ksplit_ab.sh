#!/bin/bash
# K-split launches finished inside the launch (irr_conv2d_fwd_h2_kfused, IRR_X3_KSPLIT_FUSED=1) against the finishing launch (default):
# same box, alternating.   bash tools/r6_ksplit_ab.sh [rounds]
R=${1:-3}
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3"
for i in $(seq 1 $R); do
  python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('finishing launch ', d['value'], d['ms_per_step'])"
  IRR_X3_KSPLIT_FUSED=1 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inside the launch', d['value'], d['ms_per_step'])"
done
