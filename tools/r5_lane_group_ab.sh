#!/bin/bash
# GPU box: lane group size (launches handed to the lane per main-stream event) after round 5's changes, same box, alternating
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 20 --warmup 5"
for rep in 1 2; do
  for G in 4 2 8 16 32; do
    R=$(IRR_LANE_GROUP=$G python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "rep $rep IRR_LANE_GROUP=$G : $R"
  done
done
