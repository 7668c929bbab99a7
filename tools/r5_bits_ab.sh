#!/bin/bash
# GPU box: same-box A/B of the bit masks of the streaming kernel (IRR_X3S_BITS=0: fp32 activations as LeakyReLU' masks, as before)
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 20 --warmup 5"
for rep in 1 2 3; do
  for V in 0 1; do
    R=$(IRR_X3S_BITS=$V python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "rep $rep IRR_X3S_BITS=$V : $R"
  done
done
