F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 20 --warmup 5"
for rep in 1 2; do
for L in 0 2 3 4; do
  if [ $L = 0 ]; then R=$(IRR_BRANCH_STREAMS=0 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])");
  else R=$(IRR_BRANCH_STREAMS=1 IRR_BRANCH_LEVELS=$L python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"); fi
  echo "rep $rep branch_levels<$L : $R ms/step"
done; done
