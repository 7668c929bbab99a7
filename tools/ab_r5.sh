# Same-box A/B of the current tree against the END-OF-ROUND-5 tree (r5tree/ = git archive of 82f2c7e with its own library; not tracked):
#   bash tools/ab_r5.sh [rounds]      (run on the GPU box from the repository root)
# Recreate r5tree/ in the CPU container before the gpurun call:
#   mkdir -p r5tree && git archive 82f2c7e irr_amd include tools bench.py oracle profiles/hbm_traffic.json profiles/hbm_traffic_448x1024.json | tar -x -C r5tree && (cd r5tree && python -m irr_amd.build)
# Alternating: round 5 as shipped, this tree (default), this tree with IRR_WGRAD_CHANNEL_SCALE=0 (round 5's arithmetic for the weight
# gradient's plain-pair operand); timers off on all three.
R=${1:-3}
F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3"
mkdir -p gpurun_out/ab5
for i in $(seq 1 $R); do
  (cd r5tree && python bench.py $F 2>/dev/null) > gpurun_out/ab5/r5_$i.json
  python bench.py $F 2>/dev/null > gpurun_out/ab5/r6_$i.json
  IRR_WGRAD_CHANNEL_SCALE=0 python bench.py $F 2>/dev/null > gpurun_out/ab5/r6off_$i.json
done
python - <<'PY'
import json, glob
for tag, what in (("r5", "end of round 5 (82f2c7e)"), ("r6", "this tree"), ("r6off", "this tree, IRR_WGRAD_CHANNEL_SCALE=0")):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/ab5/{tag}_*.json"))]
    print(f"{what:42s} pairs/s {[round(d['value'], 1) for d in v]}  ms/step {[round(d['ms_per_step'], 2) for d in v]}")
PY
