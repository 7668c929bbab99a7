# Same-box A/B of the current tree against the round-3 tree (r3tree/ = git archive of 0e92235, built here; not tracked):
#   bash tools/ab_r3.sh [rounds]      (run on the GPU box from the repository root)
# Recreate r3tree/ here (CPU container) before the gpurun call:
#   mkdir -p r3tree && git archive 0e92235 irr_amd include tools bench.py oracle | tar -x -C r3tree && (cd r3tree && python -m irr_amd.build)
# Alternates the two benches so that box-to-box and thermal differences cancel; prints pairs/s and ms/step of each run.
R=${1:-3}
for i in $(seq 1 $R); do
  (cd r3tree && python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null) > gpurun_out/ab_r3_$i.json
  python bench.py --no-cpu-baseline --no-secondary --no-extra-legs --steps 10 --warmup 3 2>/dev/null > gpurun_out/ab_r4_$i.json
  IRR_BENCH_NOTIMER=1 python bench.py --no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3 2>/dev/null > gpurun_out/ab_r4nt_$i.json
  (cd r3tree && python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 2>/dev/null) > gpurun_out/ab_r3nt_$i.json
done
python - <<'PY'
import json, glob
for tag in ("r3", "r4", "r3nt", "r4nt"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/ab_{tag}_*.json")):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            vals.append((d["value"], d["ms_per_step"]))
        except Exception as e:
            vals.append(("?", str(e)[:40]))
    print(tag, vals)
PY
