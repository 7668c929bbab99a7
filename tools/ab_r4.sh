# Same-box A/B of the current tree against the END-OF-ROUND-4 tree (r4tree/ = git archive of 4bada97 with its own library; not tracked):
#   bash tools/ab_r4.sh [rounds]      (run on the GPU box from the repository root)
# Recreate r4tree/ here (CPU container) before the gpurun call:
#   mkdir -p r4tree && git archive 4bada97 irr_amd include tools bench.py oracle | tar -x -C r4tree && (cd r4tree && python -m irr_amd.build)
# Alternates the benches so that box-to-box and thermal differences cancel: round 4 as shipped (streaming kernel on bf16x3), round 4
# with IRR_X3S_H2=1 (its fastest, NaN-prone configuration), this tree (default), each with and without the kernel timers.
R=${1:-3}
for i in $(seq 1 $R); do
  (cd r4tree && python bench.py --no-cpu-baseline --no-secondary --no-extra-legs --steps 10 --warmup 3 2>/dev/null) > gpurun_out/ab4_r4_$i.json
  python bench.py --no-cpu-baseline --no-secondary --no-extra-legs --steps 10 --warmup 3 2>/dev/null > gpurun_out/ab4_r5_$i.json
  (cd r4tree && IRR_X3S_H2=1 python bench.py --no-cpu-baseline --no-secondary --no-extra-legs --steps 10 --warmup 3 2>/dev/null) > gpurun_out/ab4_r4h2_$i.json
  python bench.py --no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3 2>/dev/null > gpurun_out/ab4_r5nt_$i.json
  (cd r4tree && python bench.py --no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 10 --warmup 3 2>/dev/null) > gpurun_out/ab4_r4nt_$i.json
done
python - <<'PY'
import json, glob
for tag in ("r4", "r4h2", "r5", "r4nt", "r5nt"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/ab4_{tag}_*.json")):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            vals.append((d["value"], d["ms_per_step"]))
        except Exception as e:
            vals.append(("?", str(e)[:40]))
    print(tag, vals)
PY
