# same-box A/B of environment knobs: bash tools/ab_knobs.sh  (prints pairs/s and ms/step per setting)
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run torch_events+record_stream IRR_LANE_TORCH_EVENTS=1 IRR_LANE_RECORD_STREAM=1
run device_events+record_stream IRR_LANE_RECORD_STREAM=1
run device_events X=1
run torch_events IRR_LANE_TORCH_EVENTS=1
done
