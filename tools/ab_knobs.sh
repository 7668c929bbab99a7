# same-box A/B of environment knobs: bash tools/ab_knobs.sh  (prints pairs/s and ms/step per setting)
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run lanes1 IRR_WGRAD_LANES=1
run lanes2_45k IRR_WGRAD_LANES=2
run lanes3_45k IRR_WGRAD_LANES=3
run lanes3_11k IRR_WGRAD_LANES=3 IRR_WGRAD_SMALL_PIX=11000
run lanes3_180k IRR_WGRAD_LANES=3 IRR_WGRAD_SMALL_PIX=180000
run lanes4_45k IRR_WGRAD_LANES=4
run lanes1 IRR_WGRAD_LANES=1
