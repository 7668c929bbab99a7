F="--no-cpu-baseline --no-secondary --no-extra-legs --no-kernel-timer --steps 4 --warmup 1"
run() { echo "== $1"; shift; env "$@" IRR_GRAD_FINITE_LOG=1 python bench.py $F 2>&1 | grep -a "grad log\|NaN\|\"value\"" | cut -c1-160; }
run "bits, write only (data gradients read the fp32 masks)" IRR_X3S_BITS_NOREAD=1
run "bits, lane group 1" IRR_LANE_GROUP=1
run "bits, lane lead 1" IRR_LANE_MAX_LEAD=1
run "bits, hold occup" IRR_LANE_HOLD_OCCUP=1
run "bits (default)" IRR_DUMMY=1
run "no bits" IRR_X3S_BITS=0
