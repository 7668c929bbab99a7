#!/bin/bash
# GPU box: the guarded stores of conv_x3s_kernel (product library) against the compiler's own placement (tagged build
#   IRR_BUILD_TAG=unguarded IRR_DEFS="-DX3S_STORE_UNGUARDED=1" python -m irr_amd.build), NOTES D.5:
# 1. the in-process case: bs32 backward passes with the weight-gradient lane on (tools/lane_race_probe.py, bit masks on),
# 2. the two-process case of NOTES D.4: two copies of tools/r5_concurrency_probe.py at once (3-9 % wrong launches with the round-5 library).
U=$PWD/irr_amd/lib_unguarded/libirr_hip.so
echo "== 1. lane probe, PB=32, 8 passes: unguarded"; IRR_HIP_LIB=$U PB=32 python tools/lane_race_probe.py 8 2>&1 | grep -a "^lane" | cut -c1-64
echo "== 1. lane probe, PB=32, 8 passes: guarded (product)"; PB=32 python tools/lane_race_probe.py 8 2>&1 | grep -a "^lane" | cut -c1-64
echo "== 2. two copies of r5_concurrency_probe.py 400: unguarded"; bash tools/r5_two_copies.sh 400 $U
echo "== 2. two copies of r5_concurrency_probe.py 400: guarded (product)"; bash tools/r5_two_copies.sh 400
echo "== 2. two copies of r5_concurrency_probe.py 400: unguarded again"; bash tools/r5_two_copies.sh 400 $U
echo "== 2. two copies of r5_concurrency_probe.py 400: guarded again"; bash tools/r5_two_copies.sh 400
