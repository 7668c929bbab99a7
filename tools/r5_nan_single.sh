#!/bin/bash
# Round 5, NOTES C.5: is the two-rank NaN of the plain-low-piece build a matter of rank 1's DATA?  Single process, 448x1024 x 8 pairs,
# the streaming kernel's fp16x2 form on, batches of rank <offset>: bash tools/r5_nan_single.sh <runs> <offset> [env ...]
R=${1:-3}; OFF=${2:-1}; shift; shift
OUT=gpurun_out/nan_single.txt
mkdir -p gpurun_out
echo "=== single process, seed offset $OFF ($*)" | tee -a $OUT
for i in $(seq 1 $R); do
  ( export IRR_X3S_H2=1 IRR_BENCH_SEED_OFFSET=$OFF IRR_GRAD_FINITE_LOG=1; for kv in "$@"; do export "$kv"; done
    timeout 300 python bench.py --batch 8 --height 448 --width 1024 --steps 12 --warmup 2 --no-secondary --no-extra-legs --no-cpu-baseline --no-kernel-timer > /tmp/ns_o.txt 2> /tmp/ns_e.txt
    echo "run $i: rc=$? NaN=$(grep -c 'is NaN' /tmp/ns_e.txt) gradlog=$(grep -c 'grad log' /tmp/ns_e.txt)" ) | tee -a $OUT
  grep -h "grad log" /tmp/ns_e.txt | cut -c1-4000 | head -2 | tee -a $OUT
done
