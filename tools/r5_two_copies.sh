#!/bin/bash
# bash tools/r5_two_copies.sh <N> [IRR_HIP_LIB path]: two copies of tools/r5_concurrency_probe.py at once, twice
for rep in 1 2; do
  IRR_HIP_LIB=$2 python tools/r5_concurrency_probe.py $1 2>/dev/null | tail -1 & P1=$!
  IRR_HIP_LIB=$2 python tools/r5_concurrency_probe.py $1 2>/dev/null | tail -1 & P2=$!
  wait $P1 $P2
done
