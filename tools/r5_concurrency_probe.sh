#!/bin/bash
# bash tools/r5_concurrency_probe.sh [N]      (GPU box, repo root): the probe alone and beside a second process, both libraries
N=${1:-150}
L=$PWD/irr_amd/lib_loplain/libirr_hip.so
for lib in "" "$L"; do
  echo "== alone, library ${lib:-product}"; IRR_HIP_LIB=$lib python tools/r5_concurrency_probe.py $N 2>&1 | tail -2
  echo "== beside bench.py (second process), library ${lib:-product}"
  IRR_HIP_LIB=$lib python bench.py --steps 60 --warmup 2 --no-cpu-baseline --no-secondary --no-extra-legs > /dev/null 2>&1 &
  BP=$!
  sleep 25
  IRR_HIP_LIB=$lib python tools/r5_concurrency_probe.py $N 2>&1 | tail -2
  wait $BP
done
