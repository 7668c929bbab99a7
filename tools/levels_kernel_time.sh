#!/bin/bash
# GPU box: kernel tables of back-to-back passes truncated after level 0, 1, 2, 3 (tools/levels_kernel_time.py)
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
for L in 0 1 2 3; do
  rocprofv3 --kernel-trace --stats -d $OUT/lk$L -o lk$L -- python3 tools/levels_kernel_time.py $L > $OUT/r5_levels_kernel_time_L$L.txt 2>/dev/null
  python tools/rocpd_stats.py $(find $OUT/lk$L -name "*.db" | head -1) 45 >> $OUT/r5_levels_kernel_time_L$L.txt
  rm -rf $OUT/lk$L
done
head -3 $OUT/r5_levels_kernel_time_L*.txt
