export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kts -o kts -- python3 bench.py --no-cpu-baseline --no-secondary --no-async-wgrad --steps 5 > $OUT/r3_bench_serial_under_rocprof.json 2>> $OUT/r3_bench.err
python tools/rocpd_stats.py $(find $OUT/kts -name "*.db" | head -1) 50 > $OUT/r3_kernel_stats_serial.txt
python tools/rocpd_timeline.py $(find $OUT/kts -name "*.db" | head -1) > $OUT/r3_timeline_serial.txt 2>&1
rm -rf $OUT/kts
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline --no-secondary --steps 5 > $OUT/r3_bench_under_rocprof.json 2>> $OUT/r3_bench.err
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) 50 > $OUT/r3_kernel_stats.txt
python tools/rocpd_timeline.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/r3_timeline_two_lane.txt 2>&1
rm -rf $OUT/kt
grep "^torch / runtime\|^step span" $OUT/r3_timeline_serial.txt $OUT/r3_timeline_two_lane.txt
