# on the GPU box: the unprofiled bench lines of a profile set (tools/profile_round.sh step 1 and the A/B legs) -> gpurun_out/prof/
TAG=${1:-r3}
OUT=gpurun_out/prof; mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --no-async-wgrad --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_serial.json 2>> $OUT/${TAG}_bench.err
IRR_CONV_MATH=f32 python bench.py --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_math_f32.json 2>> $OUT/${TAG}_bench.err
python bench.py --no-cpu-baseline --batch 8 --height 448 --width 1024 > $OUT/${TAG}_bench_448x1024_bs8.json 2>> $OUT/${TAG}_bench.err
cut -c1-140 $OUT/${TAG}_bench.json $OUT/${TAG}_bench_serial.json $OUT/${TAG}_bench_math_f32.json $OUT/${TAG}_bench_448x1024_bs8.json
