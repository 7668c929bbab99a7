#!/bin/bash
# Runs on the GPU box (gpurun): refreshes every artefact under gpurun_out/prof/ that profiles/ is built from.
#   bash tools/profile_round.sh <tag>
# 1. bench.py default (two-lane) with the CPU baseline          -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats, default and single-stream -> <tag>_kernel_stats{,_serial}.txt (+ the JSON line of the same run)
# 3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE)            -> <tag>_hbm_traffic.txt, hbm_traffic.json
TAG=${1:-r5}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
# PMC passes first: their per-launch bytes go into profiles/hbm_traffic*.json ON THIS BOX before the benches run, so that the bench
# lines of this round carry roofline.traffic (bench.py reports the bytes only for the very sources they were measured on)
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf -- python3 bench.py --no-cpu-baseline --no-secondary --no-extra-legs --no-async-wgrad --no-kernel-timer --steps 2 --warmup 1 > /dev/null 2>> $OUT/${TAG}_bench.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw -- python3 bench.py --no-cpu-baseline --no-secondary --no-extra-legs --no-async-wgrad --no-kernel-timer --steps 2 --warmup 1 > /dev/null 2>> $OUT/${TAG}_bench.err
F=$(find $OUT/pf -name "*.db" | head -1); W=$(find $OUT/pw -name "*.db" | head -1)
python tools/rocpd_traffic.py $F $W > $OUT/${TAG}_hbm_traffic.txt
python tools/rocpd_traffic_json.py $F $W > $OUT/hbm_traffic.json
rm -rf $OUT/pf $OUT/pw
cp $OUT/hbm_traffic.json profiles/hbm_traffic.json
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf2 -o pf2 -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-async-wgrad --no-kernel-timer --batch 8 --height 448 --width 1024 --steps 2 --warmup 1 > /dev/null 2>> $OUT/${TAG}_bench.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw2 -o pw2 -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-async-wgrad --no-kernel-timer --batch 8 --height 448 --width 1024 --steps 2 --warmup 1 > /dev/null 2>> $OUT/${TAG}_bench.err
F=$(find $OUT/pf2 -name "*.db" | head -1); W=$(find $OUT/pw2 -name "*.db" | head -1)
python tools/rocpd_traffic.py $F $W > $OUT/${TAG}_hbm_traffic_448x1024.txt
python tools/rocpd_traffic_json.py $F $W > $OUT/hbm_traffic_448x1024.json
rm -rf $OUT/pf2 $OUT/pw2
cp $OUT/hbm_traffic_448x1024.json profiles/hbm_traffic_448x1024.json
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --no-async-wgrad --no-cpu-baseline --no-secondary --no-extra-legs > $OUT/${TAG}_bench_serial.json 2>> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline --no-secondary --no-extra-legs --steps 5 > $OUT/${TAG}_bench_under_rocprof.json 2>> $OUT/${TAG}_bench.err
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) 50 > $OUT/${TAG}_kernel_stats.txt
python tools/rocpd_timeline.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/${TAG}_timeline_two_lane.txt 2>&1
rm -rf $OUT/kt
rocprofv3 --kernel-trace --stats -d $OUT/kts -o kts -- python3 bench.py --no-cpu-baseline --no-secondary --no-extra-legs --no-async-wgrad --steps 5 > $OUT/${TAG}_bench_serial_under_rocprof.json 2>> $OUT/${TAG}_bench.err
python tools/rocpd_stats.py $(find $OUT/kts -name "*.db" | head -1) 50 > $OUT/${TAG}_kernel_stats_serial.txt
python tools/rocpd_timeline.py $(find $OUT/kts -name "*.db" | head -1) > $OUT/${TAG}_timeline_serial.txt 2>&1
rm -rf $OUT/kts
IRR_CONV_MATH=f32 python bench.py --no-cpu-baseline --no-secondary --no-extra-legs > $OUT/${TAG}_bench_math_f32.json 2>> $OUT/${TAG}_bench.err
IRR_CONV_MATH=x3 python bench.py --no-cpu-baseline --no-secondary --no-extra-legs > $OUT/${TAG}_bench_math_x3.json 2>> $OUT/${TAG}_bench.err
IRR_X3S_H2=0 python bench.py --no-cpu-baseline --no-secondary --no-extra-legs > $OUT/${TAG}_bench_x3s_bf16x3.json 2>> $OUT/${TAG}_bench.err      # (round 5: the fp16x2 form is the default; this leg = round 4's routing)
IRR_LANE_MAX_LEAD=1 python bench.py --no-cpu-baseline --no-secondary --no-extra-legs > $OUT/${TAG}_bench_bounded_lead.json 2>> $OUT/${TAG}_bench.err
# 4. the second crop of north_star (per-GPU share of configs[4]): kernel stats + PMC passes of its own
python bench.py --no-cpu-baseline --no-extra-legs --batch 8 --height 448 --width 1024 > $OUT/${TAG}_bench_448x1024_bs8.json 2>> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt2 -o kt2 -- python3 bench.py --no-cpu-baseline --no-extra-legs --batch 8 --height 448 --width 1024 --steps 5 > $OUT/${TAG}_bench_448x1024_under_rocprof.json 2>> $OUT/${TAG}_bench.err
python tools/rocpd_stats.py $(find $OUT/kt2 -name "*.db" | head -1) 50 > $OUT/${TAG}_kernel_stats_448x1024.txt
rm -rf $OUT/kt2
[ -n "$QUICK" ] && { python tools/lane_race_probe.py 12 2>/dev/null > $OUT/${TAG}_lane_probe.txt; python tools/level_times.py 2>/dev/null > $OUT/${TAG}_level_times.txt; ls -la $OUT; exit 0; }   # QUICK=1: default-config numbers only
bash tools/pmc_ops.sh ${TAG} > /dev/null 2>&1
# 5. microbenches of the kernels north_star names + the same-box A/B against the round-3 tree (if r3tree/ is there)
python tools/corr_bench.py 2>/dev/null > $OUT/${TAG}_corr_bench_run.txt
python tools/warp_bench.py 2>/dev/null > $OUT/${TAG}_warp_bench.txt
python tools/wx3_check.py 2>/dev/null > $OUT/${TAG}_wgrad_x3_microbench.txt
python tools/h2_check.py 2>/dev/null > $OUT/${TAG}_h2_check.txt
python tools/x3s_h2_check.py 2>/dev/null > $OUT/${TAG}_x3s_h2_check.txt
python tools/lane_race_probe.py 12 2>/dev/null > $OUT/${TAG}_lane_probe.txt
IRR_CONV_MATH=x3 python tools/lane_race_probe.py 24 2>/dev/null > $OUT/${TAG}_lane_probe_x3.txt
python tools/pair_probe.py 2>/dev/null > $OUT/${TAG}_pair_probe.txt
python tools/truth_probe.py 2>/dev/null > $OUT/${TAG}_math_agreement.txt
python tools/level_times.py 2>/dev/null > $OUT/${TAG}_level_times.txt
python tools/conv_breakdown.py 32 --fp32 2>/dev/null > $OUT/${TAG}_conv_breakdown.txt
python tools/op_census.py 2>/dev/null > $OUT/${TAG}_op_census.txt
[ -d r4tree ] && bash tools/ab_r4.sh 3 > $OUT/${TAG}_ab_vs_r4.txt 2>&1
ls -la $OUT
