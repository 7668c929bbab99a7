# on the GPU box: kernel sequences of truncated passes -> gpurun_out/seq/
export TMPDIR=/tmp
O=gpurun_out/seq; mkdir -p $O
for K in ${LEVELS:-1 2}; do
rocprofv3 --kernel-trace -d $O/k$K -o t -- python3 tools/fwd_levels_trace.py $K > $O/log$K.txt 2>&1
python tools/rocpd_seq.py $(find $O/k$K -name "*.db" | head -1) > $O/seq${BWD:+_bwd}_levels0to$K.txt 2>&1
rm -rf $O/k$K
done
