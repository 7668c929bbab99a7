#!/bin/bash
# PMC passes over tools/pmc_run.py on the GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA-pipe
# utilisation of the heavy conv shapes + cost volume / warp at 96x112x64.   bash tools/pmc_ops.sh <tag>
TAG=${1:-r3}
OUT=gpurun_out/prof
mkdir -p $OUT
cd /tmp 2>/dev/null; cd - > /dev/null
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf -- python3 tools/pmc_run.py > /dev/null 2> $OUT/${TAG}_pmc.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw -- python3 tools/pmc_run.py > /dev/null 2>> $OUT/${TAG}_pmc.err
F=$(find $OUT/pf -name "*.db" | head -1); W=$(find $OUT/pw -name "*.db" | head -1)
python tools/rocpd_traffic.py $F $W > $OUT/${TAG}_traffic_ops.txt
rm -rf $OUT/pf $OUT/pw
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $OUT/pm -o pm -- python3 tools/pmc_run.py > /dev/null 2>> $OUT/${TAG}_pmc.err
python tools/rocpd_pmc.py $(find $OUT/pm -name "*.db" | head -1) > $OUT/${TAG}_pmc_x3.txt
rm -rf $OUT/pm
cat $OUT/${TAG}_traffic_ops.txt
