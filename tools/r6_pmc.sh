#!/bin/bash
# Round-6 PMC passes over tools/pmc_wgrad.py on the GPU box: MFMA-pipe busy, issue stalls, LDS instruction /
# conflict / stall counters and the clock of the kernels on the default path.  Separate rocprofv3 runs per counter group (8 SQ
# slots per pass; --pmc never combined with a trace).   bash tools/r6_pmc.sh [tag]
TAG=${1:-r6}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/p_$name -o p -- python3 tools/pmc_wgrad.py > $OUT/${TAG}_pmc_$name.out 2> $OUT/${TAG}_pmc_$name.err
  local db=$(find $OUT/p_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_pmc.py $db conv_ > $OUT/${TAG}_pmc_$name.txt; else echo "no db for $name" > $OUT/${TAG}_pmc_$name.txt; tail -5 $OUT/${TAG}_pmc_$name.err >> $OUT/${TAG}_pmc_$name.txt; fi
  rm -rf $OUT/p_$name
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
python tools/r6_pmc_table.py $OUT/${TAG}_pmc_mfma.txt $OUT/${TAG}_pmc_lds.txt $OUT/${TAG}_pmc_wait.txt > $OUT/${TAG}_pmc_wgrad.txt 2>&1
cat $OUT/${TAG}_pmc_wgrad.txt
