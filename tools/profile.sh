#!/bin/bash
# Usage (on the GPU box, from the repo root):  tools/profile.sh <tag> [bench.py args...]
# Collects, for the bench command: a kernel trace with stats, then PMC passes (each in its own run,
# --pmc never combined with other trace domains).  Outputs under gpurun_out/prof_<tag>/.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-secondary $@"
# the kernel trace runs bench.py's default step counts, so its average matches the bench line's HIP-event average
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/trace.log 2>&1
[ "$TRACE_ONLY" = 1 ] && { ls $OUT/trace/*/*kernel_stats.csv; exit 0; }
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" \
  "TCC_HIT TCC_MISS TCC_REQ" \
  "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
