#!/bin/bash
# On the GPU box: kernel trace + PMC passes for one tools/loop.py target.  usage: tools/prof_loop.sh <target> [tag]
T=$1; TAG=${2:-$1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/loop.py $T 200 > $OUT/trace.log 2>&1
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" \
  "TCC_HIT TCC_MISS TCC_REQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $R/tools/loop.py $T 60 > $OUT/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
