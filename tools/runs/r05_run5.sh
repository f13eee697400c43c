cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
for v in nozskip zskip; do
  export SS_LIB_PATH=$R/ab/lib_$v.so
  OUT=$R/gpurun_out/prof_r05_$v; mkdir -p $OUT
  i=0
  for set in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
    "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TA_BUSY" \
    "TCC_HIT TCC_MISS TCC_REQ"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --workload cfg3 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $OUT > $R/gpurun_out/r05/pmc_cfg3_$v.txt 2>&1
  rm -rf $OUT
done
cd $R
paste gpurun_out/r05/pmc_cfg3_nozskip.txt gpurun_out/r05/pmc_cfg3_zskip.txt | cut -c1-200
