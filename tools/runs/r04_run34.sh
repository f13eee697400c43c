cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for v in b0 b1 b2 b4 b8; do
  export SS_LIB_PATH=$R/ab/lib_$v.so
  OUT=$R/gpurun_out/pmcx_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/pmc1.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $OUT 2>&1 | grep -A6 "mfcc_c2048" | head -8
  rm -rf $OUT
done
