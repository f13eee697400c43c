cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -2
bash tools/ablate_run.sh "head b128" 10 --workload cfg2 2>&1 | tee gpurun_out/r04/ab_cfg2_tight_b128.txt | awk '{k=$2" "$3; a[k]+=$4; n[k]++} END{for(k in a) print k, a[k]/n[k], n[k]}'
cd /tmp && export TMPDIR=/tmp
for v in b128; do
  export SS_LIB_PATH=$R/ab/lib_$v.so
  OUT=$R/gpurun_out/pmcx_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/pmc1.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $OUT 2>&1 | grep -A6 "mfcc_c256" | head -8
  rm -rf $OUT
done
