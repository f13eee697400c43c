#!/bin/bash
# Builds variants of one kernel file into ab/lib_<name>.so for a same-box timing comparison (tools/ablate_run.sh).
# usage: [SRC=ss_mfcc512] [BASE="-fno-slp-vectorize"] tools/ablate.sh name1="-DSS_ABLATE=1" name2="-DSS_X=3 -fno-signed-zeros" ...
# Variants are LAB builds of that file (-DSS_LAB=1: the stage-removal switches SS_ABLATE / SS_ABL5 and the experiment switch
# SS_X exist only there), linked with the LAB objects of every other file (the process-wide test aids the lab kernels consult
# exist in lab objects only); build a variant with no flags as the baseline of a comparison.  SS_ABLATE bits remove a stage (timing
# attribution, results wrong by design); SS_X bits are experiments with correct results.
set -e
cd "$(dirname "$0")/.."
make -C mfcc-rust_amd/csrc -j8 lab 2>&1 | grep -E "error" || true
mkdir -p ab
L=mfcc-rust_amd/lib/lab
SRC=${SRC:-ss_mfcc512}
BASE=${BASE--fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None}
for kv in "$@"; do
  n=${kv%%=*}; fl=${kv#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Imfcc-rust_amd/csrc -DSS_LAB=1 $BASE $fl \
     -c mfcc-rust_amd/csrc/$SRC.hip -o ab/var_$n.o &
done
wait
for kv in "$@"; do
  n=${kv%%=*}
  OBJS=$(ls $L/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ab/lib_$n.so $OBJS ab/var_$n.o -ldl -Wl,-rpath,/opt/rocm/lib
done
ls ab/*.so
