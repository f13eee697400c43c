#!/bin/bash
# Builds variants of one kernel file into ab/lib_<name>.so for a same-box timing comparison (tools/ablate_run.sh).
# usage: [SRC=ss_mfcc512] tools/ablate.sh name1="-DSS_ABLATE=1" name2="-DSS_OPT=3 -fno-signed-zeros" ...
# SS_ABLATE bits remove a stage (timing attribution, results wrong by design); SS_OPT bits are experiments with correct results.
set -e
cd "$(dirname "$0")/.."
make -C mfcc-rust_amd/csrc -j8 2>&1 | grep -E "error" || true
mkdir -p ab
L=mfcc-rust_amd/lib
SRC=${SRC:-ss_mfcc512}
for kv in "$@"; do
  n=${kv%%=*}; fl=${kv#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Imfcc-rust_amd/csrc -fno-slp-vectorize $fl \
     -c mfcc-rust_amd/csrc/$SRC.hip -o ab/var_$n.o &
done
wait
for kv in "$@"; do
  n=${kv%%=*}
  OBJS=$(ls $L/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ab/lib_$n.so $OBJS ab/var_$n.o -Wl,-rpath,/opt/rocm/lib
done
ls ab/*.so
