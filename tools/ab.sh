#!/bin/bash
# Build HEAD (ab/lib_prev.so) and the working tree (ab/lib_new.so), then run the interleaved A/B on one GPU box.
# usage: tools/ab.sh [rounds] [extra bench.py args]
set -e
cd "$(dirname "$0")/.."
R=${1:-3}; shift || true
make -C mfcc-rust_amd/csrc 2>&1 | grep -E "error|warning" || true
mkdir -p ab && cp mfcc-rust_amd/lib/libspeechsauce_amd.so ab/lib_new.so
rm -rf /tmp/old/repo && mkdir -p /tmp/old/repo && git archive HEAD | tar -x -C /tmp/old/repo
sed -i "s#ROOT    := .*#ROOT    := /tmp/old/repo#" /tmp/old/repo/mfcc-rust_amd/csrc/Makefile
make -C /tmp/old/repo/mfcc-rust_amd/csrc 2>&1 | grep -E "error" || true
cp /tmp/old/repo/mfcc-rust_amd/lib/libspeechsauce_amd.so ab/lib_prev.so
gpurun --timeout 900 -- "timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; for w in ${WL:-cfg2}; do tools/ab_bench.sh \$PWD/ab/lib_prev.so \$PWD/ab/lib_new.so $R --workload \$w $*; done" 2>&1 | grep -E "passed|failed|lib_|error"
