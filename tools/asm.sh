#!/bin/bash
# Compile one kernel file of the product to a device-assembly listing and audit its vector-memory waits.
# usage: tools/asm.sh ss_mfcc512 [name-filter] [extra hipcc flags]   -> /tmp/dis/<file>.s
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
F=$1; shift; FLT=${1:-}; [ $# -gt 0 ] && shift
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$R/include" -I"$R/mfcc-rust_amd/csrc" -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
  -S --cuda-device-only "$R/mfcc-rust_amd/csrc/$F.hip" -o /tmp/dis/$F.s 2>&1 | grep -v "hip-link" || true
python3 "$R/tools/vmcnt_audit.py" /tmp/dis/$F.s $FLT | cut -c1-230
python3 - "$F" "$FLT" <<'PY'
import re, sys
f, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
txt = open(f"/tmp/dis/{f}.s").read()
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt):
    if flt in m.group(1):
        print(f"{m.group(1)[:100]}  scratch {m.group(2)}  sgpr {m.group(3)}  vgpr {m.group(4)}")
PY
