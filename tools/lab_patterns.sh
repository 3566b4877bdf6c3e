#!/bin/bash
# one timing build (extra -D flags), many dispatch patterns:  tools/lab_patterns.sh "<-D flags>" "<bench args>" pat1 pat2 ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="$1"; ARGS="$2"; shift; shift
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -fPIC -shared -I$ROOT/include -mllvm -amdgpu-mfma-vgpr-form \
  -DCA_BUILD_ID=\"lab_pat\" $FLAGS -o /tmp/lab_pat.so $ROOT/clonealign_amd/csrc/clonealign_hip.hip -ldl -pthread 2>/dev/null || { echo "build failed"; exit 1; }
for pat in "$@"; do
  CLONEALIGN_HIP_LIB=/tmp/lab_pat.so python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --busy-seconds 0 --allow-foreign-lib $ARGS --tune ride_pattern=$pat 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_iter_warmup']
print('%-14s %-10s %7.1f it/s %.4f ms  ' % ('$FLAGS', '$pat', d['value'], d['ms_per_step']), {n: round(v*1e3) for n,v in k.items()})"
done
