#!/bin/bash
# Timing builds of the engine (extra -D flags; results may be WRONG on purpose) benchmarked beside the tree's build, on the GPU box:
#   tools/lab_ab.sh "<bench args>" name1="-DFLAG=1 ..." name2="..." ...
# Each variant is compiled to /tmp/lab_<name>.so and run through CLONEALIGN_HIP_LIB; the first line is the tree's own library.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ARGS="$1"; shift
run() {
  python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --busy-seconds 0 --allow-foreign-lib $ARGS 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_iter_warmup']
print('%-28s %7.1f it/s %.4f ms  ' % ('$1', d['value'], d['ms_per_step']), {n: round(v*1e3) for n,v in k.items()})"
}
run tree
for v in "$@"; do
  name="${v%%=*}"; flags="${v#*=}"
  /opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -fPIC -shared -I$ROOT/include -mllvm -amdgpu-mfma-vgpr-form \
    -DCA_BUILD_ID=\"lab_$name\" $flags -o /tmp/lab_$name.so $ROOT/clonealign_amd/csrc/clonealign_hip.hip -ldl -pthread 2>/dev/null || { echo "$name: build failed"; continue; }
  CLONEALIGN_HIP_LIB=/tmp/lab_$name.so run "$name"
done
