#!/bin/bash
# us per iteration (tools/lab_time.py) of the tree's library and of lab builds:  tools/lab_t.sh "<cells genes clones>" name1="-DFLAG" ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ARGS="$1"; shift
echo -n "tree: "; python3 $ROOT/tools/lab_time.py $ARGS 2>&1 | grep -v amdgpu | tail -1
for v in "$@"; do
  name="${v%%=*}"; flags="${v#*=}"
  /opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -fPIC -shared -I$ROOT/include -mllvm -amdgpu-mfma-vgpr-form \
    -DCA_BUILD_ID=\"lab_$name\" $flags -o /tmp/lab_$name.so $ROOT/clonealign_amd/csrc/clonealign_hip.hip -ldl -pthread 2>/dev/null || { echo "$name: build failed"; continue; }
  echo -n "$name: "; CLONEALIGN_HIP_LIB=/tmp/lab_$name.so python3 $ROOT/tools/lab_time.py $ARGS 2>&1 | grep -v amdgpu | tail -1
done
