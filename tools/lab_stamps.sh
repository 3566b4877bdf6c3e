#!/bin/bash
# builds the stamped library on the GPU box and prints the timeline of one merged forward launch:  tools/lab_stamps.sh "<extra -D flags>" [stamps.py args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="$1"; shift
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -fPIC -shared -I$ROOT/include -mllvm -amdgpu-mfma-vgpr-form \
  -DCA_BUILD_ID=\"lab_stamps\" -DCA_LAB $FLAGS -o /tmp/lab_stamps.so $ROOT/clonealign_amd/csrc/clonealign_hip.hip -ldl -pthread 2>/dev/null || { echo "build failed"; exit 1; }
CLONEALIGN_HIP_LIB=/tmp/lab_stamps.so python3 $ROOT/tools/stamps.py "$@"
