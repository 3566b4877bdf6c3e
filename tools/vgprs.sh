#!/bin/bash
# register / LDS / scratch use per kernel of the engine library (device-only assembly of the same sources and flags):
#   tools/vgprs.sh [name-filter]        -> build_ab/isa/ca.s + a table
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build_ab/isa
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 --offload-device-only -S -I$ROOT/include -mllvm -amdgpu-mfma-vgpr-form \
  -DCA_BUILD_ID=\"asm\" $EXTRA -o $ROOT/build_ab/isa/ca.s $ROOT/clonealign_amd/csrc/clonealign_hip.hip 2>/dev/null || exit 1
python3 - "$ROOT/build_ab/isa/ca.s" "${1:-}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2]
# amdhsa.kernels metadata: one block per kernel
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
    ag, lds, name, scratch, sg, vg = m.groups()
    if flt and flt not in name: continue
    print(f"{name[:110]:110s} vgpr {vg:>4s} agpr {ag:>3s} sgpr {sg:>4s} lds {lds:>6s} scratch {scratch}")
PY
