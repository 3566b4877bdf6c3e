#!/bin/bash
# gaps of one steady-state iteration under engine variants: tools/tl_var.sh "<bench args>" ...   (run from the repo root on the GPU box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
i=0
for a in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/tl/v$i
  rocprofv3 --kernel-trace -d $R/gpurun_out/tl/v$i -o t --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline $a > /dev/null 2>&1
  f=$(find $R/gpurun_out/tl/v$i -name "*kernel_trace.csv" | head -1)
  echo "== $a"
  python3 $R/tools/timeline.py $f 60 | head -14
  python3 $R/tools/gaps.py $f 2>/dev/null | grep -E "mean [1-9]|mean 0\.[3-9]" | head -12
done
