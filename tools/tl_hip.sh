#!/bin/bash
# kernel trace + HIP API trace of a short bench run, to see how far ahead of the GPU the host's launch calls are:
#   tools/tl_hip.sh <tag> <bench args...>      (run from the repo root on the GPU box; output under gpurun_out/tl/<tag>)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
mkdir -p $R/gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl/$tag
rocprofv3 --kernel-trace --hip-runtime-trace -d $R/gpurun_out/tl/$tag -o t --output-format csv -- python3 $R/bench.py --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 $R/tools/host_ahead.py $R/gpurun_out/tl/$tag
