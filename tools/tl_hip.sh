#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl/h1
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace -d $R/gpurun_out/tl/h1 -o t --output-format csv -- python3 $R/bench.py --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline --cells 12500 > /dev/null 2>&1
ls -la $R/gpurun_out/tl/h1/*
