set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
for cfg in "12500 5000 8" "10000 2000 4"; do
  set -- $cfg
  rocprofv3 --kernel-trace -d $R/gpurun_out/tl/t$1 -o t --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --cells $1 --genes $2 --clones $3 > /dev/null 2>&1
  f=$(find $R/gpurun_out/tl/t$1 -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/timeline.py $f 60
  python3 $R/tools/gaps.py $f 2>/dev/null | tail -15
done
