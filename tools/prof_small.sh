#!/bin/bash
# Kernel-time breakdown of the per-rank share of a strong-scaling run (cells = 100000 / ranks).  Run on the GPU box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cells in "$@"; do
  OUT=$R/gpurun_out/small_$cells
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o st --output-format csv -- python3 $R/bench.py --cells $cells --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench.log 2>&1
  grep '^{' $OUT/bench.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cells $cells', round(d['value'],1), 'it/s', round(d['ms_per_step']*1000,1), 'us/iter')"
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("  %-48s %6s %9.1f us %5s%%" % (r["Name"][:48], r["Calls"], float(r["AverageNs"])/1000, r["Percentage"]))
PY
done
