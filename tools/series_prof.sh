#!/bin/bash
# kernel stats of the series path:  gpurun -- 'bash tools/series_prof.sh N G C [sweeps]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_series_$1_$4
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o st --output-format csv -- python3 $R/tools/series_prof.py $1 $2 $3 $4 > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(8), "us", r["Percentage"])
PY
