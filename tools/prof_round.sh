#!/bin/bash
# Profiles of the default bench command for profiles/: kernel stats, then separate PMC passes (MI355X_MICROARCH.md: never
# combine --pmc with trace domains other than --kernel-trace).  Run from the repo root on the GPU box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st --output-format csv -- python3 $R/bench.py --no-cpu-baseline --busy-seconds 0 > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pf --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --busy-seconds 0 > $OUT/bench_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pw --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --busy-seconds 0 > $OUT/bench_pw.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_sq -o ps --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --busy-seconds 0 > $OUT/bench_ps.log 2>&1
ls -R $OUT | head -40
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
# (the other single-GPU BASELINE configurations and the VALU fallback shapes are in the default bench line itself since round 5: other_configs / fallbacks)
