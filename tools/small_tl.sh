R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5/small_tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/a -o a --output-format csv -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --busy-seconds 0 --cells 12500 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace -d $OUT/c -o c --output-format csv -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --busy-seconds 0 > $OUT/c.log 2>&1
rocprofv3 --kernel-trace -d $OUT/b -o b --output-format csv -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --busy-seconds 0 --cells 10000 --genes 2000 --clones 4 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace -d $OUT/d -o d --output-format csv -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --busy-seconds 0 --cells 25000 > $OUT/d.log 2>&1
rocprofv3 --kernel-trace -d $OUT/e -o e --output-format csv -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --busy-seconds 0 --cells 12288 > $OUT/e.log 2>&1
cd $R
for x in a e d b c; do f=$(find $OUT/$x -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py $f 400 > $OUT/tl_$x.txt 2>&1; rm -rf $OUT/$x; done
cat $OUT/tl_a.txt $OUT/tl_e.txt $OUT/tl_d.txt $OUT/tl_b.txt $OUT/tl_c.txt
