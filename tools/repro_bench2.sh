#!/bin/bash
# tools/repro_bench2.sh SCRIPT RUNS WORLD OUTDIR [bench args...] -- the 2-rank bench of tests/test_gpu_multi.py, RUNS times, every rank on
# device 0, each run's stdout/stderr kept; prints one line per run (exit code, seconds, the ranks' own messages)
script=$1; runs=$2; world=$3; out=$4; shift 4
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0 CLONEALIGN_BENCH_DEVICE=0
fail=0
for i in $(seq 1 "$runs"); do
  port=$((20000 + RANDOM % 20000))
  t0=$(date +%s.%N)
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node="$world" --master-addr 127.0.0.1 --master-port $port "$script" \
     --gpus "$world" --steps 6 --warmup 2 --repeats 2 --cells 20000 --genes 1000 --clones 4 --no-cpu-baseline --busy-seconds 0 "$@" \
     > "$out/run_$i.out" 2> "$out/run_$i.err"
  rc=$?
  t1=$(date +%s.%N)
  [ $rc -ne 0 ] && fail=$((fail+1))
  echo "run $i rc=$rc $(python3 -c "print(round($t1 - $t0, 1))") s  $(grep -h -E '^\[rank|EngineError|CA_ERR|SystemExit' "$out/run_$i.err" | grep -v elastic | head -3 | tr '\n' '|')"
done
echo "failed $fail of $runs"
