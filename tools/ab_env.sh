#!/bin/bash
# A/B over debug-environment settings: tools/ab_env.sh "ENV=VAL ..." ...   (each argument: env assignments, then optional "-- bench args")
for a in "$@"; do
  envs="${a%%--*}"; args=""; [[ "$a" == *"--"* ]] && args="${a#*--}"
  ( for kv in $envs; do export "$kv"; done; export CLONEALIGN_DEBUG_ENV=1
    python3 bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline $args 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_iter_warmup']
print('%-44s %7.1f it/s %.4f ms  ' % ('$a', d['value'], d['ms_per_step']), {n: round(v*1e3) for n,v in k.items()})" )
done
