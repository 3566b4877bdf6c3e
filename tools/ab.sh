#!/bin/bash
# A/B of one engine switch on the bench shapes: tools/ab.sh VAR=VALUE [cells ...]   (against the default, interleaved twice)
KV=$1; shift
for c in "$@"; do
  for rep in 1 2; do
    for mode in base alt; do
      if [ $mode = alt ]; then export "$KV"; else unset "${KV%%=*}"; fi
      python3 bench.py --cells $c --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', '$KV', d['config']['cells'], round(d['value'],1), 'it/s', round(d['ms_per_step']*1000,1), 'us')"
    done
  done
done
