#!/bin/bash
# A/B of engine variants on the default workload: tools/ab.sh "<bench args>" "<bench args>" ...   (prints it/s and the profiled per-class ms)
for a in "$@"; do
  python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --busy-seconds 0 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_iter_warmup']
print('%-40s %7.1f it/s %.4f ms  ' % ('$a', d['value'], d['ms_per_step']), {n: round(v*1e3) for n,v in k.items()})"
done
