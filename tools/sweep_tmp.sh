timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "backward_sweep or fused_sweep or mid" 2>&1 | tail -2
for cs in default 51 102 204 306 391; do
  if [ $cs = default ]; then unset CA_CSPLIT_M; else export CA_CSPLIT_M=$cs; fi
  timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cs', round(d['value'],1), {k: round(v,4) for k,v in d['kernel_ms_per_iter_warmup'].items()})
"; done
