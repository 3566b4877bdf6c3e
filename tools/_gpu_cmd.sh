rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
python -m pytest tests/test_gpu_parity.py -x -q -k "queued_ahead or merged_update" 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
for cfg in "100000 5000 8" "12500 5000 8" "10000 2000 4"; do
  echo "== gated $cfg"; python tools/fit_time.py $cfg 2>&1 | grep "caller" | tail -1
  echo "== lock-step $cfg"; CLONEALIGN_DEBUG_ENV=1 CA_RUN_GATE=0 python tools/fit_time.py $cfg 2>&1 | grep "caller" | tail -1
  echo "== ca_iterate $cfg"; python tools/lab_time.py $cfg 2>&1 | tail -1
done
