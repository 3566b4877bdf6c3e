python tools/corun.py 420 6144 &
sleep 8
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_boundary.py tests/test_gpu_sharding.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|^E  |^FAILED" | tail -10
kill %1 2>/dev/null; wait
