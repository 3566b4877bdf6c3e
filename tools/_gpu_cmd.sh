for cfg in "12500 5000 8" "25000 5000 8" "50000 5000 8"; do
  python tools/shard_seq_time.py $cfg 2>&1 | tail -1
  python tools/shard_seq_time.py $cfg --variant-off p2p_ride 2>&1 | tail -1
done
python -m pytest tests/test_gpu_multi.py tests/test_gpu_sharding.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
