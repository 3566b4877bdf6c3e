python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "two_mc_samples" 2>&1 | tail -12
