python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "mc_samples or s2 or k2p1s2x" 2>&1 | tail -3
python3 tools/s2_time.py 2>&1 | grep -v amdgpu | tail -3 | head -2
python3 tools/s2_time.py 12500 5000 8 2 2>&1 | grep -v amdgpu | tail -3 | head -2
python3 tools/s2_time.py 50000 5000 8 2 2>&1 | grep -v amdgpu | tail -3 | head -2
python3 tools/s2_time.py 10000 2000 4 2 2>&1 | grep -v amdgpu | tail -3 | head -2
