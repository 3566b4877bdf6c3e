python3 tools/fuzz_parity.py 300 901 2>&1 | grep -v amdgpu | tail -6
python3 tools/fuzz_sharded.py 30 2>&1 | grep -v amdgpu | tail -3
python3 tools/fuzz_large.py 6 29 2>&1 | grep -v amdgpu | tail -3
