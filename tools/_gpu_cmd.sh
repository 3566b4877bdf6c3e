python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
python3 tools/s2_time.py 2>&1 | grep -v amdgpu | tail -3
python3 tools/s2_time.py 12500 5000 8 2 2>&1 | grep -v amdgpu | tail -3
python3 tools/s2_time.py 10000 2000 4 2 2>&1 | grep -v amdgpu | tail -3
bash tools/small_tl.sh 2>&1 | grep -A4 "iteration length"
