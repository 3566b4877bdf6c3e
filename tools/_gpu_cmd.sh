for shape in "100000 5000 8" "12500 5000 8" "10000 2000 4"; do
  python3 tools/fit_time.py $shape 2>&1 | grep -v amdgpu | grep "caller's eps\|steady" | tail -2
  CLONEALIGN_DEBUG_ENV=1 CA_RUN_GATE=0 python3 tools/fit_time.py $shape 2>&1 | grep -v amdgpu | grep "steady" | tail -1
  python3 tools/lab_time.py $shape 2>&1 | grep -v amdgpu | tail -1
done
