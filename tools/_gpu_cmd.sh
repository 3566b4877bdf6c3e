python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu -k "queued_ahead or multifit or cancelled" 2>&1 | tail -3
for shape in "12500 5000 8" "10000 2000 4" "100000 5000 8"; do
  python3 tools/fit_time.py $shape 2>&1 | grep -v amdgpu | grep "steady" | tail -1
  FIT_VARIANT_ON=run_fwd python3 tools/fit_time.py $shape 2>&1 | grep -v amdgpu | grep "steady" | tail -1
done
