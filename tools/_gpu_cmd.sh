export CLONEALIGN_DEBUG_ENV=1
for shape in "12500 5000 8" "10000 2000 4" "25000 5000 8" "100000 5000 8"; do
  for w in 0 -1 16 0 -1; do
    echo -n "== $shape warm=$w  "; CA_FWD_WARM=$w python3 tools/lab_time.py $shape 2>&1 | grep -v amdgpu | tail -1
  done
done
