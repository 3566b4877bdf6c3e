mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py -x -q -k "merged_update" > gpurun_out/r4/merge_tests4.txt 2>&1; tail -4 gpurun_out/r4/merge_tests4.txt
for cfg in "12500 5000 8" "10000 2000 4" "100000 5000 8" "50000 5000 8"; do
  for rep in 1 2; do
  echo "== $cfg merged"; python tools/lab_time.py $cfg 2>&1 | tail -1
  echo "== $cfg two-launch"; CLONEALIGN_DEBUG_ENV=1 CA_UPDATE_MERGE=0 python tools/lab_time.py $cfg 2>&1 | tail -1
  done
done > gpurun_out/r4/merge_time4.txt 2>&1
cat gpurun_out/r4/merge_time4.txt
