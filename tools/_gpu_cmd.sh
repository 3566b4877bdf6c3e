mkdir -p gpurun_out/r4
for e in "" "CA_UPDATE_MERGE=0" "CA_Y_MFMA1=1,CA_Y_RIDE=1" "CA_BWD_MFMA=0" "CA_FWD_MFMA=0"; do
  echo "### extra env: $e"
  FUZZ_ONLY=256 FUZZ_ENV="$e" python tools/fuzz_parity.py 300 401 2>&1 | grep -v amdgpu.ids | tail -4
done > gpurun_out/r4/fuzz_replay_256.txt 2>&1
cat gpurun_out/r4/fuzz_replay_256.txt
