python tools/fuzz_parity.py 300 501 2>&1 | grep -v amdgpu > gpurun_out/r4/fuzz_parity_300_501.log; grep -E "FAIL|ERROR|agree" gpurun_out/r4/fuzz_parity_300_501.log | cut -c1-400
python tools/fuzz_sharded.py 40 2>&1 | grep -v amdgpu > gpurun_out/r4/fuzz_sharded_40b.log; tail -1 gpurun_out/r4/fuzz_sharded_40b.log
python tools/fuzz_large.py 6 23 2>&1 | grep -v amdgpu > gpurun_out/r4/fuzz_large_6_23.log; tail -1 gpurun_out/r4/fuzz_large_6_23.log
