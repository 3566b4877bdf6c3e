bash tools/lab_stamps.sh "" --cells 12500 2>&1 | grep -v amdgpu | head -12
bash tools/lab_stamps.sh "" --cells 10000 --genes 2000 --clones 4 2>&1 | grep -v amdgpu | head -12
bash tools/lab_stamps.sh "" --cells 100000 2>&1 | grep -v amdgpu | head -12
