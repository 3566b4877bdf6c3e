rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
python tools/repro_flake3.py 60 2>&1 | tail -12
