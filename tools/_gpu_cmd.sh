rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|^E  " | tail -8
