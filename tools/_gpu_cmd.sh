python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
