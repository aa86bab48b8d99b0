timeout 900 python -m pytest tests -m gpu -x -q -k "sweep" 2>&1 | tail -30
