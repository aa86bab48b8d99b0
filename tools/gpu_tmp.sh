timeout 900 python -m pytest tests -m gpu -x -q -k "grid or modules" 2>&1 | tail -5
python tools/gpu_grid_bench.py 2 2>&1 | tail -3
python tools/gpu_grid_bench.py 16 2>&1 | tail -3
