#!/bin/bash
# Standard GPU-box batch: parity tests, smoke, bench lines.  Run through gpurun:
#   gpurun --timeout 1500 -- bash tools/gpu_check.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
echo "== pytest -m gpu" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee gpurun_out/smoke.log
echo "== bench"
for dt in bf16 fp32; do for inp in model test; do
  timeout 300 python bench.py --steps 30 --warmup 5 --dtype $dt --inputs $inp --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_matrix.log
done; done
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json
