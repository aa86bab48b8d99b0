#!/bin/bash
# Round-3 iteration batch: dense-kernel parity tests, then C2 bf16 kernel times (bench line + rocprofv3 stats).
#   gpurun --timeout 1200 -- bash tools/gpu_dense.sh [pytest -k expression]
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest dense"; timeout 900 python -m pytest tests/test_gpu_dense.py -m gpu -x -q ${1:+-k "$1"} 2>&1 | tail -15 | tee gpurun_out/pytest_dense.log
for inp in model test; do
  echo "== bench bf16 $inp"
  timeout 300 python bench.py --steps 200 --warmup 20 --dtype bf16 --inputs $inp --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config'].get('parity_gate'), json.dumps(d['roofline'].get('kernels')))"
done
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dense -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --dtype bf16 --no-cpu-baseline --no-check > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_dense -name "*.db" | head -1) | head -14 | tee gpurun_out/kernel_stats_dense.txt
rm -rf gpurun_out/prof_dense
