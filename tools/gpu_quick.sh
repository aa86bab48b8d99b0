#!/bin/bash
# Quick iteration batch: parity tests (-x), then serial-schedule kernel times for C2 bf16 / fp32.
#   gpurun --timeout 1200 -- bash tools/gpu_quick.sh [pytest -k expression]
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q ${1:+-k "$1"} 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
for dt in bf16 fp32; do
  echo "== bench $dt"
  timeout 300 python bench.py --steps 30 --warmup 5 --dtype $dt --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], json.dumps(d.get('kernels', d.get('roofline'))))"
done
for dt in bf16 fp32; do
  cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_q_$dt -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --dtype $dt --variant 4 --no-cpu-baseline > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/rocpd_stats.py $(find gpurun_out/prof_q_$dt -name "*.db" | head -1) | head -12 | tee gpurun_out/kernel_stats_q_$dt.txt
  rm -rf gpurun_out/prof_q_$dt
done
