#!/bin/bash
# Default workload at several batch sizes (images per GPU): value, ms/step, per-kernel us.
#   gpurun --timeout 900 -- bash tools/gpu_batch_sweep.sh [bench args]
export TMPDIR=/tmp
mkdir -p gpurun_out
for b in ${BATCHES:-1 2 8 16}; do
  timeout 300 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --batch $b "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('B=%-3s Gpts/s %7.3f  step %.4f ms |' % ('$b', d['value'], d['ms_per_step']), {a: round(1e3*v['avg_ms'],1) for a,v in k.items()})"
done | tee gpurun_out/batch_sweep.log
