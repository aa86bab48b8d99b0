#!/bin/bash
# Decoder-sized shapes (C3'', C5'', C3): default against riders off (--opt 15=1) and, float32, the VALU accumulate (--opt 19=1).
#   gpurun --timeout 1200 -- bash tools/gpu_small_shapes_ab.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 1500 --warmup 50 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    rot=d.get('resident') or {}
    print('%-50s %s %.4f ms resident %s | '%(' '.join(sys.argv[1:]), d['dtype'], d['ms_per_step'], rot.get('ms_per_step')) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
for wl in C3pp C5pp C3; do
  run --workload $wl --dtype bf16
  run --workload $wl --dtype bf16 --opt 15=1
  run --workload $wl --dtype fp32
  run --workload $wl --dtype fp32 --opt 15=1
  run --workload $wl --dtype fp32 --opt 19=1
done 2>&1 | tee gpurun_out/small_ab.log
