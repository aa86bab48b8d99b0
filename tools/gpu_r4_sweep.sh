#!/bin/bash
# A/B sweep of the rider count (boxattn_set_option(20) = shifts | 64 v riders << 8) and the chunk size.
#   gpurun --timeout 1500 -- bash tools/gpu_r4_sweep.sh            (BOXATTN_HIP_LIB selects a build variant)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # dtype, extra args...
  local dt=$1; shift
  timeout 300 python bench.py --steps 300 --warmup 20 --dtype $dt --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    print('%-34s %s %.4f Gpts/s %.4f ms | '%(' '.join(sys.argv[1:]), d['dtype'], d['value'], d['ms_per_step']) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
: > gpurun_out/r4_sweep.log
for v in 3 4 5 6 8 10 16; do
  run bf16 --opt 20=$(( 1 | (1 << 4) | (v << 8) )) | tee -a gpurun_out/r4_sweep.log
done
for c in 2048 1536 768; do
  run bf16 --opt 10=$c | tee -a gpurun_out/r4_sweep.log
done
run bf16 --opt 15=3 | tee -a gpurun_out/r4_sweep.log
run fp32 | tee -a gpurun_out/r4_sweep.log
