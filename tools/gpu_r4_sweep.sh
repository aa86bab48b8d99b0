#!/bin/bash
# A/B sweep of the rider geometry (boxattn_set_option(20): placement shifts | rider count) and the chunk size.
#   gpurun --timeout 1500 -- bash tools/gpu_r4_sweep.sh
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
for d in 1 2 3 4; do for sc in 1 2 3; do for sf in 1 2 3; do
  [ $sc != $sf ] && [ $sc != 1 ] && continue
  v=$(( sc | (sf << 4) | (d << 8) ))
  run bf16 --opt 20=$v | tee -a gpurun_out/r4_sweep.log
done; done; done
for c in 512 256 128; do
  run bf16 --opt 20=785 --opt 10=$c | tee -a gpurun_out/r4_sweep.log    # 785 = 1 | 1<<4 | 3<<8
done
run fp32 --opt 20=785 | tee -a gpurun_out/r4_sweep.log
run fp32 --opt 20=529 | tee -a gpurun_out/r4_sweep.log
