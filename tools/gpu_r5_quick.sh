#!/bin/bash
# Round-5 iteration batch: parity tests (optionally a -k subset), then the C2 step with per-kernel times in a few
# configurations.   gpurun --timeout 1500 -- bash tools/gpu_r5_quick.sh [pytest -k expression | skip] [extra bench args]
mkdir -p gpurun_out
export TMPDIR=/tmp
sel=$1; shift
if [ "$sel" != "skip" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q ${sel:+-k "$sel"} 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
fi
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
echo "== bench" | tee gpurun_out/r5_quick.log
run bf16 "$@" | tee -a gpurun_out/r5_quick.log
run bf16 --opt 21=1 "$@" | tee -a gpurun_out/r5_quick.log
for u in 1 2 3 4 6 8; do run bf16 --opt 22=$u "$@" | tee -a gpurun_out/r5_quick.log; done
run bf16 --entry function "$@" | tee -a gpurun_out/r5_quick.log
run bf16 --workload C2p "$@" | tee -a gpurun_out/r5_quick.log
run bf16 --inputs test "$@" | tee -a gpurun_out/r5_quick.log
run fp32 "$@" | tee -a gpurun_out/r5_quick.log
