#!/bin/bash
# Round-6 iteration batch: parity tests (optionally a -k subset, or "skip"), then the C2 step with per-kernel times.
#   gpurun --timeout 1500 -- bash tools/gpu_r6_quick.sh [pytest -k expression | skip | all] [extra bench args]
mkdir -p gpurun_out
export TMPDIR=/tmp
sel=$1; shift
if [ "$sel" != "skip" ]; then
  [ "$sel" = "all" ] && sel=""
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q ${sel:+-k "$sel"} 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
fi
run() {  # dtype, extra args...
  local dt=$1; shift
  timeout 300 python bench.py --steps 2000 --warmup 50 --dtype $dt --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    rot=d.get('rotated') or d.get('resident') or {}
    print('%-34s %s %.4f Gpts/s %.4f ms other-leg %s | '%(' '.join(sys.argv[1:]), d['dtype'], d['value'], d['ms_per_step'], rot.get('ms_per_step')) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
echo "== bench" | tee gpurun_out/r6_quick.log
run bf16 "$@" | tee -a gpurun_out/r6_quick.log
run bf16 --entry reference "$@" | tee -a gpurun_out/r6_quick.log
run bf16 --workload C2p "$@" | tee -a gpurun_out/r6_quick.log
run bf16 --inputs test "$@" | tee -a gpurun_out/r6_quick.log
run fp32 "$@" | tee -a gpurun_out/r6_quick.log
