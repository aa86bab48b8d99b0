#!/bin/bash
# Instance attention float32: split-MFMA accumulate (default) against the VALU list walk (--opt 19=1), same box.
#   gpurun --timeout 1200 -- bash tools/gpu_inst_f32_ab.sh
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q -k "instance or inst or C3" 2>&1 | tail -8 | tee gpurun_out/inst_f32_pytest.log
run() {
  timeout 300 python bench.py --steps 800 --warmup 30 --dtype fp32 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    rot=d.get('resident') or {}
    print('%-40s %s %.4f Gpts/s %.4f ms resident %s | '%(' '.join(sys.argv[1:]), d['dtype'], d['value'], d['ms_per_step'], rot.get('ms_per_step')) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
: > gpurun_out/inst_f32_ab.log
for rep in 1 2; do
  for wl in C3 C3p; do
    run --workload $wl | tee -a gpurun_out/inst_f32_ab.log
    run --workload $wl --opt 19=1 | tee -a gpurun_out/inst_f32_ab.log
  done
done
