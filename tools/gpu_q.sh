#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -k "fullsize or golden or vs_oracle or dense or sweep or linearity or nonfinite or outside" 2>&1 | tail -4
run() {  # dtype, extra args...
  local dt=$1; shift
  timeout 300 python bench.py --steps 300 --warmup 20 --dtype $dt --no-cpu-baseline --rotate 0 "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    print('%-34s %s %.4f Gpts/s %.4f ms | '%(' '.join(sys.argv[1:]), d['dtype'], d['value'], d['ms_per_step']) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
run fp32
run fp32 --opt 15=1
run fp32 --opt 19=2
run fp32 --workload C2p
run fp32 --workload C5p
run fp32 --workload C5p --opt 21=1
