#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "instance or inst or C3" 2>&1 | tail -4
for w in C3 C3p; do for dt in fp32 bf16; do
timeout 300 python bench.py --workload $w --dtype $dt --steps 2000 --warmup 100 --rotate 0 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$w $dt', d['ms_per_step'], {k: v['avg_ms'] for k, v in r['kernels'].items()})
"; done; done
