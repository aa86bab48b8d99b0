#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dense.py -m gpu -q -x -k "float32" 2>&1 | tail -15 | tee gpurun_out/q_pytest.log
run() { # workload dtype opt
timeout 300 python bench.py --workload $1 --dtype $2 --steps 1500 --warmup 100 --rotate 0 --no-cpu-baseline --opt 19=$3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1 $2 opt19=$3', d['ms_per_step'], {k: v['avg_ms'] for k, v in r['kernels'].items()})
    elif 'rror' in l or 'mismatch' in l: print(l.strip()[:300])
"; }
for w in C2 C2p C5p; do for v in 0 3 0 3; do run $w fp32 $v; done; done
