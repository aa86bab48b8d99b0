#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # workload dtype opt
timeout 300 python bench.py --workload $1 --dtype $2 --steps 1500 --warmup 100 --rotate 0 --no-cpu-baseline --no-check --opt 20=$3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1 $2 opt20=$3', d['ms_per_step'], {k: v['avg_ms'] for k, v in r['kernels'].items()})
"; }
for o in 0 512 768 1024 1280 1536 2048; do run C2 fp32 $o; done
for o in 0 768 1280 1536; do run C2p fp32 $o; done
for o in 0 2 3; do run C2 fp32 $o; done
