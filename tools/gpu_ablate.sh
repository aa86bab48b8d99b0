#!/bin/bash
run() { echo -n "$* : "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; print('Gpts/s', d['value'], 'ms/step', d['ms_per_step'], 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], {k: v['avg_ms'] for k, v in r['kernels'].items()})"; }
run --variant 0
run --variant 4
run --variant 0 --dtype fp32
run --variant 0 --inputs test
run --variant 0 --workload C2p
run --variant 0 --workload C3p --dtype fp32
