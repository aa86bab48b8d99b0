#!/bin/bash
run() { echo -n "$* : "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; print('ms/step', d['ms_per_step'], {k: v['avg_ms'] for k, v in r['kernels'].items()})"; }
run --variant 0
run --variant 10
run --variant 11
run --variant 0 --dtype fp32
run --variant 10 --dtype fp32
run --variant 11 --dtype fp32
