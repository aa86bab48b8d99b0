#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; print('ms/step', d['ms_per_step'], {k: v['avg_ms'] for k, v in r['kernels'].items()})"; }
run BOXATTN_DBG=0
run BOXATTN_DBG=64
