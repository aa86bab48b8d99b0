#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; print('ms/step', d['ms_per_step'], {k: v['avg_ms'] for k, v in r['kernels'].items()})"; }
run BOXATTN_CHUNK=1024 BOXATTN_WGS=1300
run BOXATTN_CHUNK=512 BOXATTN_WGS=2200
run BOXATTN_CHUNK=512 BOXATTN_WGS=832
run BOXATTN_CHUNK=384 BOXATTN_WGS=832
run BOXATTN_CHUNK=256 BOXATTN_WGS=3800
run BOXATTN_CHUNK=768 BOXATTN_WGS=1600
