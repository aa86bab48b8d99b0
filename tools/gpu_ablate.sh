#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; print('ms/step', d['ms_per_step'], 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'K1', round(r['kernels']['bwd_ms'],4))"; }
run BOXATTN_DBG=0
run BOXATTN_DBG=15
run BOXATTN_DBG=1
run BOXATTN_DBG=0 BOXATTN_WGS=64
run BOXATTN_DBG=0 BOXATTN_WGS=128
run BOXATTN_DBG=0 BOXATTN_WGS=416
