#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; print('Gpts/s', d['value'], 'ms/step', d['ms_per_step'], 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'])"; }
run BOXATTN_SIDE_PRIO=-1
run BOXATTN_SIDE_PRIO=0
run BOXATTN_SIDE_PRIO=1
run BOXATTN_SIDE_PRIO=-1
run BOXATTN_SIDE_PRIO=0
run BOXATTN_SIDE_PRIO=1
