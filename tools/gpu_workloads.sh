#!/bin/bash
for w in C2 C2p C3 C3p C3pp C5 C5p C5pp; do for dt in bf16 fp32; do
  echo -n "$w $dt : "; timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --rotate 0 --workload $w --dtype $dt 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); r = d['roofline']
    print('Gpts/s', d['value'], 'ms/step', d['ms_per_step'], 'step frac', r['fwd_bwd']['frac'], {k: v['avg_ms'] for k, v in r['kernels'].items()})
except Exception as e:
    print('FAILED', e)"
done; done
