#!/bin/bash
# every workload of bench.py in both storage types: cache-cold headline (8 input sets cycled) and the resident leg
for w in C2 C2p C3 C3p C3pp C5 C5p C5pp; do for dt in bf16 fp32; do
  echo -n "$w $dt : "; timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --workload $w --dtype $dt "$@" 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); r = d['roofline']
    print('Gpts/s', d['value'], 'us/step', round(1000 * d['ms_per_step'], 1), 'resident', round(1000 * (d.get('resident') or {}).get('ms_per_step', 0), 1), 'step frac', r['fwd_bwd']['frac'], {k: round(1000 * v['avg_ms'], 1) for k, v in r['kernels'].items()})
except Exception as e:
    print('FAILED', e)"
done; done
