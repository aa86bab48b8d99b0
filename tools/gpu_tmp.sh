timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for w in C3 C3p; do for v in 0 5; do for dt in fp32 bf16; do
echo -n "$w variant $v $dt: "; timeout 300 python bench.py --steps 40 --warmup 10 --workload $w --variant $v --dtype $dt --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); r=d['roofline']
    print(d['value'], d['ms_per_step'], 'fwd_ms', r['fwd_ms'], 'bwd_ms', r['bwd_ms'])
except Exception as e: print('FAILED', e)"
done; done; done
