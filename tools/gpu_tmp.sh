timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for w in C3 C3p; do for dt in fp32 bf16; do
echo -n "$w $dt: "; timeout 300 python bench.py --workload $w --dtype $dt --variant 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], {k: round(1e3*v['avg_ms'],1) for k,v in r['kernels'].items()})"
done; done
