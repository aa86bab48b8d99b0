timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for dt in bf16 fp32; do for v in 0 4 6; do
echo -n "$dt variant $v: "; timeout 300 python bench.py --steps 100 --warmup 20 --variant $v --dtype $dt --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], {k: round(1e3*v['avg_ms'],1) for k,v in r['kernels'].items()})"
done; done
