for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; done
for dt in bf16 fp32; do
echo -n "$dt: "; timeout 300 python bench.py --dtype $dt --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], {k: round(1e3*v['avg_ms'],1) for k,v in r['kernels'].items()})"
done
for w in C3 C3pp C5pp; do echo -n "$w: "; timeout 300 python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], {k: round(1e3*v['avg_ms'],1) for k,v in r['kernels'].items()})"
done
