for o in "15=0" "15=1"; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --opt $o 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('$o', d['value'], d['ms_per_step'], {n:round(v['avg_ms']*1e3,1) for n,v in k.items()})"
done
