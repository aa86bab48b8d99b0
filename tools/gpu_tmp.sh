for cap in 0 32 64 96 128 192 256; do
echo -n "cap $cap: "; BOXATTN_EXP_ACC_CAP=$cap timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'fwd_ms', r['fwd_ms'], 'bwd_ms', r['bwd_ms'])"
done
