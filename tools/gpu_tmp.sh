for pad in 0 16000 30000 43000 70000; do
echo -n "pad $pad: "; BOXATTN_EXP_LDS_PAD=$pad timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('fwd %.1f us' % (1e3*k['fwd']['avg_ms']))"
done
