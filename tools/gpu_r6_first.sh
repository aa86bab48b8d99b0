export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/pytest_gpu_all.log
run() {
  timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    print('%-30s %s cold %.4f ms resident %s | '%(' '.join(sys.argv[1:]), d['dtype'], d['ms_per_step'], (d.get('resident') or {}).get('ms_per_step')) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
for r in 1 2; do
echo "rec8"; run; run --workload C2p
echo "rec16"; BOXATTN_HIP_LIB=boxer_amd/variants/libboxattn_rec16.so run; BOXATTN_HIP_LIB=boxer_amd/variants/libboxattn_rec16.so run --workload C2p
done | tee gpurun_out/r6_rec8.log
