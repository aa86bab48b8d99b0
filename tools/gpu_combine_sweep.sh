export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    print('%-44s %s %.4f Gpts/s %.4f ms | '%(' '.join(sys.argv[1:]), d['dtype'], d['value'], d['ms_per_step']) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
for wl in C2 C2p C5p C3p; do for dt in bf16 fp32; do for o in 0 3; do
  run --workload $wl --dtype $dt --opt 15=$o
done; done; done 2>&1 | tee gpurun_out/combine_sweep.log
