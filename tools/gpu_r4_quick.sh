#!/bin/bash
# Round-4 iteration batch: parity tests, then the C2 step (both entries, riders on / off, both storage types),
# then rocprofv3 kernel stats of the default step.
#   gpurun --timeout 1500 -- bash tools/gpu_r4_quick.sh [pytest -k expression | skip]
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" != "skip" ]; then
  echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q ${1:+-k "$1"} 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
fi
run() {  # dtype, extra args...
  local dt=$1; shift
  timeout 300 python bench.py --steps 300 --warmup 20 --dtype $dt --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline'].get('kernels',{})
    print('%-34s %s %.4f Gpts/s %.4f ms | '%(' '.join(sys.argv[1:]), d['dtype'], d['value'], d['ms_per_step']) + ' '.join('%s=%.1f'%(n,v['avg_ms']*1e3) for n,v in k.items()))
except Exception as e: print('bench failed', sys.argv[1:], e)
" "$@"
}
echo "== bench" | tee gpurun_out/r4_quick.log
for dt in bf16 fp32; do
  run $dt | tee -a gpurun_out/r4_quick.log
  run $dt --opt 15=1 | tee -a gpurun_out/r4_quick.log
  run $dt --entry function | tee -a gpurun_out/r4_quick.log
  run $dt --entry function --graph | tee -a gpurun_out/r4_quick.log
done
run fp32 --opt 19=2 | tee -a gpurun_out/r4_quick.log
run bf16 --inputs test | tee -a gpurun_out/r4_quick.log
run bf16 --workload C2p | tee -a gpurun_out/r4_quick.log
run fp32 --workload C3 | tee -a gpurun_out/r4_quick.log
run fp32 --workload C5 | tee -a gpurun_out/r4_quick.log
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_q -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --dtype bf16 --no-cpu-baseline --preheat-s 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_q -name "*.db" | head -1) | head -12 | tee gpurun_out/kernel_stats_q_bf16.txt
rm -rf gpurun_out/prof_q
