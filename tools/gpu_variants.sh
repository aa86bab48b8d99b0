#!/bin/bash
# Time every library under boxer_amd/variants/ (serial schedule, per-kernel HIP-event times).
#   gpurun -- bash tools/gpu_variants.sh [bench args]     (--entry ops: the compiled module links the default library)
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in boxer_amd/variants/libboxattn_*.so; do
  name=$(basename $lib .so); name=${name#libboxattn_}
  for dt in ${VARIANT_DTYPES:-bf16 fp32}; do
    BOXATTN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --entry ops --steps 300 --warmup 20 --dtype $dt --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
    print('%-14s %-5s Gpts/s %7.3f  step %.4f ms | fwd %.1f  pts %.1f  acc %.1f  bin %.1f us' % ('$name','$dt',d['value'],d['ms_per_step'],1e3*k['fwd']['avg_ms'],1e3*k['bwd_points']['avg_ms'],1e3*k['bwd_accumulate']['avg_ms'],1e3*(k.get('bwd_binning',{}).get('avg_ms') or 0)))
except Exception as e: print('$name $dt FAILED', e)
"
  done
done | tee gpurun_out/variants.log
