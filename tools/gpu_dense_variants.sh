#!/bin/bash
# C2 bf16 kernel times of every library under boxer_amd/variants/ with the window-staged kernels on
export TMPDIR=/tmp
for lib in boxer_amd/variants/libboxattn_*.so; do
  name=$(basename $lib .so); name=${name#libboxattn_}
  for inp in ${VARIANT_INPUTS:-model}; do
  BOXATTN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 200 --warmup 20 --dtype bf16 --inputs $inp --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
    print('%-14s %-5s Gpts/s %7.3f  step %.4f ms | fwd %.1f  pts %.1f  acc %.1f  bin %.1f us | %s' % ('$name','$inp',d['value'],d['ms_per_step'],1e3*k['fwd']['avg_ms'],1e3*k['bwd_points']['avg_ms'],1e3*k['bwd_accumulate']['avg_ms'],1e3*k['bwd_binning']['avg_ms'], d['config'].get('parity_gate')))
except Exception as e: print('$name $inp FAILED', e)
"
  done
done | tee gpurun_out/variants_dense.log
