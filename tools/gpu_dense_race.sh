#!/bin/bash
# run-to-run check of the dense kernels for every library under boxer_amd/variants/
for lib in boxer_amd/variants/libboxattn_*.so; do
  echo "== $lib"
  BOXATTN_HIP_LIB=$PWD/$lib python tools/gpu_dense_debug.py 4lv_odd model 2>&1 | grep -v "amdgpu.ids"
done
