#!/bin/bash
# A/B of the libraries under boxer_amd/variants/: kernel times (tools/gpu_variants.sh), then the
# HBM traffic of each (tools/gpu_traffic.sh, serial schedule).
#   gpurun --timeout 1500 -- bash tools/gpu_ab_traffic.sh [bench args]
export TMPDIR=/tmp
mkdir -p gpurun_out
VARIANT_DTYPES=${VARIANT_DTYPES:-bf16} bash tools/gpu_variants.sh "$@"
VARIANT_DTYPES=${VARIANT_DTYPES:-bf16} bash tools/gpu_variants.sh "$@" > /dev/null; cat gpurun_out/variants.log
for lib in boxer_amd/variants/libboxattn_*.so; do
  name=$(basename $lib .so); name=${name#libboxattn_}
  export BOXATTN_HIP_LIB=$GRAFT_REPO_ROOT/$lib
  bash tools/gpu_traffic.sh ab_$name --variant 4 "$@" 2>&1 | grep -i "accumulate\|fwd2\|pointgrad"
  cd $GRAFT_REPO_ROOT
  rm -rf gpurun_out/pmc_ab_${name}_FETCH_SIZE gpurun_out/pmc_ab_${name}_WRITE_SIZE
done
