#!/bin/bash
# Build tuning variants of the HIP library: tools/build_variants.sh name1 "-DFLAG=1 ..." name2 "..."
# -> boxer_amd/variants/libboxattn_<name>.so (git-ignored; travels to the GPU box with gpurun)
cd "$(dirname "$0")/.."
mkdir -p boxer_amd/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift; shift
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16 $flags \
      -o boxer_amd/variants/libboxattn_$name.so boxer_amd/csrc/boxattn_capi.hip && echo "built $name" ) &
  # at most 4 compilers at a time
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
