#!/bin/bash
# Build tuning variants of the HIP library: tools/build_variants.sh name1 "-DFLAG=1 ..." name2 "..."
# -> boxer_amd/variants/libboxattn_<name>.so (git-ignored; travels to the GPU box with gpurun)
cd "$(dirname "$0")/.."
mkdir -p boxer_amd/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift; shift
  ( python -m boxer_amd._lib --out boxer_amd/variants/libboxattn_$name.so $flags > /dev/null && echo "built $name" ) &
  # at most 3 variants (two compilers each) at a time
  while [ $(jobs -r | wc -l) -ge 3 ]; do sleep 1; done
done
wait
