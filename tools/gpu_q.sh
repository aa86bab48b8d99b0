#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
VARIANT_DTYPES="bf16 fp32" bash tools/gpu_variants.sh --workload C2 --rotate 0 --steps 1500 --warmup 100
cp gpurun_out/variants.log gpurun_out/variants_c2_$i.log
done
VARIANT_DTYPES="bf16 fp32" bash tools/gpu_variants.sh --workload C2p --rotate 0 --steps 1000 --warmup 100
