#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
VARIANT_DTYPES="bf16" bash tools/gpu_variants.sh --workload C2 --rotate 0 --no-check --opt 15=1
VARIANT_DTYPES="bf16" bash tools/gpu_variants.sh --workload C2 --rotate 0 --no-check
done
