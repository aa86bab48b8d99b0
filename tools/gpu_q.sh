#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_parity.py -m gpu -q -x -k "riders_match or rider_placement or match_oracle or algorithms or undersized or golden" 2>&1 | tail -4
for i in 1 2; do
VARIANT_DTYPES="bf16 fp32" bash tools/gpu_variants.sh --workload C2 --rotate 0 --no-check
done
VARIANT_DTYPES="bf16" bash tools/gpu_variants.sh --workload C2 --rotate 0 --no-check --opt 15=1
VARIANT_DTYPES="bf16" bash tools/gpu_variants.sh --workload C2p --rotate 0 --no-check
VARIANT_DTYPES="bf16" bash tools/gpu_variants.sh --workload C5p --rotate 0 --no-check
VARIANT_DTYPES="fp32" bash tools/gpu_variants.sh --workload C3 --rotate 0 --no-check
