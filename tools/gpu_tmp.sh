VARIANT_DTYPES=bf16 bash tools/gpu_variants.sh
