#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -2; done
timeout 900 python tools/gpu_soak_step.py 3000 2>&1 | tail -11
