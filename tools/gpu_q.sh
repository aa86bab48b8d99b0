#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
bash tools/gpu_train_steps.sh 2>&1 | tail -12
