#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_dense.py -m gpu -q -x 2>&1 | tail -60 | tee gpurun_out/q_pytest.log
