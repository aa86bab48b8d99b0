export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu_all.log
