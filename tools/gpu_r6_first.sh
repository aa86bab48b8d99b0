export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > gpurun_out/smoke.log
bash tools/gpu_workloads.sh > gpurun_out/r6_workloads.log 2>&1
