#!/bin/bash
# Last call of a round: the full GPU test suite, smoke, the three bench lines and every workload on the final code
# -> gpurun_out/final/ (copied over the matching profiles/r0N_* files).
#   gpurun --timeout 2400 -- bash tools/gpu_final_check.sh
R=gpurun_out/final; rm -rf $R; mkdir -p $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > $R/smoke.log
timeout 600 python bench.py > $R/bench_default.json 2> $R/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > $R/bench_driver_flags.json 2>> $R/bench.err
timeout 300 python bench.py --entry function --no-cpu-baseline > $R/bench_function_entry.json 2>> $R/bench.err
bash tools/gpu_workloads.sh > $R/workloads.log 2>&1
cat $R/pytest_gpu.log $R/smoke.log; cut -c1-330 $R/bench_default.json $R/bench_driver_flags.json $R/bench_function_entry.json; cut -c1-70 $R/workloads.log
