#!/bin/bash
# Round checkpoint: parity tests, smoke, bench matrix, rocprofv3 kernel stats, HBM traffic.
bash tools/gpu_check.sh > /dev/null 2>&1
tail -3 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/smoke.log | cut -c1-60
for cfg in "bf16_model" "bf16_model_serial --variant 4" "fp32_model --dtype fp32" "fp32_model_serial --dtype fp32 --variant 4" "bf16_test --inputs test"; do
  set -- $cfg; tag=$1; shift
  bash tools/gpu_prof.sh $tag "$@" > /dev/null 2>&1
  python tools/rocpd_stats.py gpurun_out/prof_$tag/trace_results.db | head -12 > gpurun_out/kernel_stats_$tag.txt
  rm -rf gpurun_out/prof_$tag
  bash tools/gpu_traffic.sh $tag "$@" > gpurun_out/traffic_$tag.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
done
head -6 gpurun_out/kernel_stats_bf16_model.txt
