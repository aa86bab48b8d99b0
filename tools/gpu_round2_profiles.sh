#!/bin/bash
# Round-2 checkpoint on the GPU box: smoke, bench lines, rocprofv3 kernel stats, HBM traffic (PMC,
# separate passes), workload matrix, synthetic training steps.  Everything under gpurun_out/r02/.
#   gpurun --timeout 2400 -- bash tools/gpu_round2_profiles.sh
R=gpurun_out/r02; mkdir -p $R; export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $R/gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > $R/smoke.log
timeout 600 python bench.py > $R/bench_default.json 2> $R/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > $R/bench_driver_flags.json 2>> $R/bench_default.err
for dt in bf16 fp32; do for inp in model test; do
  timeout 300 python bench.py --steps 200 --warmup 20 --dtype $dt --inputs $inp --no-cpu-baseline 2>/dev/null | tail -1 >> $R/bench_matrix.log
done; done
for cfg in "C2_bf16_model" "C2_fp32_model --dtype fp32" "C2_bf16_test --inputs test"; do
  set -- $cfg; tag=$1; shift
  bash tools/gpu_prof.sh $tag "$@" > /dev/null 2>&1
  python tools/rocpd_stats.py gpurun_out/prof_$tag/trace_results.db | head -14 > $R/kernel_stats_$tag.txt
  rm -rf gpurun_out/prof_$tag
  bash tools/gpu_traffic.sh $tag "$@" > $R/pmc_traffic_$tag.txt 2>&1
  cp gpurun_out/traffic_$tag.json $R/ 2>/dev/null
  rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
done
bash tools/gpu_workloads.sh > $R/workloads.log 2>&1
rm -f $R/train_step.log
for args in "" "--fused-grid 1" "--fused-grid 1 --fused-pointwise" "--fused-grid 2 --fused-pointwise" \
            "--fused-grid 1 --fused-pointwise --split-k-wgrad" "--graph" \
            "--fused-grid 1 --fused-pointwise --split-k-wgrad --graph" "--dtype fp32" \
            "--dtype fp32 --fused-grid 1 --fused-pointwise --split-k-wgrad --graph" "--mask-decoder" \
            "--mask-decoder --fused-grid 1 --fused-pointwise --split-k-wgrad --graph" "--model 3d" \
            "--model 3d --fused-grid 1 --fused-pointwise --split-k-wgrad" \
            "--model 3d --fused-grid 1 --fused-pointwise --split-k-wgrad --graph"; do
  echo "bench_train.py $args" >> $R/train_step.log
  timeout 600 python bench_train.py $args 2>/dev/null | tail -1 >> $R/train_step.log
done
python tools/gpu_module_bench.py 2>/dev/null | grep -v amdgpu > $R/module_bench.log
python tools/gpu_chunk_sweep.py C3 C3pp C5 2>/dev/null | grep -v amdgpu > $R/chunk_sweep.log
bash tools/gpu_pmc.sh r02sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" > $R/pmc_sq.txt 2>&1
# the opt-in query-grid experiments next to the defaults
python tools/gpu_tile_bench.py C2 fwd 2>/dev/null | grep -v amdgpu > $R/experiment_tile_forward.log
python tools/gpu_qg_bench.py C2 model 2>/dev/null | grep -v amdgpu > $R/experiment_qgrid_backward.log
ls $R
