#!/bin/bash
# Round-5 checkpoint on the GPU box: smoke, bench lines (default, driver flags, function entry), the dtype x input
# matrix, rocprofv3 kernel stats, HBM traffic (PMC, separate passes), SQ instruction mix, every workload, A/B of
# this round's switches.  -> gpurun_out/r05/
#   gpurun --timeout 2400 -- bash tools/gpu_round5_profiles.sh
R=gpurun_out/r05; mkdir -p $R; export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $R/gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > $R/smoke.log
timeout 600 python bench.py > $R/bench_default.json 2> $R/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > $R/bench_driver_flags.json 2>> $R/bench_default.err
timeout 300 python bench.py --entry function --no-cpu-baseline > $R/bench_function_entry.json 2>> $R/bench_default.err
timeout 300 python bench.py --entry reference --no-cpu-baseline > $R/bench_reference_entry.json 2>> $R/bench_default.err
for dt in bf16 fp32; do for inp in model test; do
  timeout 300 python bench.py --steps 200 --warmup 20 --dtype $dt --inputs $inp --no-cpu-baseline --rotate 0 2>/dev/null | tail -1 >> $R/bench_matrix.log
done; done
for cfg in "C2_bf16_model" "C2_fp32_model --dtype fp32" "C2_bf16_test --inputs test"; do
  set -- $cfg; tag=$1; shift
  bash tools/gpu_prof.sh $tag "$@" > /dev/null 2>&1
  python tools/rocpd_stats.py gpurun_out/prof_$tag/trace_results.db | head -14 > $R/kernel_stats_$tag.txt
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/rocprofv3_kernel_stats_$tag.csv
  rm -rf gpurun_out/prof_$tag
  bash tools/gpu_traffic.sh $tag "$@" > $R/pmc_traffic_$tag.txt 2>&1
  cp gpurun_out/traffic_$tag.json $R/ 2>/dev/null
  rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
done
bash tools/gpu_workloads.sh > $R/workloads.log 2>&1
# instruction mix and wait states of every kernel of the step (three passes)
bash tools/gpu_pmc_multi.sh r05sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU" > /dev/null 2>&1
python tools/sq_summary.py gpurun_out/pmc_r05sq_1 gpurun_out/pmc_r05sq_2 gpurun_out/pmc_r05sq_3 > $R/pmc_sq.txt 2>&1
rm -rf gpurun_out/pmc_r05sq_1 gpurun_out/pmc_r05sq_2 gpurun_out/pmc_r05sq_3
# ... and of the float32 step
bash tools/gpu_pmc_multi.sh r05sqf "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU" -- --dtype fp32 > /dev/null 2>&1
python tools/sq_summary.py gpurun_out/pmc_r05sqf_1 gpurun_out/pmc_r05sqf_2 gpurun_out/pmc_r05sqf_3 > $R/pmc_sq_fp32.txt 2>&1
rm -rf gpurun_out/pmc_r05sqf_1 gpurun_out/pmc_r05sqf_2 gpurun_out/pmc_r05sqf_3
# switches, one at a time against the default (boxattn_set_option key=value)
for o in "" "15=1" "15=2" "17=1" "11=1"; do
  for inp in model test; do
    echo -n "opt ${o:-default} inputs $inp : " >> $R/ab_switches.log
    timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --rotate 0 --inputs $inp ${o:+--opt $o} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('Gpts/s', d['value'], 'us/step', round(1000 * d['ms_per_step'], 1), {k: round(1000 * v['avg_ms'], 1) for k, v in r['kernels'].items()})" >> $R/ab_switches.log
  done
done
# float32 switches: accumulate flavour (19), window-staged kernels (21)
for o in "" "19=1" "19=2" "21=1" "15=1"; do
  echo -n "fp32 opt ${o:-default} : " >> $R/ab_switches.log
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --rotate 0 --dtype fp32 ${o:+--opt $o} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('Gpts/s', d['value'], 'us/step', round(1000 * d['ms_per_step'], 1), {k: round(1000 * v['avg_ms'], 1) for k, v in r['kernels'].items()})" >> $R/ab_switches.log
done
ls $R
