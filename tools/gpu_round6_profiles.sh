#!/bin/bash
# Round-6 checkpoint on the GPU box: smoke, bench lines (default = reference entry + cache-cold cycle; driver flags; ops /
# function entries), the dtype x input matrix, rocprofv3 kernel stats, fabric traffic (PMC, separate passes), SQ instruction
# mix, every workload, the one-pass / two-pass A/B and the rider trace.  -> gpurun_out/r06/
#   gpurun --timeout 2400 -- bash tools/gpu_round6_profiles.sh
R=gpurun_out/r06; mkdir -p $R; export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $R/gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 > $R/smoke.log
timeout 900 python bench.py > $R/bench_default.json 2> $R/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 > $R/bench_driver_flags.json 2>> $R/bench_default.err
timeout 300 python bench.py --entry ops --no-cpu-baseline > $R/bench_ops_entry.json 2>> $R/bench_default.err
timeout 300 python bench.py --entry function --no-cpu-baseline > $R/bench_function_entry.json 2>> $R/bench_default.err
for dt in bf16 fp32; do for inp in model test; do
  timeout 300 python bench.py --steps 300 --warmup 20 --dtype $dt --inputs $inp --no-cpu-baseline 2>/dev/null | tail -1 >> $R/bench_matrix.log
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
bash tools/gpu_pmc_multi.sh r06sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU" > /dev/null 2>&1
python tools/sq_summary.py gpurun_out/pmc_r06sq_1 gpurun_out/pmc_r06sq_2 gpurun_out/pmc_r06sq_3 > $R/pmc_sq.txt 2>&1
rm -rf gpurun_out/pmc_r06sq_1 gpurun_out/pmc_r06sq_2 gpurun_out/pmc_r06sq_3
# one-pass fill against the two-pass riders on this box, resident and cache-cold
( python tools/gpu_onepass_ab.py bf16 C2; python tools/gpu_onepass_ab.py fp32 C2; python tools/gpu_onepass_ab.py bf16 C2p ) 2>&1 | grep " set" > $R/onepass_ab.log
python tools/gpu_onepass_stats.py 8 800 bf16 C2 2>&1 | grep sets >> $R/onepass_ab.log
# switches, one at a time against the default
for o in "" "15=1" "15=2" "15=4" "11=1"; do
  for inp in model test; do
    echo -n "opt ${o:-default} inputs $inp : " >> $R/ab_switches.log
    timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --inputs $inp ${o:+--opt $o} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('Gpts/s', d['value'], 'us/step', round(1000 * d['ms_per_step'], 1), 'resident', round(1000 * d['resident']['ms_per_step'], 1), {k: round(1000 * v['avg_ms'], 1) for k, v in r['kernels'].items()})" >> $R/ab_switches.log
  done
done
for o in "" "19=1" "19=2" "11=1" "15=4"; do
  echo -n "fp32 opt ${o:-default} : " >> $R/ab_switches.log
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --dtype fp32 ${o:+--opt $o} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('Gpts/s', d['value'], 'us/step', round(1000 * d['ms_per_step'], 1), 'resident', round(1000 * d['resident']['ms_per_step'], 1), {k: round(1000 * v['avg_ms'], 1) for k, v in r['kernels'].items()})" >> $R/ab_switches.log
done
ls $R
