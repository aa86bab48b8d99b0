#!/bin/bash
# Per-kernel rocprofv3 durations for every library under boxer_amd/variants/ (serial schedule).
#   gpurun --timeout 1200 -- bash tools/gpu_ab_stats.sh [bench args]
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in boxer_amd/variants/libboxattn_*.so; do
  name=$(basename $lib .so); name=${name#libboxattn_}
  export BOXATTN_HIP_LIB=$GRAFT_REPO_ROOT/$lib
  bash tools/gpu_prof.sh ab_$name --variant 4 "$@" > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "== $name"
  python tools/rocpd_stats.py gpurun_out/prof_ab_$name/trace_results.db | head -11 | cut -c1-150
  rm -rf gpurun_out/prof_ab_$name
done | tee gpurun_out/ab_stats.log
