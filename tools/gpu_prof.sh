#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench workload -> gpurun_out/prof_<tag>/
# usage: gpurun -- bash tools/gpu_prof.sh <tag> [bench args...]
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --preheat-s 0 "$@" > $out/bench.log 2>&1
tail -1 $out/bench.log
find $out -name "*kernel_stats*" | head -3
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && column -s, -t "$f" | cut -c1-200 | head -20
# keep only the stats (the raw trace is large)
find $out -name "*kernel_trace.csv" -size +2M -delete
