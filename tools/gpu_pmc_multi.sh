#!/bin/bash
# Several rocprofv3 PMC passes (each its own run, kernel-trace only) over the default bench workload ->
# gpurun_out/pmc_<tag>_<n>/ and one summary on stdout.
# usage: gpurun -- bash tools/gpu_pmc_multi.sh <tag> "<counters pass 1>" "<counters pass 2>" ... [-- bench args]
tag=$1; shift
passes=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do passes+=("$1"); shift; done
[ "$1" == "--" ] && shift
n=0
for ctrs in "${passes[@]}"; do
    n=$((n + 1))
    bash $GRAFT_REPO_ROOT/tools/gpu_pmc.sh ${tag}_$n "$ctrs" "$@"
done
