#!/bin/bash
# rocprofv3 PMC pass (own run, kernel-trace only) -> gpurun_out/pmc_<tag>/
# usage: gpurun -- bash tools/gpu_pmc.sh <tag> "<counters>" [bench args]
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --preheat-s 0 "$@" > $out/bench.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
echo "file: $f"
python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"][:60]
        if "boxattn" not in k: continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s n=%3d  avg=%.4g" % (c, len(v), sum(v) / len(v)))
PY
