#!/bin/bash
# HBM traffic of the op's kernels from the L2 memory-side counters, two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Writes gpurun_out/traffic_<tag>.json
# usage: gpurun -- bash tools/gpu_traffic.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$c
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --preheat-s 0 "$@" > $out/bench.log 2>&1
done
python3 - $GRAFT_REPO_ROOT/gpurun_out $tag <<'PY'
import csv, sys, collections, json, glob, os
root, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(root, "pmc_%s_%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "boxattn" in row["Kernel_Name"] and row["Counter_Name"] == c:
            agg[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        res[k][c + "_KB_avg"] = sum(v) / len(v)
json.dump(res, open(os.path.join(root, "traffic_%s.json" % tag), "w"), indent=1)
for k, v in res.items():
    print("%-70s %s" % (k[:70], {a: round(b, 1) for a, b in v.items()}))
PY
