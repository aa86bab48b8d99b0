#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -x -q -k "parked or parks or compiled or graph or plan or host_table" 2>&1 | tail -8
for e in ops function reference; do
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --entry $e 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(sys.argv[1], d['value'], d['ms_per_step'], d.get('rotated'), d['roofline']['step_frac'])" $e
done
