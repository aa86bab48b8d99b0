#!/bin/bash
for lib in boxer_amd/variants/libboxattn_*.so; do
  echo "== $lib"
  BOXATTN_HIP_LIB=$PWD/$lib python -m pytest tests/test_gpu_parity.py -m gpu -q -k "graph_capture" 2>&1 | tail -3
done
