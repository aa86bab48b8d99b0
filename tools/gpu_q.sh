#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -k "g9 or golden" 2>&1 | grep -E "Error|assert|error|passed|failed" | head -30
