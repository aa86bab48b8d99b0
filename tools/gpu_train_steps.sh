#!/bin/bash
# Check of the synthetic training steps around the operator (same configurations as
# profiles/r02_train_step.log) -> gpurun_out/r04t/train_step.log (copied to profiles/r0N_train_step.log)
R=gpurun_out/r04t; mkdir -p $R; rm -f $R/train_step.log
for args in "" "--fused-grid 1 --fused-pointwise --split-k-wgrad" "--fused-grid 1 --fused-pointwise --split-k-wgrad --graph" \
            "--model 3d" "--model 3d --fused-grid 1 --fused-pointwise --split-k-wgrad --graph"; do
  echo "bench_train.py $args" >> $R/train_step.log
  timeout 600 python bench_train.py $args 2>/dev/null | tail -1 >> $R/train_step.log
done
cat $R/train_step.log
