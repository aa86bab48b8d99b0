timeout 900 python -m pytest tests/test_train_harness.py -m gpu -x -q 2>&1 | tail -5
for a in "--dtype fp32" "--dtype fp32 --fused-grid" "--dtype bf16" "--dtype bf16 --fused-grid"; do
timeout 600 python bench_train.py --steps 10 --warmup 3 $a 2>&1 | tail -1
done
