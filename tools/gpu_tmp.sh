./tools/microbench_gather.bin 2>&1 | tee gpurun_out/microbench_gather.log
