export TMPDIR=/tmp; mkdir -p gpurun_out
for o in 20=1024 20=1536 20=2048 20=4096; do echo "== $o"; python tools/gpu_onepass_ab.py bf16 C2 $o 2>&1 | grep "set" | tail -2; done | tee gpurun_out/r6_onepass_ab.log
for o in 20=2048 20=4096 20=6144; do echo "== C2p $o"; python tools/gpu_onepass_ab.py bf16 C2p $o 2>&1 | grep "set" | tail -1; done | tee -a gpurun_out/r6_onepass_ab.log
