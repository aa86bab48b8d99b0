"""One-pass fill under changing inputs: N input sets cycled; prints per-step time and the state's counters {chains run,
blocks redone} (boxattn_spec.h).   python tools/gpu_onepass_stats.py [sets] [steps] [dtype] [workload]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boxer_amd import ops
n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 800
dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
wl = sys.argv[4] if len(sys.argv) > 4 else "C2"
sets = [bench.make_inputs(wl, dtype, "cuda", seed=1000 + i) for i in range(n_sets)]
fns = [bench.make_step(x, "ops") for x in sets]


def counters():
    tot = [0, 0]
    for st in ops._STATE.values():
        c = st[1024:1040].view(torch.int64).cpu()
        tot[0] += int(c[0]); tot[1] += int(c[1])
    return tot


for k in range(3 * n_sets):
    fns[k % n_sets]()
torch.cuda.synchronize()
c0 = counters()
t0 = time.perf_counter()
for k in range(steps):
    fns[k % n_sets]()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
c1 = counters()
print(wl, "%d sets, %s: %.1f us/step; chains %d, blocks redone %d in %d steps (%.2f per step)" % (
    n_sets, dtype, dt * 1e6, c1[0] - c0[0], c1[1] - c0[1], steps, (c1[1] - c0[1]) / steps))
