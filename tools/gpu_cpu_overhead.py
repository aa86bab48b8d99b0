"""Host-side cost of one fwd_train + backward call pair (tiny problem: the GPU work is
negligible, so the loop time is the Python / ctypes / allocator overhead per step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boxer_amd import ops

bench.WORKLOADS["tiny"] = ([(8, 8), (4, 4)], 16, 4, "box")
for dt in (torch.bfloat16, torch.float32):
    inp = bench.make_inputs("tiny", dt, "cuda")
    step = bench.make_step(inp)
    for _ in range(50): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.1f us per step (enqueue only), %.1f us incl. drain" % (dt, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
import cProfile, pstats
inp = bench.make_inputs("tiny", torch.bfloat16, "cuda"); step = bench.make_step(inp)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
