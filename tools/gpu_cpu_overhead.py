"""Host-side cost of one training step (tiny problem: the GPU work is negligible, so the loop time is the Python /
ctypes / allocator / autograd overhead per step), for both entries of bench.py: the ops-level calls and the
autograd Functions."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

bench.WORKLOADS["tiny"] = ([(8, 8), (4, 4)], 16, 4, "box")
for entry in ("ops", "function"):
    for dt in (torch.bfloat16, torch.float32):
        inp = bench.make_inputs("tiny", dt, "cuda")
        step = bench.make_step(inp, entry)
        for _ in range(50): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2000
        for _ in range(n): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%-8s %s: host %.1f us per step (enqueue only), %.1f us incl. drain" % (entry, dt, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
import cProfile, pstats
for entry in ("ops", "function"):
    inp = bench.make_inputs("tiny", torch.bfloat16, "cuda"); step = bench.make_step(inp, entry)
    for _ in range(20): step()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): step()
    pr.disable(); torch.cuda.synchronize()
    print("==", entry)
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
