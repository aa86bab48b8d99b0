"""Per-step time of the first steps after start-up (clock ramp / allocator warm-up)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
inp = bench.make_inputs("C2", torch.bfloat16, "cuda")
step = bench.make_step(inp)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(401)]
torch.cuda.synchronize()
ev[0].record()
for i in range(400):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(400)]
for a, b in ((0, 1), (1, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 200), (200, 400)):
    print("steps %3d-%3d: %.1f us/step" % (a, b, sum(t[a:b]) / (b - a)))
