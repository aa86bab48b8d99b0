"""Step time of the small workloads against the binned backward's records-per-item setting
(boxattn_set_option 10; 0 = the shape-derived default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from boxer_amd import _lib

names = sys.argv[1:] or ["C3", "C3pp", "C5", "C5pp", "C3p", "C2"]
for name in names:
    for dtype in ("bf16", "fp32"):
        row = []
        for chunk in (0, 64, 128, 256, 512, 1024):
            _lib.set_option("bin_chunk", chunk)
            inp = bench.make_inputs(name, torch.bfloat16 if dtype == "bf16" else torch.float32,
                                    torch.device("cuda"))
            step = bench.make_step(inp)
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(200):
                step()
            t1.record()
            torch.cuda.synchronize()
            row.append("%4d: %6.1f us" % (chunk, t0.elapsed_time(t1) * 1e3 / 200))
        _lib.set_option("bin_chunk", 0)
        print(name, dtype, " | ".join(row), flush=True)
