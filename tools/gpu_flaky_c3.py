"""C3 bf16 step time over repeated fresh allocations (hunting a sporadic 3.5x slow-down)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from boxer_amd import _lib

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
keep = []
for rep in range(10):
    if rep % 2:
        keep.append(torch.empty((rep * 7 + 1) * 1000003, dtype=torch.uint8, device="cuda"))  # shift the allocator
    inp = bench.make_inputs(wl, torch.bfloat16, torch.device("cuda"))
    step = bench.make_step(inp)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(100):
        step()
    t1.record()
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    print(rep, "%.1f us" % (t0.elapsed_time(t1) * 10),
          {k: round((v["ms"] or 0) * 1e3, 1) for k, v in prof.items()},
          [hex(inp[k].data_ptr() & 0xfffff) for k in ("value", "loc", "attn", "grad_out")], flush=True)
