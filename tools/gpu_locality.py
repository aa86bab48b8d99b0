"""Forward / point-gradient kernel time against the locality of the sampling locations
(same shapes as C2): tells apart the L1-hit rate bound from the instruction / TA bound."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boxer_amd import ops, _lib

def run(tag, inp):
    kind = inp["kind"]
    step = bench.make_step(inp)
    for _ in range(5): step()
    torch.cuda.synchronize()
    prof = bench.kernel_profile(step, 20, 4)
    print("%-28s" % tag, {k: round(1e3 * v["ms"], 1) for k, v in prof.items()}, flush=True)

for dt in (torch.bfloat16, torch.float32):
    inp = bench.make_inputs("C2", dt, "cuda", family="model")
    run("%s model" % dt, inp)
    loc = inp["loc"]
    # every point of a query at the query's own window centre (one 2x2 cell per level)
    ctr = loc.mean(dim=4, keepdim=True).expand_as(loc).contiguous()
    inp2 = dict(inp); inp2["loc"] = ctr
    run("%s centre-only" % dt, inp2)
    # all points of everything at one pixel
    inp3 = dict(inp); inp3["loc"] = torch.full_like(loc, 0.5)
    run("%s single pixel" % dt, inp3)
