"""Soak test of the scan riders: 200 back-to-back training steps per workload and storage
type with the block scans riding in the forward kernel's launch, every 20th step compared with a step that ran
the stand-alone scan kernels (a hand-off that goes stale shows up as a wrong bin offset sooner or later)."""
import sys, torch
sys.path.insert(0, ".")
import bench
from boxer_amd import _lib, ops
lib = _lib.load()
def flat(x):
    res = []
    for t in (x if isinstance(x, (list, tuple)) else [x]):
        if torch.is_tensor(t): res.append(t)
        elif isinstance(t, (list, tuple)): res.extend(flat(t))
    return res

for wl in ("C3", "C3p", "C3pp", "C2"):
    for dtype in (torch.bfloat16, torch.float32):
        inp = bench.make_inputs(wl, dtype, "cuda", family="model", seed=3)
        step = bench.make_step(inp)
        lib.boxattn_set_option(15, 1)
        ref = step(); torch.cuda.synchronize()
        ref = [t.clone() for t in flat(ref)]
        lib.boxattn_set_option(15, 0)
        worst = 0.0
        for it in range(200):
            out = step()
            out = flat(out)
            if it % 20 == 19:
                torch.cuda.synchronize()
                for a, b in zip(out, ref):
                    worst = max(worst, (a.float() - b.float()).abs().max().item() / max(1.0, b.float().abs().max().item()))
        print(wl, dtype, "tensors", len(ref), "worst scaled diff over 200 steps vs stand-alone scans: %.3e" % worst)
