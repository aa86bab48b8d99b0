import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import test_gpu_parity as T
from boxer_amd import ops
for cfg in ([(8, 8), (4, 4)], 1, 8, 32, 16, 4), T.SEEDED[0], T.FAST_CFGS[4]:
    g = T._seeded(*cfg, seed=5, lo=-0.5, hi=1.5)
    args = [T.dev(g["value"], torch.float32), T.dev(g["shapes"]), T.dev(g["lsi"]), T.dev(g["loc"], torch.float32), T.dev(g["attn"], torch.float32)]
    gout = T.dev(g["grad_out"], torch.float32)
    # poison a lot of allocator memory
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]
    del junk
    torch.cuda.synchronize()
    res = ops.box_attn_backward(*args, gout, 64)
    torch.cuda.synchronize()
    for name, t in zip(("gv", "gl", "ga"), res):
        bad = (~torch.isfinite(t)).nonzero()
        print(cfg[0], name, "nonfinite:", len(bad), bad[:5].tolist())
