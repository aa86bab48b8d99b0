"""Soak of the default training step (riders -- also on the BEV encoder's 2 220-block maps --, in-launch combine,
direct-to-LDS staging, the float32 staged kernels and split accumulate): N back-to-back steps per
workload and storage type; every step's out / grad_loc / grad_weights must be bit-identical to the first step's
(they do not depend on any summation order) and grad_value within the storage type's tolerance of it.
    gpurun -- python tools/gpu_soak_step.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from boxer_amd import ops  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ops._Locality.enabled = False          # same kernels in every step
bad = 0
for wl in ("C2", "C2p", "C3", "C5p", "C5pp"):
    for dtype in (torch.bfloat16, torch.float32):
        inp = bench.make_inputs(wl, dtype, "cuda", family="model", batch=2, seed=3)
        step = bench.make_step(inp, "ops")
        ref = None
        n_diff = torch.zeros((), device="cuda", dtype=torch.int64)
        worst = torch.zeros((), device="cuda")
        for i in range(steps):
            out, grads = step()
            cur = (out if not isinstance(out, (tuple, list)) else out[0],) + tuple(grads)
            if ref is None:
                ref = [t.clone() for t in cur]
                continue
            for k, (a, b) in enumerate(zip(cur, ref)):
                if k == 1:                                   # grad_value: a sum in unspecified order
                    worst = torch.maximum(worst, (a.float() - b.float()).abs().max())
                else:
                    n_diff += (a != b).any().to(torch.int64)
        torch.cuda.synchronize()
        scale = max(1.0, ref[1].float().abs().max().item())
        tol = (1e-2 if dtype == torch.bfloat16 else 1e-4) * scale
        ok = n_diff.item() == 0 and worst.item() <= tol
        bad += not ok
        print("%-5s %-5s %d steps: %d tensors differed from step 0, grad_value max |diff| %.3g (tol %.3g) %s"
              % (wl, str(dtype).split(".")[-1], steps, n_diff.item(), worst.item(), tol, "ok" if ok else "FAILED"))
sys.exit(1 if bad else 0)
