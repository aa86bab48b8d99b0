import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import test_gpu_parity as T
from oracle import boxattn_oracle as oc
cfg = T.FAST_CFGS[4]
g = T._seeded(*cfg, seed=22, lo=-0.2, hi=1.2)
want = oc.instance_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["spatial_w"], g["level_w"], g["grad_out"], g["grad_mask"])
for variant in ("atomic", "binned", "generic"):
    for rep in range(2):
        out, mask, gv, gl, gs, glw = T.run_inst(g, torch.float32, variant)
        for name, got, w in (("gv", gv, want[0]), ("gl", gl, want[1]), ("gs", gs, want[2]), ("glw", glw, want[3])):
            d = np.abs(got.double().cpu().numpy() - w)
            bad = np.argwhere(d > 1e-3 * max(1, np.abs(w).max()))
            print(variant, rep, name, "maxerr %.3e scale %.3e nbad %d" % (d.max(), np.abs(w).max(), len(bad)), bad[:4].tolist())
