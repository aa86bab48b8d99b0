"""Probe of the float32 split accumulate: one point per (query, head); compares grad_value with CPU emulations of the
split scheme truncated at different orders (which partial products does the kernel actually deliver?)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxer_amd import ops, _lib

def bf16_rn(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)

def split3(x):
    x1 = bf16_rn(x); r = (x - x1).astype(np.float32); x2 = bf16_rn(r); r2 = (r - x2).astype(np.float32)
    return x1, x2, bf16_rn(r2)

rng = np.random.default_rng(1)
B, H, C, Lq, P = 1, 8, 32, 1, 1
shapes = torch.tensor([(9, 11)], device="cuda"); lsi = torch.zeros(1, dtype=torch.long, device="cuda")
S = 99
value = torch.randn(B, S, H, C, device="cuda")
loc = torch.full((B, Lq, H, 1, P, 2), 0.5, device="cuda")      # pixel centre region
loc[..., 0] = (5 + 0.5) / 11; loc[..., 1] = (4 + 0.5) / 9     # exactly on pixel (4, 5): weights 1, 0, 0, 0
attn = torch.from_numpy(rng.uniform(0.2, 1.0, (B, Lq, H, 1, P)).astype(np.float32)).cuda()
gout = torch.from_numpy((rng.standard_normal((B, Lq, H * C))).astype(np.float32)).cuda()
for mode in (0, 1):
    _lib.load().boxattn_set_option(19, mode)
    gv = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64)[0]
    torch.cuda.synchronize()
    row = gv[0, 4 * 11 + 5].cpu().numpy().reshape(H, C)          # the pixel that gets weight 1 * a
    a = attn.cpu().numpy().reshape(H, 1); g = gout.cpu().numpy().reshape(H, C)
    want = (a.astype(np.float64) * g.astype(np.float64))
    w1, w2, w3 = split3(np.broadcast_to(a, g.shape).copy()); g1, g2, g3 = split3(g)
    def emu(terms):
        acc = np.zeros_like(g)
        for x, y in terms:
            acc = (acc.astype(np.float64) + x.astype(np.float64) * y.astype(np.float64)).astype(np.float32)
        return acc
    e6 = emu(((w1, g3), (w1, g2), (w1, g1), (w2, g2), (w2, g1), (w3, g1)))
    e3 = emu(((w1, g2), (w1, g1), (w2, g1)))
    e1 = emu(((w1, g1),))
    ulp = 2.0 ** (np.floor(np.log2(np.abs(want))) - 23)
    print("mode", mode, "err vs exact (ulp)", (np.abs(row - want) / ulp).max(),
          "| vs emu6", (np.abs(row - e6) / ulp).max(), "vs emu3", (np.abs(row - e3) / ulp).max(), "vs emu1", (np.abs(row - e1) / ulp).max())
    for name, terms in (("no w1g3", ((w1, g2), (w1, g1), (w2, g2), (w2, g1), (w3, g1))),
                        ("no w2g2", ((w1, g3), (w1, g2), (w1, g1), (w2, g1), (w3, g1))),
                        ("no w3g1", ((w1, g3), (w1, g2), (w1, g1), (w2, g2), (w2, g1)))):
        print("     vs emu", name, (np.abs(row - emu(terms)) / ulp).max())
