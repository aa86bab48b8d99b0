"""Debug aid: dense kernels on vs off on one ad-hoc encoder case; prints where the point gradients differ."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_gpu_dense as T
from boxer_amd import _lib

lv = sys.argv[1] if len(sys.argv) > 1 else "1lv"
fam = sys.argv[2] if len(sys.argv) > 2 else "test"
lib = _lib.load()
inp = T.make_case(T.LEVELS[lv], fam)
lib.boxattn_set_option(11, 2)
out_d, g_d = T.run(inp)
lib.boxattn_set_option(11, 1)
out_g, g_g = T.run(inp)
levels = T.LEVELS[lv]
for name, d, g in zip(("grad_value", "grad_loc", "grad_attn"), g_d, g_g):
    d, g = d.float().cpu().numpy(), g.float().cpu().numpy()
    err = np.abs(d - g)
    print(name, "max err", err.max(), "max ref", np.abs(g).max())
    if name == "grad_attn":
        bad = np.argwhere(err > 1e-4 * max(1, np.abs(g).max()))
        print("bad points:", len(bad), "of", err.size)
        loc = inp["loc"].cpu().numpy()
        S0 = np.cumsum([0] + [h * w for h, w in levels])
        for (b, q, h, l, p) in bad[:40]:
            lq = int(np.searchsorted(S0, q, side="right") - 1)
            qq = q - S0[lq]
            Hq, Wq = levels[lq]
            x, y = loc[b, q, h, l, p]
            Hl, Wl = levels[l]
            print("b%d q%d(lq%d y%d x%d) h%d l%d p%d  pix=(%.2f,%.2f) got %.5f want %.5f" % (
                b, q, lq, qq // Wq, qq % Wq, h, l, p, x * Wl - .5, y * Hl - .5, d[b, q, h, l, p], g[b, q, h, l, p]))
# corner sums S_k of every point, dumped by the kernel (boxattn_set_debug_buffer), against torch
import ctypes
lib.boxattn_set_option(11, 2)
d = inp["dims"]
B, Lq, H, L, P, C = d["B"], d["Lq"], d["H"], d["L"], d["P"], d["C"]
loc = inp["loc"]
val = inp["value"].float()              # (B,S,H,C)
go = inp["grad_out"].float().view(B, Lq, H, C)
S0 = np.cumsum([0] + [h * w for h, w in levels])
# expected corner sums
want = torch.zeros(B, Lq, H, L, P, 4, device="cuda")
for l, (Hl, Wl) in enumerate(levels):
    x = loc[:, :, :, l, :, 0] * Wl - 0.5
    y = loc[:, :, :, l, :, 1] * Hl - 0.5
    inside = (x > -1) & (y > -1) & (x < Wl) & (y < Hl)
    x0 = torch.floor(x).long(); y0 = torch.floor(y).long()
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = y0 + dy, x0 + dx
        ok = inside & (yy >= 0) & (yy < Hl) & (xx >= 0) & (xx < Wl)
        pix = int(S0[l]) + yy.clamp(0, Hl - 1) * Wl + xx.clamp(0, Wl - 1)       # (B,Lq,H,P)
        bi = torch.arange(B, device="cuda")[:, None, None, None].expand_as(pix)
        hi = torch.arange(H, device="cuda")[None, None, :, None].expand_as(pix)
        rows = val[bi, pix, hi]                                                 # (B,Lq,H,P,C)
        dot = (rows * go[:, :, :, None, :]).sum(-1)
        want[:, :, :, l, :, k] = torch.where(ok, dot, torch.zeros_like(dot))
dbg = torch.zeros(B * Lq * H * L * P * 8, device="cuda")
fn = lib.boxattn_set_debug_buffer
fn.argtypes = [ctypes.c_void_p]; fn.restype = None
fn(dbg.data_ptr())
junk = torch.empty(1 << 28, device="cuda", dtype=torch.uint8)
# expected x0 per point
x0_want = torch.zeros(B, Lq, H, L, P, device="cuda")
for l, (Hl, Wl) in enumerate(levels):
    x = loc[:, :, :, l, :, 0] * Wl - 0.5
    y = loc[:, :, :, l, :, 1] * Hl - 0.5
    inside = (x > -1) & (y > -1) & (x < Wl) & (y < Hl)
    x0_want[:, :, :, l, :] = torch.where(inside, torch.floor(x), torch.zeros_like(x))
for ablate in (0, 1):
    lib.boxattn_set_option(14, ablate)
    counts = []
    for trial in range(40):
        junk.fill_(trial)
        dbg.zero_()
        out, g = T.run(inp)
        got = dbg.view(B, Lq, H, L, P, 8)
        counts.append(int((got[..., 6] != x0_want).sum()))
    print("ablate", ablate, "points with a wrong x0 per run:", counts)
lib.boxattn_set_option(14, 0)
fn(None)
