"""Debugging aid for the window-staged forward: structured inputs on a small encoder shape."""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bench
from boxer_amd import _lib, ops
import test_gpu_dense as tgd

lib = _lib.load()
levels = [(16, 16)] if len(sys.argv) < 2 else eval(sys.argv[1])
inp = tgd.make_case(levels, "model", H=1, B=1, seed=1)
v, sh, ls, loc, attn = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn"))
S = v.shape[1]
def run(vv):
    r = {}
    for mode in (0, 2):
        lib.boxattn_set_option(17, mode)
        r[mode] = ops.box_attn_forward(vv, sh, ls, loc, attn, 64).float()
        torch.cuda.synchronize()
    lib.boxattn_set_option(17, 0)
    return r
for name, vv in (("ones", torch.ones_like(v)),
                 ("channel index", (torch.arange(32, device="cuda").float() / 8).expand_as(v).contiguous().bfloat16()),
                 ("pixel index", (torch.arange(S, device="cuda").float() / 16).view(1, S, 1, 1).expand_as(v).contiguous().bfloat16()),
                 ("random", v)):
    r = run(vv)
    d = (r[2] - r[0]).abs()
    print("%-14s max diff %.4g  ref[0,0,:6] %s  got[0,0,:6] %s" % (name, d.max().item(), [round(x, 3) for x in r[0][0, 0, :6].tolist()], [round(x, 3) for x in r[2][0, 0, :6].tolist()]))
    if name == "ones":
        print("   ref[0,:4,0]", r[0][0, :4, 0].tolist(), " got", r[2][0, :4, 0].tolist())
        bad = (d.amax(-1) > 0.02).view(-1)
        print("   queries off:", int(bad.sum()), "of", bad.numel(), " first bad:", bad.nonzero().view(-1)[:12].tolist())
if len(sys.argv) > 2:       # library built with -DBOXATTN_DENSE_DEBUG=3: per-lane dump of the first point step
    import ctypes
    dbg = torch.zeros(64 * 16, device="cuda")
    fn = lib.boxattn_set_debug_buffer
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = None
    fn(dbg.data_ptr())
    lib.boxattn_set_option(17, 2)
    ops.box_attn_forward(torch.ones_like(v), sh, ls, loc, attn, 64)
    torch.cuda.synchronize()
    fn(None)
    lib.boxattn_set_option(17, 0)
    names = "pack pk addr a0lo a0hi a1lo a1hi b0 b1 b2 b3 w0 w1 y0 x0 d".split()
    print(" lane " + " ".join("%8s" % n for n in names))
    for ln, row in enumerate(dbg.view(64, 16).cpu().tolist()[:20]):
        print("%5d " % ln + " ".join("%8.4g" % x for x in row))
