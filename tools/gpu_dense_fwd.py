"""Window-staged matrix-core forward (boxattn_set_option(17, 2)) next to fwd2_kernel at a bench workload:
largest difference of `out`, share of differing elements, kernel times (HIP events of the library)."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from boxer_amd import _lib, ops

lib = _lib.load()
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
fam = sys.argv[2] if len(sys.argv) > 2 else "model"
inp = bench.make_inputs(wl, torch.bfloat16, "cuda", family=fam)
v, sh, ls, loc, attn = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn"))
res = {}
for mode in (0, 2):
    lib.boxattn_set_option(17, mode)
    out = ops.box_attn_forward(v, sh, ls, loc, attn, 64)
    torch.cuda.synchronize()
    res[mode] = out.float()
    for _ in range(20):
        ops.box_attn_forward(v, sh, ls, loc, attn, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.box_attn_forward(v, sh, ls, loc, attn, 64)
    e1.record()
    torch.cuda.synchronize()
    print("option 17 = %d: %.1f us per forward call" % (mode, e0.elapsed_time(e1) * 5.0))
lib.boxattn_set_option(17, 0)
d = (res[2] - res[0]).abs()
scale = res[0].abs().max().item()
print("max |diff| %.4g (scale %.3g), elements off by more than 1e-2 scale: %.4f %%, more than 2^-7 relative: %.4f %%" % (
    d.max().item(), scale, 100.0 * (d > 1e-2 * scale).float().mean().item(),
    100.0 * (d > 2.0 ** -7 * res[0].abs().clamp_min(1e-3 * scale)).float().mean().item()))
