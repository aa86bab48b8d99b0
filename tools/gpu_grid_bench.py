"""Time the box -> grid tail of `_where_to_attend` (forward + backward) at C2 shapes:
torch elementwise ops (the reference's code path) against BoxGridFunction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxer_amd import BoxAttention, Box3dAttention, InstanceAttention

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

torch.manual_seed(0)
d, nl, nh = 256, 4, 8
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2      # B = 2 is launch / CPU bound in eager mode
for name, m, Lq, D in (("BoxAttention 2x2 encoder", BoxAttention(d, nl, nh, 2), 13294, 4),
                       ("InstanceAttention 14x14 decoder", InstanceAttention(d, nl, nh, 14), 300, 4),
                       ("Box3dAttention 2x2 rot", Box3dAttention(d, nl, nh, True, 2), 13294, 7)):
    m = m.cuda()
    query = torch.randn(B, Lq, d, device="cuda", requires_grad=True)
    ref = torch.rand(B, Lq, D, device="cuda")
    vr = 0.6 + 0.4 * torch.rand(B, 1, 1, nl, 1, 2, device="cuda")
    P = m.kernel_indices.size(0)
    g = torch.randn(B, Lq, nh, nl, P, 2, device="cuda")
    res = {}
    # the box-offset projection (a GEMM, identical in both paths) is kept out of the timing
    offs = m._box_offsets(query, ref, m.linear_box_bias.numel() // (nh * nl)).detach().requires_grad_()
    m._box_offsets = lambda q, r, n, o=offs: o
    for fused in (False, True):
        m.fused_grid = fused
        def step():
            offs.grad = None
            grid = m._where_to_attend(query, vr, ref)
            grid.backward(g)
        res[fused] = timeit(step)
    print("%-34s torch %.1f us   fused %.1f us   (grid %.1f MB)" % (
        name, res[False], res[True], B * Lq * nh * nl * P * 8 / 1e6), flush=True)
