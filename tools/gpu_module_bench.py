"""Forward + backward of one BoxAttention module at BoxeR-R50 COCO shapes (encoder: one query per
pixel; decoder: 300 queries), fused_grid 0 / 1 x fused_pointwise, bf16 storage under autocast."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from boxer_amd import BoxAttention, layers
from gpu_tile_bench import timeit

ONLY = sys.argv[1:4]        # e.g. "encoder 2 1": one configuration (for rocprofv3)
levels = [(100, 167), (50, 84), (25, 42), (13, 21)]
shapes = torch.tensor(levels, device="cuda")
lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S, d, B = int(shapes.prod(1).sum()), 256, 2
torch.manual_seed(0)
m = BoxAttention(d, 4, 8).cuda()
m.native_bf16 = True
with torch.no_grad():
    m.linear_box_weight.normal_(0, 0.02)
value = torch.randn(B, S, d, device="cuda")
for name, Lq, ref in (("encoder", S, layers.encoder_ref_windows_2d(levels, B, device="cuda")),
                      ("decoder", 300, torch.rand(B, 300, 4, device="cuda") * 0.4 + 0.1)):
    query = torch.randn(B, Lq, d, device="cuda", requires_grad=True)
    if ONLY and ONLY[0] != name:
        continue
    for fg in (0, 1):
        for fp in (False, True):
            if ONLY and (fg, fp) != (int(ONLY[1]), ONLY[2] == "1"):
                continue
            m.fused_grid, m.fused_pointwise = fg, fp
            def step():
                query.grad = None
                m.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = m(query, value, shapes, None, lsi, None, ref)[0]
                out.float().sum().backward()
            print(name, "fused_grid", fg, "fused_pointwise", fp, "%.0f us" % timeit(step, 30), flush=True)
