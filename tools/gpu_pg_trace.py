"""Phase timestamps of the point-gradient kernel (library built with -DBOXATTN_TUNE_PG_TRACE=1:
wave 0 of every workgroup overwrites its first pair's grad_weight with s_memtime deltas):
start -> loop entry -> tile 1..4 -> end, in shader cycles; median and quartiles over the workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from boxer_amd import ops

inp = bench.make_inputs("C2", torch.bfloat16, torch.device("cuda"))
v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
for _ in range(3):
    gv, gl, ga = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64)
torch.cuda.synchronize()
B, Lq, H, L, P = attn.shape
ga = ga.reshape(-1, L * P).cpu().numpy()           # rows = (b, q, h) pairs
# a workgroup = 4 waves x 16 pairs; wave 0's first pair is row 64 * wg (plain mapping; the XCD
# remap permutes workgroups, not the row -> stamp relation)
rows = ga[::64]
d = rows[:, 1:8]
ok = (d[:, 6] > 0) & (d[:, 6] < 1e7)
d = d[ok]
print("workgroups:", len(d))
if os.environ.get("PG_TRACE") == "2":          # -DBOXATTN_TUNE_PG_TRACE=2: stamps of the prologue
    for k, n in ((1, "explicit kernel arguments arrived"), (2, "pair index computed (grid size read)"),
                 (5, "first loads issued"), (0, "loop entry (level table published)"), (6, "end")):
        print("%-40s at %7.0f cycles" % (n, np.median(d[:, k])))
    sys.exit(0)
names = ["loop entry", "tile 1", "tile 2", "tile 3", "tile 4", "(unused)", "end"]
print("first loads issued (before the level-table barrier) at %.0f cycles" % np.median(d[:, 5]))
prev = np.zeros(len(d))
for k, n in enumerate(names):
    if n == "(unused)":
        continue
    col = d[:, k]
    print("%-11s at %7.0f cycles (q25 %7.0f q75 %7.0f)   +%6.0f" % (
        n, np.median(col), np.percentile(col, 25), np.percentile(col, 75), np.median(col - prev)))
    prev = col
start = rows[ok][:, 0]
print("wave lifetime median %.0f cycles; kernel span of start stamps %.0f cycles (24-bit wrap ignored)" % (
    np.median(d[:, 6]), np.percentile(start, 99) - np.percentile(start, 1)))
