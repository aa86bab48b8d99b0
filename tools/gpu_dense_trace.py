"""s_memtime phase times of the dense point-gradient kernel's waves at C2 bf16 (library built with
-DBOXATTN_DENSE_DEBUG=2; BOXATTN_HIP_LIB selects it).  Prints mean cycles per phase and query level."""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from boxer_amd import _lib

lib = _lib.load()
lib.boxattn_set_option(11, 2)
inp = bench.make_inputs(sys.argv[1] if len(sys.argv) > 1 else "C2", torch.bfloat16, "cuda",
                        family=sys.argv[2] if len(sys.argv) > 2 else "model")
step = bench.make_step(inp)
for _ in range(20):
    step()
n_waves = 200000
dbg = torch.zeros(n_waves * 20, device="cuda")
fn = lib.boxattn_set_debug_buffer
fn.argtypes = [ctypes.c_void_p]
fn.restype = None
fn(dbg.data_ptr())
step()
torch.cuda.synchronize()
fn(None)
d = dbg.view(-1, 20).cpu().numpy()
d = d[d[:, 19] > 0]
print("waves traced:", len(d))
names = ["decode", "issue rows", "issue own loads", "wait data", "lds writes+barrier", "levels", "epilogue+stores"]
for lq in range(4):
    w = d[d[:, 0] == lq]
    if not len(w):
        continue
    n = int(w[0, 19])
    ts = w[:, 2:1 + n]
    dt = np.diff(np.concatenate([np.zeros((len(w), 1)), ts], axis=1), axis=1)
    print("lq %d: %5d waves, life %7.0f cycles (min %d max %d)" % (lq, len(w), ts[:, -1].mean(), ts[:, -1].min(), ts[:, -1].max()))
    print("   " + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, dt.mean(0))))
start = d[:, 17].astype(np.int64)          # s_memrealtime, 10 ns ticks
end = d[:, 18].astype(np.int64)
t0 = start.min()
rel_s, rel_e = (start - t0) / 100.0, (end - t0) / 100.0          # us
print("wave starts (us): p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(rel_s, [10, 50, 90, 99, 100])))
print("wave ends   (us): p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(rel_e, [10, 50, 90, 99, 100])))
for lq in range(4):
    m = d[:, 0] == lq
    if m.any():
        print("  lq %d: starts p1 %.1f p50 %.1f p99 %.1f | ends p50 %.1f p99 %.1f max %.1f | life p50 %.1f us" % (
            (lq,) + tuple(np.percentile(rel_s[m], [1, 50, 99])) + tuple(np.percentile(rel_e[m], [50, 99, 100])) +
            (np.median(rel_e[m] - rel_s[m]),)))
edges = np.linspace(0, rel_e.max(), 31)
print("waves in flight every %.1f us:" % (edges[1] - edges[0]), [int(((rel_s <= t) & (rel_e > t)).sum()) for t in edges])
