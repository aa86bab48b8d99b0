"""Where the time of the rider workgroups goes (library built with -DBOXATTN_RIDE_TRACE=1: every rider writes
s_memrealtime stamps -- 100 MHz, common to all XCDs -- into the debug buffer).
  two-pass riders (boxattn_set_option(15, 4)): start, count done, ticket taken, sub-range scan done, block scan done
      (training forward); start / end of the fill riders (backward);
  one-pass riders (default, boxattn_spec.h): slots 0-3 = time thread 0 spent in rank / claim / barrier wait / store,
      5 / 6 = start / end of the fill, 7 = end of the chain (the slice's last rider).
    BOXATTN_HIP_LIB=boxer_amd/variants/libboxattn_trace.so python tools/gpu_ride_trace.py [--opt KEY=VALUE ...]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boxer_amd import _lib
lib = _lib.load()
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.replace("--opt", "").strip().split("=")
        lib.boxattn_set_option(int(k), int(v))
inp = bench.make_inputs("C2", torch.bfloat16, "cuda")
step = bench.make_step(inp, "ops")
for _ in range(20): step()
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
lib.boxattn_set_debug_buffer.argtypes = [ctypes.c_void_p]
lib.boxattn_set_debug_buffer(dbg.data_ptr())
torch.cuda.synchronize()
step()
torch.cuda.synchronize()
lib.boxattn_set_debug_buffer(0)
t = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
us = lambda x: x / 100.0
f = t[:, 5] > 0
one_pass = f.any() and np.median(t[f, 0]) < 1e7
if not one_pass:
    live = t[:, 0] > 0
    c = t[live]
    if len(c):
        t0 = c[:, 0].min()
        print("count riders: %d; start .. %.1f us after the first; count pass %.1f (median) %.1f (max) us; last pass ends at %.1f us" % (
            len(c), us((c[:, 0] - t0).max()), us(np.median(c[:, 1] - c[:, 0])), us((c[:, 1] - c[:, 0]).max()), us((c[:, 1] - t0).max())))
        last = c[:, 3] > 0
        print("last arrivers: %d; chain ends at %.1f us" % (last.sum(), us((c[last, 4] - t0).max())))
if f.any():
    r = t[f]
    f0 = r[:, 5].min()
    print("fill riders: %d; start .. %.1f us, duration %.1f (median) %.1f (max) us, all done at %.1f us" % (
        len(r), us((r[:, 5] - f0).max()), us(np.median(r[:, 6] - r[:, 5])), us((r[:, 6] - r[:, 5]).max()), us((r[:, 6] - f0).max())))
    if one_pass:
        print("one-pass riders: rank %.1f, claim %.1f, wait %.1f, store %.1f us (median per rider; max %.1f / %.1f / %.1f / %.1f)" % (
            tuple(us(np.median(r[:, i])) for i in range(4)) + tuple(us(r[:, i].max()) for i in range(4))))
        c = r[:, 7] > 0
        if c.any():
            print("chains: %d, %.1f us (median) after their fill; last chain ends at %.1f us" % (
                c.sum(), us(np.median(r[c, 7] - r[c, 6])), us((r[c, 7] - f0).max())))
