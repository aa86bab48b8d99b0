"""Where the time of the rider workgroups goes (library built with -DBOXATTN_RIDE_TRACE=1: every rider writes
s_memrealtime stamps -- 100 MHz, common to all XCDs -- into the debug buffer): start, count done, ticket taken,
sub-range scan done, block scan done (training forward); start / end of the fill riders (backward).
    BOXATTN_HIP_LIB=boxer_amd/variants/libboxattn_trace.so python tools/gpu_ride_trace.py [boxattn_set_option(20) value]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boxer_amd import _lib, ops
lib = _lib.load()
if len(sys.argv) > 1:
    lib.boxattn_set_option(20, int(sys.argv[1]))
inp = bench.make_inputs("C2", torch.bfloat16, "cuda")
step = bench.make_step(inp)
for _ in range(20): step()
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
lib.boxattn_set_debug_buffer.argtypes = [__import__("ctypes").c_void_p]
lib.boxattn_set_debug_buffer(dbg.data_ptr())
torch.cuda.synchronize()
step()
torch.cuda.synchronize()
lib.boxattn_set_debug_buffer(0)
t = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
live = t[:, 0] > 0
t = t[live]
us = lambda x: x / 100.0
t0 = t[:, 0].min()
print("riders:", len(t))
print("count riders: start %.1f .. %.1f us after the first; count pass %.1f (median) %.1f (max) us; end of pass at most %.1f us" % (
    us(np.median(t[:, 0] - t0)), us((t[:, 0] - t0).max()), us(np.median(t[:, 1] - t[:, 0])), us((t[:, 1] - t[:, 0]).max()), us((t[:, 1] - t0).max())))
print("ticket taken: %.1f us (median) after the pass; last ticket at %.1f us" % (us(np.median(t[:, 2] - t[:, 1])), us((t[:, 2] - t0).max())))
last = t[:, 3] > 0
print("last arrivers: %d; sub-range scan %.1f us, block scan %.1f us; chain ends at %.1f us" % (
    last.sum(), us(np.median(t[last, 3] - t[last, 2])), us(np.median(t[last, 4] - t[last, 3])), us((t[last, 4] - t0).max())))
f = t[:, 5] > 0
f0 = t[f, 5].min()
print("fill riders: start .. %.1f us, duration %.1f (median) %.1f (max) us, all done at %.1f us" % (
    us((t[f, 5] - f0).max()), us(np.median(t[f, 6] - t[f, 5])), us((t[f, 6] - t[f, 5]).max()), us((t[f, 6] - f0).max())))
