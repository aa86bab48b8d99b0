"""A/B on one box: the one-pass fill (default) against the two-pass riders (boxattn_set_option(15, 4)), resident
(1 input set) and cache-cold (8 sets cycled), alternating, 3 rounds.   python tools/gpu_onepass_ab.py [dtype] [workload]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from boxer_amd import _lib, ops
dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
wl = sys.argv[2] if len(sys.argv) > 2 else "C2"
extra = [kv for kv in sys.argv[3:] if "=" in kv]
lib = _lib.load()
sets = [bench.make_inputs(wl, dtype, "cuda", seed=1000 + i) for i in range(8)]


def timed(mode, n_sets, steps=1500):
    lib.boxattn_set_option(15, mode)
    for kv in extra:
        k, v = kv.split("="); lib.boxattn_set_option(int(k), int(v))
    fns = [bench.make_step(x, "ops") for x in sets[:n_sets]]
    for k in range(40):
        fns[k % n_sets]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fns[k % n_sets]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for rnd in range(3):
    row = []
    for n_sets in (1, 8):
        for mode in (0, 4):
            row.append("%s %d set%s %.1f" % ("one-pass" if mode == 0 else "two-pass", n_sets, "s" if n_sets > 1 else "", timed(mode, n_sets)))
    print(wl, dtype, " | ".join(row), flush=True)
