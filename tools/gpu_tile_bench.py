"""Forward / backward timing of the query-grid kernels against the row-gather ones (GPU box).
usage: python tools/gpu_tile_bench.py [workload] [fwd|bwd|step]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from boxer_amd import _lib, ops


def timeit(fn, iters=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
    for dtype in (torch.bfloat16, torch.float32):
        for fam in ("model", "test"):
            inp = bench.make_inputs(wl, dtype, "cuda", family=fam)
            v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
            if what == "fwd":
                fn = lambda: ops.box_attn_forward(v, sh, ls, loc, attn, 64)
            elif what == "bwd":
                fn = lambda: ops.box_attn_backward(v, sh, ls, loc, attn, go, 64)
            else:
                fn = bench.make_step(inp)
            row = []
            _lib.set_option("tile_fwd", 1)
            _lib.set_variant(7)
            row.append(("gather", timeit(fn)))
            _lib.set_variant(0)
            cfgs = [("16x8", {"tile_shape": 1}), ("8x8r600", {"tile_shape": 2, "tile_rows": 600})]
            for base_name, base in list(cfgs):
                cfgs.append((base_name + "/static4", dict(base, tile_static_q16=64)))
                cfgs.append((base_name + "/static4/nocompute", dict(base, tile_static_q16=64, tile_ablate=1)))
                cfgs.append((base_name + "/static4/nostage", dict(base, tile_static_q16=64, tile_ablate=2)))
                cfgs.append((base_name + "/static4/neither", dict(base, tile_static_q16=64, tile_ablate=3)))
                cfgs.append((base_name + "/nocompute", dict(base, tile_ablate=1)))
            for name, opts in cfgs:
                old = {k: _lib.set_option(k, val) for k, val in opts.items()}
                try:
                    row.append((name, timeit(fn)))
                except Exception as e:
                    row.append((name, float("nan")))
                for k, val in old.items():
                    _lib.set_option(k, val)
            print(wl, what, str(dtype).split(".")[-1], fam, "  ".join("%s %.1f us" % r for r in row), flush=True)


if __name__ == "__main__":
    main()
