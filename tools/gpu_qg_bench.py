"""Backward timing: query-grid backward against the binned one (GPU box).
usage: python tools/gpu_qg_bench.py [workload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from boxer_amd import _lib, ops
from gpu_tile_bench import timeit


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    fams = sys.argv[2].split(",") if len(sys.argv) > 2 else ["model"]
    for fam in fams:
        inp = bench.make_inputs(wl, torch.bfloat16, "cuda", family=fam)
        step = bench.make_step(inp)
        for name, variant, opts in (("binned", 8, {}), ("qgrid/512", 0, {}),
                                    ("qgrid/norounds", 0, {"qg_ablate": 1}), ("qgrid/nocand", 0, {"qg_ablate": 2}),
                                    ("qgrid/nocand+norounds", 0, {"qg_ablate": 3}),
                                    ("qgrid/nothing", 0, {"qg_ablate": 7})):
            _lib.set_variant(variant)
            _lib.set_option("qg_bwd", 1)
            old = {k: _lib.set_option(k, v) for k, v in opts.items()}
            t = timeit(step, 50)
            _lib.profile_begin()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            prof = _lib.profile_end()
            for k, v in old.items():
                _lib.set_option(k, v)
            print(wl, fam, name, "step %.1f us |" % t,
                  "  ".join("%s %.1f" % (k, v["ms"] * 1e3) for k, v in prof.items() if v["ms"]), flush=True)
        _lib.set_variant(0)


if __name__ == "__main__":
    main()
