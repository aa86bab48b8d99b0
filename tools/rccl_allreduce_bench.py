#!/usr/bin/env python3
"""All-reduce bus bandwidth of the node's collective library (RCCL over xGMI on a GPU node), at the two
message sizes of the reference's trainer: one 25 MB DistributedDataParallel bucket and the ~160 MB of all
gradients of the BoxeR-2D R50 model (~40 M float32 parameters; e2edet/trainer/base_trainer.py:119-129 wraps
the model in DDP, e2edet/utils/distributed.py:319 is its explicit all_reduce).  SURVEY.md 8(e) asks for this
number next to the DDP training step (bench_train.py --gpus N prints the step's all-reduce share).

    python tools/rccl_allreduce_bench.py --gpus 8            # starts its own ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        tools/rccl_allreduce_bench.py --gpus 8               # or under a launcher

Per size: W warm-up + K timed all-reduces bracketed by barrier + device sync, MAX over ranks.
  algbw = bytes / time;  busbw = algbw * 2 (N - 1) / N  (what every link of a ring carries; the xGMI bound per
  direction is one ~153 GB/s link for a single ring, more if the library stripes rings over several links).
Prints one JSON line per size on rank 0.  --backend gloo --device cpu runs the same protocol on the host
(tests/test_dist_gloo.py)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZES_MB = (25, 160)


def bus_bandwidth(nbytes, seconds, world):
    """-> (algbw, busbw) in GB/s for one all-reduce of nbytes per rank."""
    alg = nbytes / seconds / 1e9
    return alg, alg * 2.0 * (world - 1) / world


def time_allreduce(dist, tensor, iters, warmup, sync):
    for _ in range(warmup):
        dist.all_reduce(tensor)
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_reduce(tensor)
    sync()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=tensor.device)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    return float(elapsed.item()) / iters


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sizes-mb", type=float, nargs="+", default=list(SIZES_MB))
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])        # "nccl" IS RCCL on ROCm
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench                                    # the repository's self-launcher (fresh children, before
        sys.exit(bench.spawn_ranks(args.gpus, cmd=[sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if args.device == "cuda":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=device)
        sync = torch.cuda.synchronize
    else:
        device = torch.device("cpu")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        sync = lambda: None
    for mb in args.sizes_mb:
        n = max(1, int(mb * 1e6) // 4)
        t = torch.ones(n, dtype=torch.float32, device=device)
        sec = time_allreduce(dist, t, args.iters, args.warmup, sync)
        # the sum of ones over (1 + warm-up + iters) rounds stays exact in float32 for these counts: a wrong
        # collective does not go unnoticed
        want = float(world) ** (args.iters + args.warmup)
        ok = bool(torch.isinf(t[0]) or abs(float(t[0]) / want - 1.0) < 1e-3)
        alg, bus = bus_bandwidth(n * 4, sec, world)
        if rank == 0:
            print(json.dumps({"metric": "all-reduce bus bandwidth", "bytes": n * 4, "n_ranks": world,
                              "backend": args.backend, "ms": round(sec * 1e3, 4), "algbw_GBs": round(alg, 2),
                              "busbw_GBs": round(bus, 2), "checked": ok, "iters": args.iters}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
