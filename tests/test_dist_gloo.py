"""CPU, world_size 2, gloo: the multi-process timing / aggregation protocol of bench.py
(barrier + sync on both sides of the timed region, MAX over ranks, whole-job throughput) and
the data-parallel sharding rule (each rank owns its own images; no data-path collective)."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    calls = []

    def step():                      # rank 1 is the slow one
        calls.append(1)
        time.sleep(0.01 * (rank + 1))

    elapsed = bench.run_timed(step, steps=5, warmup=2, sync=lambda: None, dist=dist,
                              device=torch.device("cpu"))
    value, ms = bench.throughput(elapsed, 1000, world, 5)
    # sharding: ranks draw different images (seed = rank) of identical shape
    a = bench.make_inputs("C2", torch.float32, "cpu", family="test", batch=1, seed=rank)
    sig = torch.tensor([float(a["value"][0, 0, 0, 0]), float(a["loc"].numel())],
                       dtype=torch.float64)
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(sigs, sig)
    if rank == 0:
        torch.save(dict(elapsed=elapsed, value=value, ms=ms, calls=len(calls),
                        sigs=torch.stack(sigs)), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bench_protocol_world2(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["calls"] == 7                                   # 2 warm-up + 5 timed
    assert 0.095 <= r["elapsed"] < 0.5                       # MAX over ranks: 5 x 20 ms
    assert abs(r["value"] - 1000 * 2 * 5 / r["elapsed"] / 1e9) < 1e-12
    assert abs(r["ms"] - r["elapsed"] / 5 * 1e3) < 1e-9
    sigs = r["sigs"]
    assert sigs[0, 1] == sigs[1, 1] and sigs[0, 0] != sigs[1, 0]   # same shape, own data
