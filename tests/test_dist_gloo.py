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


_CHILD = """
import os, sys, torch, torch.distributed as dist
dist.init_process_group("gloo")          # RANK / WORLD_SIZE / MASTER_* from bench.spawn_ranks
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print("SUM=%d WORLD=%d LOCAL=%s" % (int(t.item()), dist.get_world_size(), os.environ["LOCAL_RANK"]))
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(t.item()) == 3 else 5)
"""


@pytest.mark.timeout(300)
def test_bench_self_launcher_starts_its_ranks(tmp_path, capfd):
    """``python bench.py --gpus N`` without a launcher: bench.spawn_ranks starts N fresh children
    with the rendezvous environment; here the children run a gloo all-reduce instead of the GPU
    bench (the reference's launcher: tools/run.py:59-75)."""
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    env_before = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "MASTER_PORT")}
    rc = bench.spawn_ranks(2, cmd=[sys.executable, str(script)])
    assert rc == 0
    assert "SUM=3 WORLD=2 LOCAL=0" in capfd.readouterr().out
    assert env_before == {k: os.environ.get(k) for k in env_before}      # parent env untouched


@pytest.mark.timeout(300)
def test_allreduce_bandwidth_tool_world2(capfd):
    """tools/rccl_allreduce_bench.py (RCCL all-reduce bus bandwidth at 25 / 160 MB) with its own
    launcher, gloo on the host at world size 2 and small messages: one JSON line per size, busbw = algbw * 2 (N-1)/N."""
    import json
    import subprocess
    tool = os.path.join(ROOT, "tools", "rccl_allreduce_bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, tool, "--gpus", "2", "--backend", "gloo", "--device", "cpu", "--iters", "3",
                          "--warmup", "1", "--sizes-mb", "0.5", "2"], capture_output=True, text=True, env=env,
                         timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert [l["bytes"] for l in lines] == [500000, 2000000]
    for l in lines:
        assert l["n_ranks"] == 2 and l["checked"] and l["ms"] > 0
        assert abs(l["busbw_GBs"] - l["algbw_GBs"] * 2 * (2 - 1) / 2) < 0.02
