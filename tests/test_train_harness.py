"""The synthetic BoxeR-2D training step of bench_train.py.

CPU / gloo, world size 2: layer wiring, DistributedDataParallel gradient all-reduce and the
sharding rule (different images per rank, identical weights after every step).  The operator
itself has no CPU implementation by design, so here -- and only here, in a test -- the modules'
autograd Function is replaced by the oracle's grid_sample formulation.
GPU: one step of a small configuration on the HIP kernels."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LEVELS = [(9, 7), (5, 4)]


class _OracleBox:
    @staticmethod
    def apply(value, shapes, lsi, loc, attn, im2col_step):
        from oracle import torch_fallback as tf
        return tf.box_attn(value, shapes, loc, attn)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(device, mask_decoder=False, model="2d"):
    import bench_train
    torch.manual_seed(0)
    net = bench_train.SyntheticBoxeR(LEVELS, model, d_model=32, n_head=8 if model == "3d" else 4,
                                     d_ffn=64, n_enc=2, n_dec=2, n_query=10, n_class=7,
                                     use_mask=mask_decoder).to(device)
    for m in net.attention_modules():
        with torch.no_grad():
            m.linear_box_weight.normal_(0, 0.05)
            m.linear_attn_weight.normal_(0, 0.05)
    return bench_train, net


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boxer_amd import modules
    modules.BoxAttnFunction = _OracleBox                     # test-only stand-in (CPU)
    bench_train, model = _build("cpu")
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(50 + rank)             # every rank its own images
    s = sum(h * w for h, w in LEVELS)
    batch = (torch.randn(2, s, 32, generator=g), 0.1 * torch.randn(1, s, 32, generator=g),
             torch.randn(2, 10, 7, generator=g), torch.rand(2, 10, 4, generator=g))
    losses = [float(bench_train.train_step(ddp, opt, batch)) for _ in range(3)]
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double()
    sums = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, torch.stack([flat.sum(), flat.abs().sum()]))
    grads_ok = all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    if rank == 0:
        torch.save(dict(losses=losses, sums=torch.stack(sums), grads_ok=grads_ok,
                        src0=float(batch[0][0, 0, 0])), out)
    else:
        torch.save(dict(src0=float(batch[0][0, 0, 0]), losses=losses), out + ".r1")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_training_step_world2(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out), torch.load(out + ".r1")
    assert r0["grads_ok"]
    assert r0["src0"] != r1["src0"]                          # different shards
    assert r0["losses"] != r1["losses"]
    assert r0["losses"][-1] < r0["losses"][0]                # it trains
    # the all-reduce keeps the replicas identical: same parameter checksums on both ranks
    assert torch.equal(r0["sums"][0], r0["sums"][1])


def test_encoder_reference_windows():
    from boxer_amd import layers
    ref = layers.encoder_ref_windows_2d([(2, 4)], 1)[0]
    assert ref.shape == (8, 4)
    assert torch.allclose(ref[0], torch.tensor([0.125, 0.25, 1.0, 2.0]), atol=1e-5)   # cx, cy, 4/W, 4/H
    assert torch.allclose(ref[-1, :2], torch.tensor([0.875, 0.75]), atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("native_bf16,fused,mask_decoder,kind",
                         [(False, False, False, "2d"), (True, True, False, "2d"),
                          (False, True, True, "2d"), (True, False, True, "2d"),
                          (False, False, False, "3d"), (True, True, False, "3d")])
def test_training_step_on_gpu(native_bf16, fused, mask_decoder, kind):
    bench_train, model = _build("cuda", mask_decoder, kind)
    for m in model.attention_modules():
        m.native_bf16, m.fused_grid = native_bf16, fused
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1)
    s = sum(h * w for h, w in LEVELS)
    batch = tuple(t.cuda() for t in (torch.randn(2, s, 32, generator=g),
                                     0.1 * torch.randn(1, s, 32, generator=g),
                                     torch.randn(2, 10, 7, generator=g),
                                     torch.rand(2, 10, 4, generator=g)))
    amp = torch.bfloat16 if native_bf16 else None
    losses = [float(bench_train.train_step(model, opt, batch, amp)) for _ in range(4)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]


@pytest.mark.gpu
def test_training_step_as_hip_graph():
    """bench_train.graphed_step: forward + backward + AdamW captured in one HIP graph (the
    operator's entry points neither allocate nor read on the host); replays train like eager
    steps from the same initial weights (same losses to the summation order of grad_value)."""
    import copy
    bench_train, model = _build("cuda")
    for m in model.attention_modules():
        m.native_bf16, m.fused_grid, m.fused_pointwise = True, 1, True
    eager = copy.deepcopy(model)
    g = torch.Generator().manual_seed(3)
    s = sum(h * w for h, w in LEVELS)
    batch = tuple(t.cuda() for t in (torch.randn(2, s, 32, generator=g),
                                     0.1 * torch.randn(1, s, 32, generator=g),
                                     torch.randn(2, 10, 7, generator=g),
                                     torch.rand(2, 10, 4, generator=g)))
    opt_e = torch.optim.AdamW(eager.parameters(), lr=1e-3, capturable=True)
    ref = [float(bench_train.train_step(eager, opt_e, batch, torch.bfloat16)) for _ in range(8)]
    opt_g = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    step = bench_train.graphed_step(model, opt_g, batch, torch.bfloat16)     # 3 eager warm-up steps, then the capture
    got = [float(step()) for _ in range(4)]
    torch.cuda.synchronize()
    assert all(l == l for l in got) and got[-1] < ref[0]
    # capturing records the step without running it: replay k is eager step 3 + k
    for a, b in zip(got, ref[3:]):
        assert abs(a - b) <= 1e-2 * max(1.0, abs(b)), (got, ref)
