#!/usr/bin/env python3
"""Synthetic BoxeR-2D training step around the operator (BASELINE.json
configs[3]): NOT the driver's headline bench (that is bench.py) -- a stand-in for "one
detection training step" that shows the operator's share of a step and the data-parallel path.

What it reproduces from the reference: the layer wiring and tensor shapes of the box
transformer (e2edet/module/box_transformer.py:316-465: post-norm encoder layer =
BoxAttention self-attention + FFN; decoder layer = nn.MultiheadAttention over the object
queries + BoxAttention cross-attention into the encoder memory + FFN), 6 + 6 layers, d_model
256, 8 heads, FFN 1024, 300 queries, 4 feature levels of a padded 1333x800 image, batch 2 per
GPU (base_boxer2d_detection.yaml:124-158).  What it replaces by synthetic stand-ins: backbone
features (N(0,1)), positional embeddings, matcher / criterion (a dummy regression loss on the
class and box heads).  One process per GPU; gradients are all-reduced by DistributedDataParallel
(RCCL with backend "nccl", gloo in the CPU test) -- the operator itself never communicates.

  python bench_train.py [--steps 30 --warmup 10 --dtype bf16|fp32 --fused-grid --fused-pointwise
                         --model 2d|3d --mask-decoder --split-k-wgrad --graph]
  python bench_train.py --gpus N          (starts its own N ranks; or under torch.distributed.run)
prints ONE JSON line on rank 0: ms per step (MAX over ranks), images/s, operator share.
"""
import argparse
import json
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

LEVELS_COCO = [(100, 167), (50, 84), (25, 42), (13, 21)]        # 1333x800 at strides 8..64
LEVELS_BEV = [(234, 234), (117, 117)]                           # Waymo 468 x 468 pillars, neck strides 2, 2


class SyntheticBoxeR(nn.Module):
    """Encoder + decoder stacks of boxer_amd.layers (the reference's layer classes, pinned by the
    G8 goldens) on synthetic features.  model "2d": BoxeR-2D detection (or, with ``use_mask``,
    instance segmentation: InstanceAttention 14 x 14 in the decoder); "3d": BoxeR-3D on a BEV
    map (Box3dAttention: 8 fixed per-head angles in the encoder, learned rotation in the
    decoder; box3d_transformer.py)."""

    def __init__(self, levels, model="2d", d_model=256, n_head=8, d_ffn=1024, n_enc=6, n_dec=6,
                 n_query=300, n_class=91, use_mask=False):
        super().__init__()
        from boxer_amd import layers
        self.levels, self.model, self.use_mask = levels, model, use_mask
        nl = len(levels)
        if model == "2d":
            self.encoder = nn.ModuleList(layers.BoxTransformerEncoderLayer(
                d_model, n_head, nl, d_ffn, 0.0, "relu") for _ in range(n_enc))
            self.decoder = nn.ModuleList(layers.BoxTransformerDecoderLayer(
                d_model, n_head, nl, d_ffn, 0.0, "relu", use_mask, "v1") for _ in range(n_dec))
            for layer in self.decoder:
                layer.inferencing = False
                layer.multihead_attn.inferencing = False
            ref_dim = 4
        else:
            self.encoder = nn.ModuleList(layers.Box3dTransformerEncoderLayer(
                d_model, n_head, nl, d_ffn, 0.0, "relu") for _ in range(n_enc))
            self.decoder = nn.ModuleList(layers.Box3dTransformerDecoderLayer(
                d_model, n_head, nl, d_ffn, 0.0, "relu") for _ in range(n_dec))
            ref_dim = 7
        self.roi_head = nn.Linear(d_model, 1) if use_mask else None
        self.query_embed = nn.Embedding(n_query, d_model)
        self.query_pos = nn.Embedding(n_query, d_model)
        self.query_ref = nn.Embedding(n_query, ref_dim)           # logits of (cx, cy, w, h[, angle, ..])
        self.class_head, self.box_head = nn.Linear(d_model, n_class), nn.Linear(d_model, 4)
        shapes = torch.tensor(levels, dtype=torch.long)
        self.register_buffer("shapes", shapes)
        self.register_buffer("lsi", torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1])))
        if model == "2d":
            self.register_buffer("enc_ref", layers.encoder_ref_windows_2d(levels, 1)[0])
        else:
            self.register_buffer("enc_ref", layers.encoder_ref_windows_3d(levels, 1)[0])

    def attention_modules(self):
        from boxer_amd import modules
        return [m for m in self.modules() if isinstance(m, modules._BoxAttentionBase)]

    def forward(self, src, pos):
        b = src.size(0)
        memory = src
        enc_ref = self.enc_ref[None].expand(b, *self.enc_ref.shape)
        tgt = self.query_embed.weight[None].expand(b, -1, -1)
        qpos = self.query_pos.weight[None].expand(b, -1, -1)
        ref = self.query_ref.weight.sigmoid()[None].expand(b, -1, -1)
        roi = None
        if self.model == "2d":
            for layer in self.encoder:
                memory = layer(memory, pos, self.shapes, None, self.lsi, None, enc_ref)
            for layer in self.decoder:
                tgt, roi = layer(tgt, qpos, memory, self.shapes, None, self.lsi, None, ref)
        else:
            for layer in self.encoder:
                memory = layer(memory, pos, self.shapes, self.lsi, enc_ref)
            for layer in self.decoder:
                tgt = layer(tgt, qpos, memory, self.shapes, self.lsi, ref)
        logits, boxes = self.class_head(tgt), self.box_head(tgt).sigmoid()
        if roi is not None:                       # (B, Lq, 14, 14, d) -> mask logits (B, Lq, 14, 14)
            return logits, boxes, self.roi_head(roi).squeeze(-1)
        return logits, boxes


def make_batch(levels, batch, d_model, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    s = sum(h * w for h, w in levels)
    src = torch.randn(batch, s, d_model, generator=g).to(device)
    pos = (0.1 * torch.randn(1, s, d_model, generator=g)).to(device)
    return src, pos, torch.randn(batch, 300, 91, generator=g).to(device), \
        torch.rand(batch, 300, 4, generator=g).to(device)


def train_step(model, opt, batch, autocast_dtype=None):
    src, pos, cls_t, box_t = batch
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
        out = model(src, pos)
    logits, boxes = out[0], out[1]
    loss = F.mse_loss(logits.float(), cls_t[:, :logits.size(1)]) + \
        F.l1_loss(boxes.float(), box_t[:, :boxes.size(1)])
    if len(out) == 3:                                 # dummy mask loss on the RoI branch
        loss = loss + out[2].float().square().mean()
    loss.backward()
    opt.step()
    return loss.detach()


def graphed_step(model, opt, batch, autocast_dtype):
    """One training step as ONE HIP graph launch: the eager step is ~2 500 kernel launches and,
    at BoxeR-2D shapes, as long on the host as on the GPU.  Warm-up on a side stream (allocator,
    host copies of the level tables, optimizer state), then capture forward + backward +
    optimizer step; the operator's entry points neither allocate nor read on the host, so they
    capture like any other kernel launch.  Static shapes and buffers (the synthetic batch)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            train_step(model, opt, batch, autocast_dtype)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = train_step(model, opt, batch, autocast_dtype)

    def replay():
        graph.replay()
        return loss
    replay.graph = graph
    return replay


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"],
                    help="bf16: autocast for the dense layers + the operator's native bf16 mode")
    ap.add_argument("--fused-grid", nargs="?", const=1, default=0, type=int,
                    help="1: grid construction in one kernel each way; 2: inside the sampling kernels")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (forward, backward, optimizer) in one HIP graph "
                         "(1 GPU; static synthetic batch)")
    ap.add_argument("--split-k-wgrad", action="store_true",
                    help="weight gradients of the S-long projections as batched split-K products "
                         "(boxer_amd.dense)")
    ap.add_argument("--fused-pointwise", action="store_true",
                    help="softmax and value mask + cast as single HIP passes (module.fused_pointwise)")
    ap.add_argument("--mask-decoder", action="store_true",
                    help="InstanceAttention (14x14) in the decoder: the instance-segmentation model")
    ap.add_argument("--layers", type=int, default=None, help="encoder = decoder layers (2d: 6, 3d: 2)")
    ap.add_argument("--model", default="2d", choices=["2d", "3d"],
                    help="3d: BoxeR-3D on a 234^2 + 117^2 BEV map (BASELINE configs[4])")
    args = ap.parse_args()
    if args.layers is None:
        args.layers = 6 if args.model == "2d" else 2

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # start our own ranks (bench.spawn_ranks)
        import sys
        import bench
        sys.exit(bench.spawn_ranks(args.gpus, cmd=[sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    from boxer_amd import _lib
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    torch.manual_seed(0)                                     # same initial weights on all ranks
    levels = LEVELS_COCO if args.model == "2d" else LEVELS_BEV
    model = SyntheticBoxeR(levels, args.model, n_enc=args.layers, n_dec=args.layers,
                           use_mask=args.mask_decoder and args.model == "2d").to(device)
    for m in model.attention_modules():
        m.native_bf16 = args.dtype == "bf16"
        m.fused_grid = args.fused_grid
        m.fused_pointwise = args.fused_pointwise
        with torch.no_grad():                                # trained-like box offsets
            m.linear_box_weight.normal_(0, 0.02)
    from boxer_amd import dense
    dense.set_split_k(args.split_k_wgrad)
    n_params = sum(p.numel() for p in model.parameters())
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])
    if args.graph and world > 1:
        raise SystemExit("--graph captures a single-GPU step (the DDP all-reduce is not captured)")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=args.graph)
    batch = make_batch(levels, args.batch, 256, device, seed=100 + rank)
    amp = torch.bfloat16 if args.dtype == "bf16" else None

    for _ in range(args.warmup):
        train_step(model, opt, batch, amp)
    torch.cuda.synchronize()
    # operator share: HIP events around every kernel launch of the library, two eager steps
    _lib.profile_begin()
    for _ in range(2):
        train_step(model, opt, batch, amp)
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    op_ms = sum((v["ms"] or 0.0) * v["launches"] for v in prof.values()) / 2

    step = (lambda: train_step(model, opt, batch, amp)) if not args.graph else \
        graphed_step(model, opt, batch, amp)
    if args.graph:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], device=device)
    if dist is not None:
        dist.barrier()
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    ms_step = float(elapsed) / args.steps * 1e3
    # The gradient all-reduce's share of the step (reference: DDP in
    # e2edet/trainer/base_trainer.py:119-129): the same steps once more without it (no_sync: gradients
    # accumulate locally) -- the difference is the part of the collective that backward does not hide.
    allreduce = None
    if dist is not None:
        dist.barrier()
        t0 = time.perf_counter()
        with model.no_sync():
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()
        local = torch.tensor([time.perf_counter() - t0], device=device)
        dist.barrier()
        dist.all_reduce(local, op=dist.ReduceOp.MAX)
        ms_local = float(local) / args.steps * 1e3
        grad_bytes = 4 * n_params
        allreduce = {"exposed_ms_per_step": round(max(0.0, ms_step - ms_local), 3),
                     "share_of_step": round(max(0.0, ms_step - ms_local) / ms_step, 3),
                     "gradient_MB": round(grad_bytes / 1e6, 1), "ms_per_step_without": round(ms_local, 3),
                     "note": "step time with DDP's bucketed all-reduce (RCCL) minus the same steps under no_sync(); "
                             "bus bandwidth of the collective itself: tools/rccl_allreduce_bench.py"}

    if rank == 0:
        print(json.dumps({
            "metric": "synthetic BoxeR-%s training step (%d+%d layers, levels %s)" % (
                args.model.upper(), args.layers, args.layers, "/".join("%dx%d" % hw for hw in levels)),
            "ms_per_step": round(ms_step, 3), "images_per_s": round(world * args.batch / ms_step * 1e3, 2),
            "n_gpus": world, "batch_per_gpu": args.batch, "dtype": args.dtype,
            "fused_grid": args.fused_grid, "fused_pointwise": args.fused_pointwise, "split_k_wgrad": args.split_k_wgrad, "graph": args.graph, "mask_decoder": args.mask_decoder, "params_M": round(n_params / 1e6, 2),
            "operator_kernels_ms_per_step": round(op_ms, 3),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
            "operator_share": round(op_ms / ms_step, 3), "loss": round(float(loss), 4),
            "allreduce": allreduce, "data": "synthetic", "scaling": "weak"}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
