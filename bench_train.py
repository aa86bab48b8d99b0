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

  python bench_train.py [--steps 10 --warmup 3 --dtype bf16|fp32 --fused-grid]
  python -m torch.distributed.run --nproc-per-node N bench_train.py --gpus N
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


class EncoderLayer(nn.Module):
    """Post-norm: x = LN(x + BoxAttention(x + pos, x)); x = LN(x + FFN(x))."""

    def __init__(self, attn_cls, d_model, n_head, n_level, d_ffn):
        super().__init__()
        self.self_attn = attn_cls(d_model, n_level, n_head)
        self.linear1, self.linear2 = nn.Linear(d_model, d_ffn), nn.Linear(d_ffn, d_model)
        self.norm1, self.norm2 = nn.LayerNorm(d_model), nn.LayerNorm(d_model)

    def forward(self, src, pos, shapes, mask, lsi, ratios, ref_windows):
        src = self.norm1(src + self.self_attn(src + pos, src, shapes, mask, lsi, ratios,
                                              ref_windows)[0])
        return self.norm2(src + self.linear2(F.relu(self.linear1(src))))


class DecoderLayer(nn.Module):
    """Self-attention over the object queries, box cross-attention into the memory, FFN.
    With `mask_cls` (InstanceAttention, kernel 14: the instance-segmentation decoder,
    box_transformer.py:381-384) the cross-attention also returns the per-point RoI features,
    which are handed back so that their branch takes part in the backward."""

    def __init__(self, attn_cls, d_model, n_head, n_level, d_ffn, mask_cls=None):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, n_head)
        self.use_mask = mask_cls is not None
        if self.use_mask:
            self.cross_attn = mask_cls(d_model, n_level, n_head, 14)
            self.cross_attn.inferencing = False
        else:
            self.cross_attn = attn_cls(d_model, n_level, n_head)
        self.linear1, self.linear2 = nn.Linear(d_model, d_ffn), nn.Linear(d_ffn, d_model)
        self.norm1, self.norm2, self.norm3 = (nn.LayerNorm(d_model) for _ in range(3))

    def forward(self, tgt, query_pos, memory, shapes, mask, lsi, ratios, ref_windows):
        qk = (tgt + query_pos).transpose(0, 1)
        tgt = self.norm1(tgt + self.self_attn(qk, qk, tgt.transpose(0, 1))[0].transpose(0, 1))
        res = self.cross_attn(tgt + query_pos, memory, shapes, mask, lsi, ratios, ref_windows)
        tgt = self.norm2(tgt + res[0])
        tgt = self.norm3(tgt + self.linear2(F.relu(self.linear1(tgt))))
        return (tgt, res[1]) if self.use_mask else (tgt, None)


class SyntheticBoxeR2D(nn.Module):
    def __init__(self, attn_cls, levels, d_model=256, n_head=8, d_ffn=1024, n_enc=6, n_dec=6,
                 n_query=300, n_class=91, mask_cls=None):
        super().__init__()
        self.levels = levels
        n_level = len(levels)
        self.encoder = nn.ModuleList(EncoderLayer(attn_cls, d_model, n_head, n_level, d_ffn)
                                     for _ in range(n_enc))
        self.decoder = nn.ModuleList(DecoderLayer(attn_cls, d_model, n_head, n_level, d_ffn, mask_cls)
                                     for _ in range(n_dec))
        self.roi_head = nn.Linear(d_model, 1) if mask_cls is not None else None
        self.query_embed = nn.Embedding(n_query, d_model)
        self.query_pos = nn.Embedding(n_query, d_model)
        self.query_ref = nn.Embedding(n_query, 4)                 # logits of (cx, cy, w, h)
        self.class_head, self.box_head = nn.Linear(d_model, n_class), nn.Linear(d_model, 4)
        shapes = torch.tensor(levels, dtype=torch.long)
        self.register_buffer("shapes", shapes)
        self.register_buffer("lsi", torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1])))
        self.register_buffer("enc_ref", self._pixel_windows(levels))

    @staticmethod
    def _pixel_windows(levels, ref_size=4.0):
        """One window per pixel of every level, centred on it, ref_size pixels wide
        (box_transformer.py:70-116 without padding)."""
        out = []
        for (h, w) in levels:
            ys = (torch.arange(1, h + 1, dtype=torch.float32) - 0.5) / h
            xs = (torch.arange(1, w + 1, dtype=torch.float32) - 0.5) / w
            cy, cx = torch.meshgrid(ys, xs, indexing="ij")
            size = torch.tensor([ref_size / w, ref_size / h]).expand(h * w, 2)
            out.append(torch.cat([cx.reshape(-1, 1), cy.reshape(-1, 1), size], dim=1))
        return torch.cat(out, dim=0)

    def forward(self, src, pos):
        b = src.size(0)
        args = (self.shapes, None, self.lsi, None)
        memory = src
        enc_ref = self.enc_ref[None].expand(b, -1, -1)
        for layer in self.encoder:
            memory = layer(memory, pos, *args, enc_ref)
        tgt = self.query_embed.weight[None].expand(b, -1, -1)
        qpos = self.query_pos.weight[None].expand(b, -1, -1)
        ref = self.query_ref.weight.sigmoid()[None].expand(b, -1, -1)
        roi = None
        for layer in self.decoder:
            tgt, roi = layer(tgt, qpos, memory, *args, ref)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"],
                    help="bf16: autocast for the dense layers + the operator's native bf16 mode")
    ap.add_argument("--fused-grid", action="store_true")
    ap.add_argument("--mask-decoder", action="store_true",
                    help="InstanceAttention (14x14) in the decoder: the instance-segmentation model")
    ap.add_argument("--layers", type=int, default=6)
    args = ap.parse_args()

    from boxer_amd import BoxAttention, InstanceAttention, _lib
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
    model = SyntheticBoxeR2D(BoxAttention, LEVELS_COCO, n_enc=args.layers, n_dec=args.layers,
                             mask_cls=InstanceAttention if args.mask_decoder else None).to(device)
    for m in model.modules():
        if isinstance(m, (BoxAttention, InstanceAttention)):
            m.native_bf16 = args.dtype == "bf16"
            m.fused_grid = args.fused_grid
            with torch.no_grad():                            # trained-like box offsets
                m.linear_box_weight.normal_(0, 0.02)
    n_params = sum(p.numel() for p in model.parameters())
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    batch = make_batch(LEVELS_COCO, args.batch, 256, device, seed=100 + rank)
    amp = torch.bfloat16 if args.dtype == "bf16" else None

    for _ in range(args.warmup):
        train_step(model, opt, batch, amp)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, opt, batch, amp)
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], device=device)
    if dist is not None:
        dist.barrier()
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    ms_step = float(elapsed) / args.steps * 1e3

    # operator share: HIP events around every kernel launch of the library, two more steps
    _lib.profile_begin()
    for _ in range(2):
        train_step(model, opt, batch, amp)
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    op_ms = sum((v["ms"] or 0.0) * v["launches"] for v in prof.values()) / 2

    if rank == 0:
        print(json.dumps({
            "metric": "synthetic BoxeR-2D training step (6+6 layers, COCO 1333x800 shapes)",
            "ms_per_step": round(ms_step, 3), "images_per_s": round(world * args.batch / ms_step * 1e3, 2),
            "n_gpus": world, "batch_per_gpu": args.batch, "dtype": args.dtype,
            "fused_grid": args.fused_grid, "mask_decoder": args.mask_decoder, "params_M": round(n_params / 1e6, 2),
            "operator_kernels_ms_per_step": round(op_ms, 3),
            "operator_share": round(op_ms / ms_step, 3), "loss": round(float(loss), 4),
            "data": "synthetic", "scaling": "weak"}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
