"""Pure-PyTorch (grid_sample) statement of box / instance attention -- TEST INFRASTRUCTURE.

Role: the counterpart of the reference's test oracle (tests/box_attn_test.py:9-42,
tests/instance_attn_test.py:11-63 in /root/reference): a differentiable CPU formulation
built on ``F.grid_sample(bilinear, zeros, align_corners=False)``.  It is written from the
operator's specification, takes the op's native argument layout
(locations in [0, 1], value as (B, S, H, C)) and is used for two things only:

* ``tests/``: a second, independent checker next to the C oracle (autograd gives grads);
* ``bench.py``: the ``cpu_baseline`` leg ("the repo's pure-PyTorch fallback timed on the
  host CPU", BASELINE.json north_star).

``boxer_amd`` never imports it.  Pinned by tests/golden fixtures (tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as F


def _sample_levels(value, shapes, loc):
    """Yield per level the sampled values as (B, H, C, Lq, P)."""
    B, S, H, C = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    start = 0
    for lvl in range(L):
        hl, wl = int(shapes[lvl][0]), int(shapes[lvl][1])
        n = hl * wl
        # (B, n, H, C) -> (B*H, C, hl, wl)
        fmap = value[:, start:start + n].permute(0, 2, 3, 1).reshape(B * H, C, hl, wl)
        start += n
        # (B, Lq, H, P, 2) -> (B*H, Lq, P, 2); [0,1] -> [-1,1]
        grid = loc[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(B * H, Lq, P, 2)
        sampled = F.grid_sample(fmap, 2.0 * grid - 1.0, mode="bilinear",
                                padding_mode="zeros", align_corners=False)
        yield lvl, sampled.view(B, H, C, Lq, P)


def box_attn(value, shapes, loc, attn):
    """value (B,S,H,C); loc (B,Lq,H,L,P,2) in [0,1]; attn (B,Lq,H,L,...P) -> (B,Lq,H*C)."""
    B, S, H, C = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    attn = attn.reshape(B, Lq, H, L, P)
    out = value.new_zeros(B, Lq, H, C)
    for lvl, sampled in _sample_levels(value, shapes, loc):
        out = out + torch.einsum("bhcqp,bqhp->bqhc", sampled, attn[:, :, :, lvl])
    return out.reshape(B, Lq, H * C)


def instance_attn(value, shapes, loc, spatial_w, level_w):
    """-> out (B,Lq,H*C), mask_out (B,Lq,P,H*C) (native layout of the op)."""
    B, S, H, C = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    spatial_w = spatial_w.reshape(B, Lq, H, L, P)
    level_w = level_w.reshape(B, Lq, H, L, P)
    out = value.new_zeros(B, Lq, H, C)
    mask = value.new_zeros(B, Lq, P, H, C)
    for lvl, sampled in _sample_levels(value, shapes, loc):
        out = out + torch.einsum("bhcqp,bqhp->bqhc", sampled, spatial_w[:, :, :, lvl])
        mask = mask + torch.einsum("bhcqp,bqhp->bqphc", sampled, level_w[:, :, :, lvl])
    return out.reshape(B, Lq, H * C), mask.reshape(B, Lq, P, H * C)
