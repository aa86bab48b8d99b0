"""The callers of the operator inside BoxeR's transformers: the
encoder / decoder LAYERS of BoxeR-2D (e2edet/module/box_transformer.py:316-465) and BoxeR-3D
(e2edet/module/box3d_transformer.py:230-322), the reference-window builders of the two
encoders (box_transformer.py:70-116, box3d_transformer.py:62-109) and the pillar-to-BEV scatter
that produces the 3D model's feature map (point_pillar.py:8-67).

Same constructor arguments, sub-module names (``self_attn``, ``multihead_attn``, ``linear1`` ...)
and forward signatures as the reference, so its checkpoints load with ``load_state_dict`` and
the wiring is pinned by the G8 goldens (tests/golden/make_goldens.py:g8, run on the reference's
own classes).  Everything here is ordinary PyTorch around ``boxer_amd.modules``; the synthetic
training step (bench_train.py) stacks these layers.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dense
from .modules import Box3dAttention, BoxAttention, InstanceAttention


def _activation(name):
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    if name == "glu":
        return F.glu
    raise RuntimeError("activation should be relu/gelu, not %s." % name)


def _with_pos(tensor, pos):
    return tensor if pos is None else tensor + pos


class _LayerBase(nn.Module):
    """Post-norm residual blocks: x = LN(x + sublayer(x)), FFN = linear2(act(linear1(x)))."""

    def _ffn_params(self, d_model, dim_feedforward, dropout, activation, n_norm):
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        for i in range(1, n_norm + 1):
            setattr(self, "norm%d" % i, nn.LayerNorm(d_model))
            setattr(self, "dropout%d" % i, nn.Dropout(dropout))
        self.activation = _activation(activation)

    def _ffn(self, x):
        hidden = self.activation(dense.linear(x, self.linear1.weight, self.linear1.bias))
        return dense.linear(self.dropout(hidden), self.linear2.weight, self.linear2.bias)

    def _self_attention(self, tgt, query_pos):
        """nn.MultiheadAttention over the object queries (sequence-first), residual + norm1."""
        qk = _with_pos(tgt, query_pos).transpose(0, 1)
        tgt2 = self.self_attn(qk, qk, tgt.transpose(0, 1))[0].transpose(0, 1)
        return self.norm1(tgt + self.dropout1(tgt2))


class BoxTransformerEncoderLayer(_LayerBase):
    """box_transformer.py:316-365."""

    def __init__(self, d_model, nhead, nlevel, dim_feedforward, dropout, activation):
        super().__init__()
        self.self_attn = BoxAttention(d_model, nlevel, nhead)
        self._ffn_params(d_model, dim_feedforward, dropout, activation, 2)

    def forward(self, src, pos, src_shape, src_mask, src_start_index, src_valid_ratios,
                ref_windows):
        src2 = self.self_attn(_with_pos(src, pos), src, src_shape, src_mask, src_start_index,
                              src_valid_ratios, ref_windows)[0]
        src = self.norm1(src + self.dropout1(src2))
        return self.norm2(src + self.dropout2(self._ffn(src)))


class BoxTransformerDecoderLayer(_LayerBase):
    """box_transformer.py:368-465.  ``use_mask``: InstanceAttention (14 x 14) instead of
    BoxAttention; in training it also returns the per-point RoI features, which go through the
    same residual / norm / FFN blocks (``residual_mode`` "v1" or "v2")."""

    def __init__(self, d_model, nhead, nlevel, dim_feedforward, dropout, activation, use_mask,
                 residual_mode):
        super().__init__()
        assert residual_mode in ("v1", "v2")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        if use_mask:
            self.multihead_attn = InstanceAttention(d_model, nlevel, nhead, 14)
        else:
            self.multihead_attn = BoxAttention(d_model, nlevel, nhead)
        self._ffn_params(d_model, dim_feedforward, dropout, activation, 3)
        self.use_mask = use_mask
        self.residual_mode = residual_mode

    def forward(self, tgt, query_pos, memory, memory_shape, memory_mask, memory_start_index,
                memory_valid_ratios, ref_windows):
        tgt = self._self_attention(tgt, query_pos)
        res = self.multihead_attn(_with_pos(tgt, query_pos), memory, memory_shape, memory_mask,
                                  memory_start_index, memory_valid_ratios, ref_windows)
        # `inferencing` is injected by the model (base_model.py:49-67), as in the reference
        with_roi = self.use_mask and not self.inferencing
        roi = res[1] if with_roi else None
        tgt = self.norm2(tgt + self.dropout2(res[0]))
        if with_roi:
            roi = self.norm2(tgt.unsqueeze(-2).unsqueeze(-2) + self.dropout2(roi))
        tgt = self.norm3(tgt + self.dropout3(self._ffn(tgt)))
        if with_roi:
            if self.residual_mode == "v1":
                roi = roi + self.dropout3(self._ffn(roi))
            else:
                roi = tgt.unsqueeze(-2).unsqueeze(-2) + self.dropout2(roi)
            roi = self.norm3(roi)
        return tgt, roi


class Box3dTransformerEncoderLayer(_LayerBase):
    """box3d_transformer.py:230-266: Box3dAttention with fixed per-head angles."""

    def __init__(self, d_model, nhead, nlevel, dim_feedforward, dropout, activation):
        super().__init__()
        self.self_attn = Box3dAttention(d_model, nlevel, nhead, with_rotation=False)
        self._ffn_params(d_model, dim_feedforward, dropout, activation, 2)

    def forward(self, src, pos, src_shape, src_start_idx, ref_windows):
        src2 = self.self_attn(_with_pos(src, pos), src, src_shape, None, src_start_idx, None,
                              ref_windows)[0]
        src = self.norm1(src + self.dropout1(src2))
        return self.norm2(src + self.dropout2(self._ffn(src)))


class Box3dTransformerDecoderLayer(_LayerBase):
    """box3d_transformer.py:269-322: learned rotation in the cross-attention."""

    def __init__(self, d_model, nhead, nlevel, dim_feedforward, dropout, activation):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = Box3dAttention(d_model, nlevel, nhead, with_rotation=True)
        self._ffn_params(d_model, dim_feedforward, dropout, activation, 3)

    def forward(self, tgt, query_pos, memory, memory_shape, memory_start_idx, ref_windows):
        tgt = self._self_attention(tgt, query_pos)
        tgt2 = self.multihead_attn(_with_pos(tgt, query_pos), memory, memory_shape, None,
                                   memory_start_idx, None, ref_windows)[0]
        tgt = self.norm2(tgt + self.dropout2(tgt2))
        return self.norm3(tgt + self.dropout3(self._ffn(tgt)))


# --------------------------------------------------------------------------------------
# reference windows of the two encoders
# --------------------------------------------------------------------------------------
def encoder_ref_windows_2d(levels, batch, device=None, dtype=torch.float32, ref_size=4.0,
                           masks=None, eps=1e-6):
    """(B, S, 4) windows (cx, cy, w, h), one per pixel of every level, centred on it and
    ``ref_size`` pixels wide in the valid (unpadded) part of the level
    (box_transformer.py:70-116).  ``masks``: per level (B, H_l, W_l) bool padding masks or None."""
    out = []
    for i, (h, w) in enumerate(levels):
        if masks is not None:
            valid = ~masks[i]
            y = valid.cumsum(1, dtype=dtype)
            x = valid.cumsum(2, dtype=dtype)
            size_h = valid[:, :, 0].sum(-1, dtype=dtype)
            size_w = valid[:, 0, :].sum(-1, dtype=dtype)
        else:
            ys = torch.arange(1, h + 1, dtype=dtype, device=device)
            xs = torch.arange(1, w + 1, dtype=dtype, device=device)
            y, x = torch.meshgrid(ys, xs, indexing="ij")
            y, x = y[None].repeat(batch, 1, 1), x[None].repeat(batch, 1, 1)
            size_h = torch.full((batch,), float(h), dtype=dtype, device=device)
            size_w = torch.full((batch,), float(w), dtype=dtype, device=device)
        y = (y - 0.5) / (y[:, -1:, :] + eps)
        x = (x - 0.5) / (x[:, :, -1:] + eps)
        center = torch.stack([x, y], -1).flatten(1, 2)
        size = torch.stack([ref_size / size_w, ref_size / size_h], -1)
        out.append(torch.cat([center, size[:, None].expand_as(center)], -1))
    return torch.cat(out, 1)


# the 8 fixed per-head orientations of the BoxeR-3D encoder, as a fraction of a turn
# (box3d_transformer.py:63-75: angles {0, 2pi/3, -2pi/3, 0, 2pi/3, -2pi/3, 0, pi} through
# normalize_period(offset 0.5, period 2 pi))
_ANGLES_3D = [0.0, 2 * math.pi / 3, -2 * math.pi / 3, 0.0, 2 * math.pi / 3, -2 * math.pi / 3, 0.0,
              math.pi]


def encoder_ref_windows_3d(levels, batch, device=None, dtype=torch.float32, ref_size=4.0):
    """(B, S, 8, 5) windows (cx, cy, w, h, angle) per pixel and head (box3d_transformer.py:62-109)."""
    angle = (torch.tensor(_ANGLES_3D, dtype=dtype, device=device) + 0.5 * 2 * math.pi) / (2 * math.pi)
    out = []
    for (h, w) in levels:
        ys = (torch.arange(h, dtype=dtype, device=device) + 0.5) / h
        xs = (torch.arange(w, dtype=dtype, device=device) + 0.5) / w
        y, x = torch.meshgrid(ys, xs, indexing="ij")
        ones = torch.ones(h, w, 8, dtype=dtype, device=device)
        ref = torch.stack([x[..., None] * ones, y[..., None] * ones, ones * (ref_size / w),
                           ones * (ref_size / h), ones * angle], -1)
        out.append(ref.flatten(0, 1)[None].expand(batch, -1, -1, -1))
    return torch.cat(out, 1)


def pillar_scatter(voxel_features, coords, batch_size, nx, ny):
    """PointPillarsScatter (point_pillar.py:8-67): pillar features (N, C) with integer
    coordinates (N, 4) = (batch, z, y, x) -> dense BEV canvas (B, C, ny, nx), empty cells 0."""
    canvas = voxel_features.new_zeros(batch_size, voxel_features.size(1), ny * nx)
    idx = (coords[:, 2] * nx + coords[:, 3]).long()
    canvas[coords[:, 0].long(), :, idx] = voxel_features
    return canvas.view(batch_size, voxel_features.size(1), ny, nx)
