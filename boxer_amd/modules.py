"""nn.Modules with the surface of the reference's ``e2edet/module/box_attention.py``.

``BoxAttention`` (box_attention.py:140-239), ``InstanceAttention`` (:10-137) and
``Box3dAttention`` (:242-363): same constructor arguments, same parameter / buffer names and
shapes (so reference checkpoints load with ``load_state_dict``), same initialisation, same
``forward(query, value, v_shape, v_mask, v_start_index, v_valid_ratios, ref_windows)``
signature and return tuples.  The sampling itself runs in the HIP operator behind
``BoxAttnFunction`` / ``InstanceAttnFunction``.

The three reference classes repeat their code; here the shared parts (parameters, kernel
grid, box decoding, value projection) live in one base class.

Extra, not in the reference: ``native_bf16`` (default False).  When set, the op runs in the
bf16 storage mode (value / output bfloat16, fp32 locations, weights and accumulation)
instead of the reference's "always float32" contract.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dense
from .functions import (BoxAttnBF16Function, BoxAttnFunction, BoxGridFunction,
                        InstanceAttnBF16Function, InstanceAttnFunction, LogitSoftmaxFunction,
                        ValueMaskCastFunction)


def _kernel_offsets(kernel_size, divisor):
    """(k*k, 2) grid of (x, y) offsets in units of the box size, row-major over (y, x).
    Even k: half-pixel centred (-k/2+0.5 .. k/2-0.5); odd k: integer centred."""
    half = (kernel_size - 1) / 2.0
    ticks = torch.linspace(-half, half, kernel_size)
    ys, xs = torch.meshgrid(ticks, ticks, indexing="ij")
    return torch.stack([xs, ys], dim=-1).reshape(-1, 2) / divisor


class _BoxAttentionBase(nn.Module):
    """Parameters and geometry shared by the three attention flavours."""

    def __init__(self, d_model, num_level, num_head, kernel_size, box_vars, attn_points,
                 offset_divisor):
        super().__init__()
        assert d_model % num_head == 0, "d_model should be divided by num_head"
        self.im2col_step = 64
        self.d_model = d_model
        self.num_head = num_head
        self.num_level = num_level
        self.head_dim = d_model // num_head
        self.kernel_size = kernel_size
        self.native_bf16 = False
        # opt-in: box -> grid expansion in one HIP kernel each way (BoxGridFunction).  (Until round
        # 5 the value 2 built the grid inside the sampling kernels; it never removed the grid tensor -- both bin passes
        # and the point gradients read it -- and was no faster than the one-kernel grid build: removed in round 6, any
        # true value means the grid kernels.)
        self.fused_grid = False
        # opt-in: the softmax over the L*P logits and, in the bf16 storage mode, the value
        # mask-fill + bf16 cast as single HIP passes (LogitSoftmaxFunction, ValueMaskCastFunction)
        self.fused_pointwise = False

        self.linear_box_weight = nn.Parameter(torch.zeros(num_level * num_head * box_vars, d_model))
        self.linear_box_bias = nn.Parameter(torch.zeros(num_head * num_level * box_vars))
        self.linear_attn_weight = nn.Parameter(
            torch.zeros(num_head * num_level * attn_points, d_model))
        self.linear_attn_bias = nn.Parameter(torch.zeros(num_head * num_level * attn_points))
        self.value_proj = nn.Linear(d_model, d_model)
        self.out_proj = nn.Linear(d_model, d_model)
        self.register_buffer("kernel_indices", _kernel_offsets(kernel_size, offset_divisor))
        self._reset_parameters()

    def _reset_parameters(self):
        for proj in (self.out_proj, self.value_proj):
            nn.init.xavier_uniform_(proj.weight)
            nn.init.constant_(proj.bias, 0.0)
        nn.init.constant_(self.linear_attn_weight, 0.0)
        nn.init.constant_(self.linear_attn_bias, 0.0)
        nn.init.constant_(self.linear_box_weight, 0.0)
        nn.init.uniform_(self.linear_box_bias)

    # -- pieces of forward -------------------------------------------------------------
    def _project_value(self, value, v_mask):
        b, s = value.shape[:2]
        value = dense.linear(value, self.value_proj.weight, self.value_proj.bias)
        if (self.fused_pointwise and self.native_bf16 and value.is_cuda and
                value.dtype in (torch.float32, torch.bfloat16) and self.d_model % 8 == 0 and
                (v_mask is not None or value.dtype != torch.bfloat16)):   # else: nothing to do
            value = ValueMaskCastFunction.apply(value, v_mask)
        elif v_mask is not None:
            value = value.masked_fill(v_mask[..., None], float(0))
        return value.view(b, s, self.num_head, self.head_dim)

    def _softmax(self, logits):
        """softmax over the last axis (the L * P logits of a (query, head))."""
        if (self.fused_pointwise and logits.is_cuda and logits.size(-1) <= 64 and
                logits.dtype in (torch.float32, torch.bfloat16)):
            return LogitSoftmaxFunction.apply(logits)
        return F.softmax(logits, dim=-1)

    def _box_offsets(self, query, ref_windows, n_vars):
        b, l = ref_windows.shape[:2]
        off = dense.linear(query, self.linear_box_weight, self.linear_box_bias)
        return off.view(b, l, self.num_head, self.num_level, n_vars)

    @staticmethod
    def _per_head_level(ref_windows):
        """(B,Lq,D) -> (B,Lq,1,1,D);  (B,Lq,H,D) -> (B,Lq,H,1,D)."""
        if ref_windows.dim() == 3:
            return ref_windows[:, :, None, None]
        return ref_windows[:, :, :, None]

    @staticmethod
    def _decode_boxes(ref_boxes, offset_boxes):
        """ref (cx,cy,w,h) + offset/8 scaled by the reference size -> centre, size with a
        points axis: (..., 1, 2) each."""
        wh = ref_boxes[..., 2:4]
        boxes = ref_boxes + offset_boxes / 8 * torch.cat([wh, wh], dim=-1)
        boxes = boxes.unsqueeze(-2)
        return boxes[..., :2], boxes[..., 2:]

    def _box_function(self):
        return BoxAttnBF16Function if self.native_bf16 else BoxAttnFunction

    def _use_fused_grid(self, query, v_valid_ratios):
        """The fused kernel covers the reference's call shapes: CUDA tensors and
        ``v_valid_ratios`` None or (B,1,1,L,1,2) (box_transformer.py:118-138)."""
        if not (self.fused_grid and query.is_cuda) or query.dtype == torch.float64:
            return False                      # (the kernel computes in float32)
        return v_valid_ratios is None or (
            v_valid_ratios.dim() == 6
            and tuple(v_valid_ratios.shape[1:]) == (1, 1, self.num_level, 1, 2)
            and v_valid_ratios.size(0) == query.size(0))


class BoxAttention(_BoxAttentionBase):
    def __init__(self, d_model, num_level, num_head, kernel_size=2):
        super().__init__(d_model, num_level, num_head, kernel_size, box_vars=4,
                         attn_points=kernel_size ** 2, offset_divisor=kernel_size)
        self.num_point = kernel_size ** 2

    def _where_to_attend(self, query, v_valid_ratios, ref_windows):
        offset_boxes = self._box_offsets(query, ref_windows, 4)
        if self._use_fused_grid(query, v_valid_ratios):
            return BoxGridFunction.apply(ref_windows, offset_boxes, self.kernel_indices,
                                         v_valid_ratios, 0)
        center, size = self._decode_boxes(self._per_head_level(ref_windows), offset_boxes)
        grid = center + self.kernel_indices * torch.relu(size)
        if v_valid_ratios is not None:
            grid = grid * v_valid_ratios
        return grid.contiguous()

    def _softmax_weights(self, query):
        b, l1 = query.shape[:2]
        w = dense.linear(query, self.linear_attn_weight, self.linear_attn_bias)
        w = self._softmax(w.view(b, l1, self.num_head, -1))
        return w.view(b, l1, self.num_head, self.num_level, self.kernel_size, self.kernel_size)

    def _angle_mode(self):
        return 0

    def forward(self, query, value, v_shape, v_mask, v_start_index, v_valid_ratios, ref_windows):
        value = self._project_value(value, v_mask)
        attn_weights = self._softmax_weights(query)
        sampled_grid = self._where_to_attend(query, v_valid_ratios, ref_windows)
        output = self._box_function().apply(value, v_shape, v_start_index, sampled_grid,
                                            attn_weights, self.im2col_step)
        return dense.linear(output, self.out_proj.weight, self.out_proj.bias), attn_weights


class Box3dAttention(BoxAttention):
    def __init__(self, d_model, num_level, num_head, with_rotation=True, kernel_size=2):
        # note: the 3D module scales the kernel grid by 1/2 whatever the kernel size
        # (box_attention.py:291)
        _BoxAttentionBase.__init__(self, d_model, num_level, num_head, kernel_size,
                                   box_vars=5 if with_rotation else 4,
                                   attn_points=kernel_size ** 2, offset_divisor=2)
        self.with_rotation = with_rotation
        self.num_variable = 5 if with_rotation else 4
        self.num_point = kernel_size ** 2

    def _angle_mode(self):
        return 1 if self.with_rotation else 2

    def _where_to_attend(self, query, v_valid_ratios, ref_windows):
        b, l = ref_windows.shape[:2]
        offsets = self._box_offsets(query, ref_windows, self.num_variable)
        if self._use_fused_grid(query, v_valid_ratios):
            return BoxGridFunction.apply(ref_windows, offsets, self.kernel_indices,
                                         v_valid_ratios, 1 if self.with_rotation else 2)
        ref = self._per_head_level(ref_windows)       # (cx,cy,w,h,angle[,vx,vy])
        ref_boxes, ref_angles = ref[..., :4], ref[..., 4:5]
        if self.with_rotation:
            angles = (ref_angles + offsets[..., 4:5] / 16) * 2 * math.pi
            offsets = offsets[..., :4]
        else:
            angles = ref_angles.expand(b, l, self.num_head, self.num_level, 1)
        center, size = self._decode_boxes(ref_boxes, offsets)

        cos, sin = torch.cos(angles), torch.sin(angles)           # (B,Lq,H,L,1)
        local = self.kernel_indices * torch.relu(size)            # (B,Lq,H,L,P,2), box frame
        lx, ly = local[..., 0], local[..., 1]
        rotated = torch.stack([lx * cos - ly * sin, lx * sin + ly * cos], dim=-1)
        grid = center + rotated
        if v_valid_ratios is not None:
            grid = grid * v_valid_ratios
        return grid.contiguous()


class InstanceAttention(_BoxAttentionBase):
    def __init__(self, d_model, num_level, num_head, kernel_size):
        # attention logits are predicted on a 2x2 grid per level and replicated to k x k
        super().__init__(d_model, num_level, num_head, kernel_size, box_vars=4, attn_points=4,
                         offset_divisor=kernel_size)

    _where_to_attend = BoxAttention._where_to_attend

    def forward(self, query, value, v_shape, v_mask, v_start_index, v_valid_ratios, ref_windows):
        b, l1 = query.shape[:2]
        k = self.kernel_size
        value = self._project_value(value, v_mask)

        logits = dense.linear(query, self.linear_attn_weight, self.linear_attn_bias)
        logits = logits.view(b, l1, self.num_head, self.num_level, 2, 2)
        logits = logits.repeat_interleave(k // 2, dim=-1).repeat_interleave(k // 2, dim=-2)
        spatial_attn_weights = F.softmax(logits.reshape(b, l1, self.num_head, -1), dim=-1).view(
            b, l1, self.num_head, self.num_level, k, k)

        sampled_grid = self._where_to_attend(query, v_valid_ratios, ref_windows)

        # `inferencing` is injected by the model (base_model.py:49-67); like the reference,
        # a bare module without it raises AttributeError here.
        if not self.inferencing:
            level_attn_weights = F.softmax(
                logits.view(b, l1, self.num_head, self.num_level, k, k), dim=3)
            fn = InstanceAttnBF16Function if self.native_bf16 else InstanceAttnFunction
            output, mask_output = fn.apply(value, v_shape, v_start_index, sampled_grid,
                                           spatial_attn_weights, level_attn_weights, k,
                                           self.im2col_step)
            attn_weights = (spatial_attn_weights, level_attn_weights)
            mask_output = self.out_proj(mask_output)
        else:
            output = self._box_function().apply(value, v_shape, v_start_index, sampled_grid,
                                                spatial_attn_weights, self.im2col_step)
            attn_weights = (spatial_attn_weights,)
            mask_output = None
        return self.out_proj(output), mask_output, attn_weights
