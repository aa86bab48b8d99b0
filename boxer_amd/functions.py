"""autograd Functions with the reference's signatures.

``BoxAttnFunction`` / ``InstanceAttnFunction`` mirror
e2edet/module/ops/box_attention_func.py:9-64 and :67-150 argument for argument:
same positional inputs, same outputs, gradients only for value / sampling locations /
attention weights (``None`` for the rest), ``once_differentiable``, and the same AMP
contract (``custom_fwd(cast_inputs=torch.float32)``: under autocast every floating input
is cast to float32 and the op runs with autocast disabled).

``BoxAttnBF16Function`` / ``InstanceAttnBF16Function`` are the new native-bf16 mode
(BASELINE.json configs[1]): ``value`` and the upstream gradients are bfloat16, locations and
weights stay float32, accumulation is float32.  Same signatures otherwise.
"""
import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


# Whether the Functions' forward already prepares the backward's plan: the count pass and the scans of
# the destination-binned backward ride in the forward kernel's launch (ops.*_forward_train) and leave a
# small plan buffer (boxattn_plan_bytes: 1.4 MB at BoxeR-R50 encoder shapes) that lives until the matching
# backward; the backward then starts with the point-gradient kernel (fill pass riding along) instead of
# with three launches of binning.  On by default: this is the path bench.py times.  Off: the backward
# plans for itself (same results).
PLAN_IN_FORWARD = True


def set_plan_in_forward(flag):
    """-> previous setting."""
    global PLAN_IN_FORWARD
    old, PLAN_IN_FORWARD = PLAN_IN_FORWARD, bool(flag)
    return old


def _box_forward(ctx, value, shapes, lsi, loc, attn, im2col_step):
    """Training forward (also prepares the backward's plan) when asked to (PLAN_IN_FORWARD) and a
    gradient will be needed."""
    if PLAN_IN_FORWARD and any(ctx.needs_input_grad):
        return ops.box_attn_forward_train(value, shapes, lsi, loc, attn, im2col_step)
    return ops.box_attn_forward(value, shapes, lsi, loc, attn, im2col_step), None


def _inst_forward(ctx, value, shapes, lsi, loc, sw, lw, im2col_step):
    if PLAN_IN_FORWARD and any(ctx.needs_input_grad):
        return ops.instance_attn_forward_train(value, shapes, lsi, loc, sw, lw, im2col_step)
    return ops.instance_attn_forward(value, shapes, lsi, loc, sw, lw, im2col_step), None


def _box_backward(ctx, grad_output):
    if not grad_output.is_contiguous():
        grad_output = grad_output.contiguous()
    value, shapes, lsi, loc, attn = ctx.saved_tensors
    grad_value, grad_loc, grad_attn = ops.box_attn_backward(
        value, shapes, lsi, loc, attn, grad_output, ctx.im2col_step, plan=ctx.plan)
    ctx.plan = None
    return grad_value, None, None, grad_loc.to(ctx.loc_dtype), grad_attn.to(ctx.attn_dtype), None


def _inst_backward(ctx, grad_output, grad_mask_output):
    if not grad_output.is_contiguous():
        grad_output = grad_output.contiguous()
    if not grad_mask_output.is_contiguous():
        grad_mask_output = grad_mask_output.contiguous()
    value, shapes, lsi, loc, sw, lw = ctx.saved_tensors
    grad_value, grad_loc, grad_sw, grad_lw = ops.instance_attn_backward(
        value, shapes, lsi, loc, sw, lw, grad_output, grad_mask_output, ctx.im2col_step,
        plan=ctx.plan)
    ctx.plan = None
    return (grad_value, None, None, grad_loc.to(ctx.loc_dtype), grad_sw.to(ctx.w_dtype),
            grad_lw.to(ctx.w_dtype), None, None)


class BoxAttnFunction(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.loc_dtype, ctx.attn_dtype = sampling_locations.dtype, attention_weights.dtype
        output, ctx.plan = _box_forward(ctx, value, value_spatial_shapes,
                                        value_level_start_index, sampling_locations,
                                        attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @custom_bwd(device_type="cuda")
    @once_differentiable
    def backward(ctx, grad_output):
        return _box_backward(ctx, grad_output)


class InstanceAttnFunction(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                spatial_attention_weights, level_attention_weights, mask_size, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.loc_dtype, ctx.w_dtype = sampling_locations.dtype, spatial_attention_weights.dtype
        (output, mask_output), ctx.plan = _inst_forward(
            ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
            spatial_attention_weights, level_attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, spatial_attention_weights,
                              level_attention_weights)
        b, l, _, c = mask_output.shape
        return output, mask_output.view(b, l, mask_size, mask_size, c)

    @staticmethod
    @custom_bwd(device_type="cuda")
    @once_differentiable
    def backward(ctx, grad_output, grad_mask_output):
        return _inst_backward(ctx, grad_output, grad_mask_output)


def _to_bf16_args(value, loc, *weights):
    return (value.to(torch.bfloat16).contiguous(), loc.float().contiguous(),
            *[w.float().contiguous() for w in weights])


class BoxAttnBF16Function(Function):
    """Native-bf16 flavour: output and grad_value are bfloat16, accumulation is float32."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.loc_dtype, ctx.attn_dtype = sampling_locations.dtype, attention_weights.dtype
        ctx.value_dtype = value.dtype
        value, loc, attn = _to_bf16_args(value, sampling_locations, attention_weights)
        output, ctx.plan = _box_forward(ctx, value, value_spatial_shapes,
                                        value_level_start_index, loc, attn, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, loc, attn)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        grads = _box_backward(ctx, grad_output.to(torch.bfloat16))
        return (grads[0].to(ctx.value_dtype),) + grads[1:]


class InstanceAttnBF16Function(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                spatial_attention_weights, level_attention_weights, mask_size, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.loc_dtype, ctx.w_dtype = sampling_locations.dtype, spatial_attention_weights.dtype
        ctx.value_dtype = value.dtype
        value, loc, sw, lw = _to_bf16_args(value, sampling_locations, spatial_attention_weights,
                                           level_attention_weights)
        (output, mask_output), ctx.plan = _inst_forward(
            ctx, value, value_spatial_shapes, value_level_start_index, loc, sw, lw, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, loc, sw, lw)
        b, l, _, c = mask_output.shape
        return output, mask_output.view(b, l, mask_size, mask_size, c)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, grad_mask_output):
        grads = _inst_backward(ctx, grad_output.to(torch.bfloat16),
                               grad_mask_output.to(torch.bfloat16))
        return (grads[0].to(ctx.value_dtype),) + grads[1:]


class BoxGridFunction(Function):
    """(ref_windows, offsets, kernel_indices, valid_ratios, angle_mode) -> sampling grid
    (B,Lq,H,L,P,2): everything of the modules' ``_where_to_attend`` after the offset projection
    in one kernel each way (``module.fused_grid = True``; see ``ops.box_grid_forward``).
    Gradients for ``offsets`` and, if it requires one, ``ref_windows``."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, ref_windows, offsets, kernel_indices, valid_ratios, angle_mode):
        ref_windows = ref_windows.contiguous()
        offsets = offsets.contiguous()
        kernel_indices = kernel_indices.contiguous()
        if valid_ratios is not None:
            valid_ratios = valid_ratios.contiguous()
        ctx.save_for_backward(ref_windows, offsets, kernel_indices, valid_ratios)
        ctx.angle_mode = angle_mode
        return ops.box_grid_forward(ref_windows, offsets, kernel_indices, valid_ratios, angle_mode)

    @staticmethod
    @once_differentiable
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_grid):
        ref_windows, offsets, kernel_indices, valid_ratios = ctx.saved_tensors
        need_ref = ctx.needs_input_grad[0]
        grad_offsets, rows = ops.box_grid_backward(
            ref_windows, offsets, kernel_indices, valid_ratios, ctx.angle_mode,
            grad_grid.contiguous().float(), need_ref_grad=need_ref)
        grad_ref = None
        if need_ref:
            # rows: d/d(cx, cy, w, h, angle)_ref per (head, level); the windows are shared by
            # the levels and, unless given per head, by the heads
            rows = rows.sum(dim=3) if ref_windows.dim() == 4 else rows.sum(dim=(2, 3))
            grad_ref = torch.zeros_like(ref_windows)
            n = min(5, ref_windows.size(-1))
            grad_ref[..., :n] = rows[..., :n]
        return grad_ref, grad_offsets, None, None, None


class LogitSoftmaxFunction(Function):
    """softmax over the last axis in float32 (``module.fused_pointwise``): one HIP pass each way
    instead of autocast's cast + softmax (+ cast back); see ``ops.softmax_forward``."""

    @staticmethod
    def forward(ctx, logits):
        logits = logits.contiguous()
        attn = ops.softmax_forward(logits)
        ctx.save_for_backward(attn)
        ctx.logits_dtype = logits.dtype
        return attn

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_attn):
        (attn,) = ctx.saved_tensors
        return ops.softmax_backward(attn, grad_attn.contiguous().float(), ctx.logits_dtype)


class ValueMaskCastFunction(Function):
    """value -> bfloat16 with padded rows zeroed (``ops.value_mask_cast``); the gradient is the
    upstream one with the same rows zeroed, in the input's type."""

    @staticmethod
    def forward(ctx, value, v_mask):
        ctx.save_for_backward(v_mask)
        ctx.value_dtype = value.dtype
        return ops.value_mask_cast(value.contiguous(), v_mask)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        (v_mask,) = ctx.saved_tensors
        grad = grad.to(ctx.value_dtype)
        if v_mask is not None:
            grad = grad.masked_fill(v_mask[..., None], 0)
        return grad, None


def _ref_grad(ref_windows, rows):
    """(B,Lq,H,L,5) row gradients -> gradient of the reference windows (shared by the levels
    and, unless given per head, by the heads)."""
    rows = rows.sum(dim=3) if ref_windows.dim() == 4 else rows.sum(dim=(2, 3))
    grad_ref = torch.zeros_like(ref_windows)
    n = min(5, ref_windows.size(-1))
    grad_ref[..., :n] = rows[..., :n]
    return grad_ref
