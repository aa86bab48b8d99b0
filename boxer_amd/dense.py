"""The S-long projections around the operator (value / output / offset / logit projections of
the attention modules, the FFN of the encoder layers): ``y = x W^T + b`` with tens of thousands
of rows and a 256 ... 1 024-wide weight.

Forward and the input gradient are ordinary GEMMs and stay with the BLAS library.  The WEIGHT
gradient ``dW = dY^T X`` contracts over the rows (K = B*S = 44 446 at COCO shapes) into a
256 x 256 ... 1 024 output -- 16-64 output tiles on 256 CUs in the library's default choice,
160-220 us each, 9.8 ms of a 25 ms BoxeR-2D training step (profiles/r02_train_step.log).
``linear()`` below computes it as a batched product over row chunks (split-K spelled as ``bmm``:
every chunk is its own set of output tiles) plus a float32 sum of the partial products, and takes
the bias gradient from the same pass.  Opt-in (``set_split_k(True)``; bench_train.py
--split-k-wgrad), same results to the rounding of the partial products; below ``MIN_ROWS`` rows
it is plain ``F.linear``.  Host-side scheduling of library GEMMs, no kernels of ours.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

SPLIT_K = False
MIN_ROWS = 8192          # below this the library's single GEMM is fine
CHUNK_ROWS = 1024        # rows per partial product


def set_split_k(flag):
    """-> previous setting."""
    global SPLIT_K
    old, SPLIT_K = SPLIT_K, bool(flag)
    return old


def _acc_dtype(dtype):
    return torch.float32 if dtype in (torch.bfloat16, torch.float16) else dtype


_BMM_OUT_DTYPE = None     # does torch.bmm take out_dtype on this build / device? (probed once)


def _bmm_acc(a, b, acc):
    """Batched product whose partial products leave in ``acc`` (float32 for bf16 / fp16 inputs): one
    rounding per dW element instead of one bf16 rounding per row chunk.  Builds whose ``bmm`` has no
    ``out_dtype`` fall back to storage-type partials: each of the n_chunk partials then carries a
    2^-9 relative rounding, i.e. |err(dW)| <= 2^-9 * sum_chunks |partial| (documented bound)."""
    global _BMM_OUT_DTYPE
    if a.dtype == acc:
        return torch.bmm(a, b)
    if _BMM_OUT_DTYPE is not False:
        try:
            out = torch.bmm(a, b, out_dtype=acc)
            _BMM_OUT_DTYPE = True
            return out
        except (TypeError, RuntimeError):
            if _BMM_OUT_DTYPE:           # worked before: a real error of this call
                raise
            _BMM_OUT_DTYPE = False
    return torch.bmm(a, b)


def _weight_grad(gy, x):
    """dY^T X over row chunks: (rows, out), (rows, in) -> (out, in), at least float32."""
    acc = _acc_dtype(gy.dtype)
    rows = gy.size(0)
    n_chunk = max(1, rows // CHUNK_ROWS)
    per = rows // n_chunk
    body = n_chunk * per
    part = _bmm_acc(gy[:body].view(n_chunk, per, -1).transpose(1, 2),
                    x[:body].view(n_chunk, per, -1), acc)
    gw = part.sum(0, dtype=acc)
    if body < rows:                                     # the ragged tail: a short ordinary GEMM
        gw = gw + (gy[body:].t() @ x[body:]).to(acc)
    return gw


class _SplitKLinear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        amp = torch.is_autocast_enabled("cuda")
        dt = torch.get_autocast_dtype("cuda") if amp else x.dtype
        x2 = x.reshape(-1, x.size(-1)).to(dt)
        w = weight.to(dt)
        ctx.save_for_backward(x2, w)
        ctx.in_dtype, ctx.w_dtype = x.dtype, weight.dtype
        ctx.has_bias = bias is not None
        y = F.linear(x2, w, None if bias is None else bias.to(dt))
        return y.view(*x.shape[:-1], weight.size(0))

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.size(-1)).to(x2.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (gy2 @ w).view(*gy.shape[:-1], w.size(1)).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(gy2, x2).to(ctx.w_dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy2.sum(0, dtype=_acc_dtype(gy2.dtype)).to(ctx.w_dtype)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """``F.linear`` whose weight gradient is a batched split-K product when it pays."""
    if (SPLIT_K and x.is_cuda and torch.is_grad_enabled() and weight.requires_grad and
            x.numel() // x.size(-1) >= MIN_ROWS):
        return _SplitKLinear.apply(x, weight, bias)
    return F.linear(x, weight, bias)
