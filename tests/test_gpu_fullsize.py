"""Full-size GPU parity (``-m gpu``): every ``bench.py`` workload at BASELINE.json's sizes, through
the training entry points the bench times (``*_forward_train`` + ``*_backward(plan=...)``, i.e. the
C ABI's ``*_fwd_train_*`` / ``*_bwd_ws_*``), ALL outputs compared element-wise with the CPU oracle
(oracle/boxattn_oracle.c, fp64) on the same inputs.

Check per element (bench.parity_report, shared with bench.py's parity gate):
``|got - want| <= tol * (max(1, rms(want)) + |want|)`` with tol = 1e-4 (fp32 tensors) / 1e-2
(bf16 tensors) -- BASELINE.json's tolerances as an absolute + relative bound (the absolute part
follows the tensor's typical magnitude: an fp32 sum of 64 products of O(100) terms cannot be held
to 1e-4 absolute, a wrong small element among large ones still fails).  bf16 runs feed ``randn``
values rounded to bf16; the oracle gets exactly those rounded values.

Reference test this mirrors: tests/box_attn_test.py:96-159 (forward / backward allclose of the
CUDA op against the pure-PyTorch formulation).
"""
import numpy as np
import pytest
import torch

import bench
from oracle import boxattn_oracle as oc

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 1e-2}


def check(got, want, tol, what, ignore=None):
    got = got.detach().double().cpu().numpy().reshape(np.shape(want))
    want = np.asarray(want, dtype=np.float64)
    if ignore is not None:
        keep = ~np.broadcast_to(ignore, want.shape)
        got, want = got * keep, want * keep
    assert np.isfinite(got).all(), what
    scale = max(1.0, float(np.sqrt(np.mean(want * want)))) if want.size else 1.0
    ratio = np.abs(got - want) / (scale + np.abs(want))
    worst = float(ratio.max()) if want.size else 0.0
    assert worst <= tol, "%s: worst |err| / (%.3g + |want|) = %.3e > %.1e at %s" % (
        what, scale, worst, tol, np.unravel_index(int(ratio.argmax()), want.shape))


def run_workload(workload, dtype, family, batch=bench.BATCH):
    """Training forward + planned backward on the bench inputs -> bench.parity_report rows."""
    inp = bench.make_inputs(workload, dtype, "cuda", family=family, batch=batch, seed=0)
    plans = []
    from boxer_amd import ops
    orig = ops._forward_train

    def spy(*a, **k):                       # the plan the training forward hands to the backward
        plans.append(orig(*a, **k))
        return plans[-1]
    ops._forward_train = spy
    try:
        out, grads = bench.make_step(inp)()
    finally:
        ops._forward_train = orig
    torch.cuda.synchronize()
    return bench.parity_report(inp, out, grads), plans[-1]


CASES = [
    # workload, dtype, input family  (BASELINE.json configs[1], [2], [4])
    ("C2", torch.bfloat16, "model"), ("C2", torch.float32, "model"),
    ("C2", torch.bfloat16, "test"), ("C2", torch.float32, "test"),
    ("C2p", torch.bfloat16, "model"), ("C2p", torch.float32, "model"),
    ("C3", torch.float32, "model"), ("C3", torch.bfloat16, "model"),
    ("C3p", torch.float32, "model"), ("C3p", torch.bfloat16, "model"),
    ("C3pp", torch.float32, "model"),
    ("C5", torch.float32, "model"), ("C5", torch.bfloat16, "model"),
    ("C5p", torch.float32, "model"), ("C5p", torch.bfloat16, "model"),
    ("C5pp", torch.float32, "model"),
]


@pytest.mark.parametrize("workload,dtype,family", CASES,
                         ids=["%s-%s-%s" % (w, str(d).split(".")[-1], f) for w, d, f in CASES])
def test_bench_workload_matches_oracle(workload, dtype, family):
    report, plan = run_workload(workload, dtype, family)
    assert plan is not None, "the training forward did not build a backward plan"
    assert len(report) >= 4
    for name, worst, tol in report:
        assert worst <= tol, "%s %s: worst |err| / (max(1, rms) + |want|) = %.3e > %.0e" % (
            workload, name, worst, tol)


def test_bench_check_gate_catches_a_wrong_tensor():
    """bench.py's parity gate (``--check``) must refuse a step whose tensors are off."""
    inp = bench.make_inputs("C3pp", torch.float32, "cuda", family="model", seed=0)
    step = bench.make_step(inp)
    assert bench.parity_gate(inp, step) is None

    def broken():
        out, grads = step()
        grads[0][0, 5, 3, 7] += 0.5
        return out, grads
    assert bench.parity_gate(inp, broken) is not None


# ------------------------------------------------------------------ ADVICE.md round 1
def test_padded_value_tail_is_zero_filled():
    """S larger than the packed levels (value padded at the end): grad_value rows past the last
    level are zeros like the reference's at::zeros (box_attn.cu:105), whichever backward runs."""
    from boxer_amd import ops
    levels = [(20, 30), (10, 15), (5, 8), (3, 4)]
    shapes = torch.tensor(levels, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S0 = int(shapes.prod(1).sum())
    B, H, C, L, P, Lq, pad = 2, 8, 32, 4, 4, 500, 37
    g = torch.Generator(device="cuda").manual_seed(3)
    for dtype in (torch.float32, torch.bfloat16):
        value = torch.randn(B, S0 + pad, H, C, device="cuda", generator=g).to(dtype)
        loc = torch.rand(B, Lq, H, L, P, 2, device="cuda", generator=g)
        attn = torch.softmax(torch.randn(B, Lq, H, L * P, device="cuda", generator=g), -1).view(
            B, Lq, H, L, P)
        gout = torch.randn(B, Lq, H * C, device="cuda", generator=g).to(dtype)
        junk = torch.full((B, S0 + pad, H, C), float("nan"), device="cuda", dtype=dtype)
        del junk                                         # the next empty_like likely reuses it
        out, plan = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
        gv, gl, ga = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan)
        torch.cuda.synchronize()
        assert torch.equal(gv[:, S0:], torch.zeros_like(gv[:, S0:]))
        f64 = lambda t: t.detach().double().cpu().numpy()
        want = oc.box_attn_backward(f64(value), shapes.cpu().numpy(), lsi.cpu().numpy(), f64(loc),
                                    f64(attn), f64(gout))
        check(gv, want[0], TOL[dtype], "grad_value (padded S)")


def test_plan_with_ineligible_backward_falls_back():
    """A plan built by the training forward plus a backward operand the binned path rejects (a
    contiguous grad_out view at an unaligned storage offset): the backward ignores the plan and
    runs the atomic path instead of failing (launch_bwd_ws)."""
    from boxer_amd import ops
    levels = [(20, 30), (10, 15)]
    shapes = torch.tensor(levels, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    B, H, C, L, P, Lq = 1, 8, 32, 2, 4, 200
    g = torch.Generator(device="cuda").manual_seed(5)
    value = torch.randn(B, S, H, C, device="cuda", generator=g)
    loc = torch.rand(B, Lq, H, L, P, 2, device="cuda", generator=g)
    attn = torch.softmax(torch.randn(B, Lq, H, L * P, device="cuda", generator=g), -1).view(
        B, Lq, H, L, P)
    buf = torch.randn(B * Lq * H * C + 1, device="cuda", generator=g)
    gout = buf[1:].view(B, Lq, H * C)                    # contiguous, 4-byte aligned only
    assert gout.is_contiguous() and gout.data_ptr() % 16 != 0
    out, plan = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
    assert plan is not None
    gv, gl, ga = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan)
    torch.cuda.synchronize()
    f64 = lambda t: t.detach().double().cpu().numpy()
    want = oc.box_attn_backward(f64(value), shapes.cpu().numpy(), lsi.cpu().numpy(), f64(loc),
                                f64(attn), f64(gout))
    check(gv, want[0], 1e-4, "grad_value")
    check(ga, want[2], 1e-4, "grad_attn")


def test_nonfinite_upstream_row_stays_local_in_the_mfma_accumulate():
    """An Inf in grad_out[b, 0, h, :] (the row the idle lanes of a partially filled MFMA round
    used to fetch) must not leak into blocks query 0 does not touch (idle lanes now
    stage a zero row)."""
    from boxer_amd import ops
    levels = [(40, 60), (20, 30)]
    shapes = torch.tensor(levels, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    B, H, C, L, P, Lq = 1, 8, 32, 2, 4, 900
    g = torch.Generator(device="cuda").manual_seed(7)
    value = torch.randn(B, S, H, C, device="cuda", generator=g).bfloat16()
    loc = torch.rand(B, Lq, H, L, P, 2, device="cuda", generator=g)
    loc[:, 0] = 0.02                                     # query 0 samples the top-left corner only
    attn = torch.softmax(torch.randn(B, Lq, H, L * P, device="cuda", generator=g), -1).view(
        B, Lq, H, L, P)
    gout = torch.randn(B, Lq, H * C, device="cuda", generator=g).bfloat16()
    gout[:, 0] = float("inf")
    out, plan = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
    gv, _, _ = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan)
    torch.cuda.synchronize()
    gv = gv.float().view(B, S, H * C)
    bad = ~torch.isfinite(gv).all(-1)[0]                 # rows with a non-finite element
    rows0 = bad[:40 * 60].view(40, 60)
    rows1 = bad[40 * 60:].view(20, 30)
    # query 0's footprint lies inside the first 8x4 block of each level; nothing else may be hit
    assert not rows0[4:].any() and not rows0[:, 8:].any()
    assert not rows1[4:].any() and not rows1[:, 8:].any()
