#!/usr/bin/env python3
"""Benchmark of the hot path: box-attention fwd+bwd at BoxeR-R50 COCO shapes.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one box-attention forward + backward through the reference's own API -- autograd Functions in the
reference's shape (box_attention_func.py:10-64) on the compiled drop-in module's four functions (vision.cpp:7-12;
``--entry reference``, the default) -- over one per-GPU batch of synthetic input resident in HBM.  Every output element
is defined by the call (no zero-fill or conversion pass outside it).  The timed steps CYCLE 8 input / upstream-gradient
sets generated before the timed region (cache-cold, the headline ``value``); the same step replaying one set is
reported beside it as ``resident``.

Metric (BASELINE.json): Gsample-points/s, one sample point = one (b, query, head, level,
point) bilinear sample of C=32 channels; NP = B*Lq*H*L*P per step and GPU.
Default workload = BASELINE.json configs[1] ("C2" in SURVEY.md 8(d)): 4 levels
(100/50/25/13)^2, d=256, 8 heads, 2x2 grid, Lq = S = 13 294 queries, B = 2 images per GPU,
bf16 storage (value / grad_out / out / grad_value bf16; locations, weights, accumulation fp32).

Multi-GPU: the path shards over images with no data-path collective: every
rank processes its own B images; value = total points of all ranks / max-over-ranks time
("scaling": "weak").  The only collectives are the timing barrier and the MAX reduction.

``python bench.py --gpus N`` with N > 1 and no launcher environment starts its own N ranks
(one fresh ``python`` child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before
the parent touches the GPU; the counterpart of the reference's tools/run.py:59-75); under
``python -m torch.distributed.run`` it uses the ranks it is given.  Backend "nccl" (= RCCL).

Before anything is timed the step's tensors are compared with the CPU oracle (``--no-check``
skips it); a failing comparison prints no JSON line and exits non-zero.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects: "roofline" (dominant kernel vs the
8 TB/s HBM peak; step_frac = the whole step), "resident", "cpu_baseline" (the C restatement of the reference kernels on
the bench workload -- kind "port" -- with the pure-PyTorch leg at BASELINE configs[0] inside) and
"pytorch_fallback_cpu" (north_star's comparator: the pure-PyTorch formulation on the host cores at the shape of this
line, 3 iterations); rank 0, N=1 only.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# name: (levels, Lq ("S" = one query per pixel), points per level P, kind)
WORKLOADS = {
    # BASELINE.json configs[0]: the pure-PyTorch CPU path's shape (1 level 64x64, 100 queries, B = 1)
    "C1": ([(64, 64)], 100, 4, "box"),
    "C2": ([(100, 100), (50, 50), (25, 25), (13, 13)], "S", 4, "box"),
    "C2p": ([(100, 167), (50, 84), (25, 42), (13, 21)], "S", 4, "box"),
    "C3": ([(100, 100), (50, 50), (25, 25), (13, 13)], 300, 16, "instance"),
    "C3p": ([(100, 167), (50, 84), (25, 42), (13, 21)], 300, 196, "instance"),
    "C3pp": ([(100, 167), (50, 84), (25, 42), (13, 21)], 300, 4, "box"),
    "C5": ([(468, 468)], 1000, 4, "box3d"),                 # literal BEV stress shape, rotated
    "C5p": ([(234, 234), (117, 117)], "S", 4, "box3d_fixed"),   # BoxeR-3D encoder, 8 fixed angles
    "C5pp": ([(234, 234), (117, 117)], 300, 4, "box3d"),      # BoxeR-3D decoder, learned rotation
}
H_HEADS, C_HEAD, BATCH = 8, 32, 2
PREHEAT_STEPS = 100         # untimed steps (warm-up included) before the timed region, at least
PREHEAT_SECONDS = 1.0       # ... and at least this long (clock ramp; makes the run visible to samplers)


# --------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------
def encoder_ref_windows(levels, device, ref_size=4.0):
    """Pixel-centre reference windows (cx, cy, ref_size/W_l, ref_size/H_l), one per pixel of
    every level, as BoxeR's encoder builds them (box_transformer.py:70-116, no padding)."""
    out = []
    for (h, w) in levels:
        ys = (torch.arange(1, h + 1, device=device, dtype=torch.float32) - 0.5) / (h + 1e-6)
        xs = (torch.arange(1, w + 1, device=device, dtype=torch.float32) - 0.5) / (w + 1e-6)
        cy, cx = torch.meshgrid(ys, xs, indexing="ij")
        size = torch.tensor([ref_size / w, ref_size / h], device=device).expand(h * w, 2)
        out.append(torch.cat([cx.reshape(-1, 1), cy.reshape(-1, 1), size], dim=1))
    return torch.cat(out, dim=0)


def make_inputs(workload, dtype, device, family="model", batch=BATCH, seed=0):
    """Returns dict(value, shapes, lsi, loc, attn[, level_w], grad_out[, grad_mask], dims).

    family "model": reference windows + N(0,1) box offsets + softmax(N(0,1)) weights run
    through the BoxeR geometry (grid = centre + kernel_index * relu(size)), value ~ N(0,1);
    family "test": i.i.d. uniform locations, normalised uniform weights, value = rand*0.01
    (the reference's test distribution, tests/box_attn_test.py:57-60) -- worst-case locality.
    """
    levels, lq, P, kind = WORKLOADS[workload]
    rot = {"box3d": "learned", "box3d_fixed": "fixed"}.get(kind)         # rotated windows (3D)
    kind = "box" if rot else kind
    L = len(levels)
    S = sum(h * w for h, w in levels)
    Lq = S if lq == "S" else lq
    H, C, B = H_HEADS, C_HEAD, batch
    g = torch.Generator(device=device).manual_seed(seed)
    shapes = torch.tensor(levels, dtype=torch.long, device=device)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    k = int(round(math.sqrt(P)))
    assert k * k == P

    if family == "model":
        value = torch.randn(B, S, H, C, device=device, generator=g)
        if lq == "S":
            ref = encoder_ref_windows(levels, device)[None].expand(B, -1, -1)
        else:
            ctr = 0.05 + 0.9 * torch.rand(B, Lq, 2, device=device, generator=g)
            wh = 0.05 + 0.45 * torch.rand(B, Lq, 2, device=device, generator=g)
            ref = torch.cat([ctr, wh], dim=-1)
        ref = ref[:, :, None, None, :]                                   # (B,Lq,1,1,4)
        off = torch.randn(B, Lq, H, L, 4, device=device, generator=g)
        boxes = ref + off / 8 * ref[..., [2, 3, 2, 3]]
        half = (k - 1) / 2.0
        ticks = torch.linspace(-half, half, k, device=device)
        ky, kx = torch.meshgrid(ticks, ticks, indexing="ij")
        kidx = torch.stack([kx, ky], -1).reshape(-1, 2) / (2 if rot else k)   # (P,2); 3D: /2
        local = kidx * torch.relu(boxes[..., None, 2:])
        if rot:      # Box3dAttention: 8 fixed per-head angles (encoder) or a learned one per box
            ang = (torch.arange(H, device=device, dtype=torch.float32) / H * 2 * math.pi
                   )[None, None, :, None, None].expand(B, Lq, H, L, 1) if rot == "fixed" else \
                torch.rand(B, Lq, H, L, 1, device=device, generator=g) * 2 * math.pi
            lx, ly = local[..., 0], local[..., 1]
            local = torch.stack([lx * ang.cos() - ly * ang.sin(),
                                 lx * ang.sin() + ly * ang.cos()], -1)
        loc = boxes[..., None, :2] + local
        logits = torch.randn(B, Lq, H, L, P, device=device, generator=g)
        attn = torch.softmax(logits.view(B, Lq, H, L * P), -1).view(B, Lq, H, L, P)
        level_w = torch.softmax(logits, dim=3)
    elif family == "test":
        value = torch.rand(B, S, H, C, device=device, generator=g) * 0.01
        loc = torch.rand(B, Lq, H, L, P, 2, device=device, generator=g)
        a = torch.rand(B, Lq, H, L, P, device=device, generator=g) + 1e-5
        attn = a / a.sum((-1, -2), keepdim=True)
        level_w = a / a.sum(-2, keepdim=True)
    else:
        raise ValueError(family)
    grad_out = torch.randn(B, Lq, H * C, device=device, generator=g)
    d = dict(value=value.to(dtype).contiguous(), shapes=shapes, lsi=lsi,
             loc=loc.float().contiguous(), attn=attn.float().contiguous(),
             grad_out=grad_out.to(dtype).contiguous(), kind=kind,
             dims=dict(B=B, S=S, H=H, C=C, L=L, Lq=Lq, P=P))
    if dtype == torch.float64:
        d["loc"], d["attn"] = d["loc"].double(), d["attn"].double()
    if kind == "instance":
        d["level_w"] = level_w.to(d["attn"].dtype).contiguous()
        d["grad_mask"] = torch.randn(B, Lq, P, H * C, device=device, generator=g).to(dtype)
    return d


def n_points(dims):
    return dims["B"] * dims["Lq"] * dims["H"] * dims["L"] * dims["P"]


def algorithmic_bytes(dims, kind, elem):
    """Compulsory HBM bytes: every tensor once, value read capped at what is
    touched, grad_value written as fp32, zero-fills not counted.  Returns (forward, backward)
    totals of the op and the share of each kernel of the binned backward (DESIGN.md section 5:
    the backward is split into a point-gradient kernel and an accumulate kernel, so the
    locations / weights / upstream gradients are read by both)."""
    B, S, H, C, L, Lq, P = (dims[k] for k in ("B", "S", "H", "C", "L", "Lq", "P"))
    NP = B * Lq * H * L * P
    inst = kind == "instance"
    Vr = min(elem * B * S * H * C, elem * 4 * C * NP)
    W = (16 if inst else 12) * NP
    O = elem * B * Lq * H * C
    M = elem * B * Lq * P * H * C if inst else 0
    GV = 4 * B * S * H * C
    fwd = Vr + W + O + M
    bwd = Vr + W + O + M + GV + W
    # the backward's two launches split the backward's compulsory bytes between them (they add up to `bwd`): what the
    # accumulate launch re-reads of O / W -- through the bin records -- is traffic of OUR algorithm, not algorithmic
    per_kernel = {"fwd": fwd, "bwd_points": Vr + W + O + M + W, "bwd_accumulate": GV}
    return fwd, bwd, per_kernel


# --------------------------------------------------------------------------------------
# the timed step
# --------------------------------------------------------------------------------------
def reference_style_functions(mod):
    """The reference's two autograd Functions, written as the reference writes them (box_attention_func.py:10-64,
    83-150: forward saves the tensors and calls ``mod.*_forward``; backward calls ``mod.*_backward`` with nothing
    but the saved tensors) on ``mod`` = a module with the four functions of ``e2edet.ops`` -- the compiled drop-in
    (``boxer_amd._ext.load()``) or ``boxer_amd.ops``.  (No AMP decorators: the bench feeds the storage type directly.)"""
    from torch.autograd import Function
    from torch.autograd.function import once_differentiable

    class RefBoxAttn(Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, attn, im2col_step):
            ctx.im2col_step = im2col_step
            ctx.save_for_backward(value, shapes, lsi, loc, attn)
            return mod.box_attn_forward(value, shapes, lsi, loc, attn, im2col_step)

        @staticmethod
        @once_differentiable
        def backward(ctx, grad_output):
            value, shapes, lsi, loc, attn = ctx.saved_tensors
            gv, gl, ga = mod.box_attn_backward(value, shapes, lsi, loc, attn, grad_output.contiguous(),
                                               ctx.im2col_step)
            return gv, None, None, gl, ga, None

    class RefInstanceAttn(Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, sw, lw, im2col_step):
            ctx.im2col_step = im2col_step
            ctx.save_for_backward(value, shapes, lsi, loc, sw, lw)
            out, mask = mod.instance_attn_forward(value, shapes, lsi, loc, sw, lw, im2col_step)
            return out, mask

        @staticmethod
        @once_differentiable
        def backward(ctx, grad_output, grad_mask):
            value, shapes, lsi, loc, sw, lw = ctx.saved_tensors
            gv, gl, gs, glw = mod.instance_attn_backward(value, shapes, lsi, loc, sw, lw, grad_output.contiguous(),
                                                         grad_mask.contiguous(), ctx.im2col_step)
            return gv, None, None, gl, gs, glw, None
    return RefBoxAttn, RefInstanceAttn


def make_step(inp, entry="ops"):
    """One training step of the operator.  entry "ops": the e2edet.ops boundary with the plan hand-over
    (``*_forward_train`` + ``*_backward(plan=...)``) -- what the autograd Functions call; entry "function":
    the drop-in Functions themselves (``BoxAttnFunction`` / ``BoxAttnBF16Function`` /
    ``InstanceAttn*Function`` ``.apply`` + ``.backward``), i.e. what e2edet/module/box_attention.py:234 runs;
    entry "reference": Functions in the REFERENCE's own shape (reference_style_functions) on the compiled
    drop-in module -- the four functions of e2edet.ops and nothing else."""
    from boxer_amd import ops
    v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn",
                                                  "grad_out"))
    if entry == "reference":
        from boxer_amd import _ext
        torch.autograd.set_multithreading_enabled(False)       # (as for "function", see there)
        ref_box, ref_inst = reference_style_functions(_ext.load())
        vg, lg, ag = (t.detach().clone().requires_grad_() for t in (v, loc, attn))
        if inp["kind"] == "box":
            def step():
                vg.grad = lg.grad = ag.grad = None
                out = ref_box.apply(vg, sh, ls, lg, ag, 64)
                out.backward(go)
                return out, [vg.grad, lg.grad, ag.grad]
        else:
            wg = inp["level_w"].detach().clone().requires_grad_()
            gm = inp["grad_mask"]

            def step():
                vg.grad = lg.grad = ag.grad = wg.grad = None
                out, mask = ref_inst.apply(vg, sh, ls, lg, ag, wg, 64)
                torch.autograd.backward([out, mask], [go, gm.view_as(mask)])
                return (out, mask.view(gm.shape)), [vg.grad, lg.grad, ag.grad, wg.grad]
        return step
    if entry == "function":
        import boxer_amd
        # One tiny autograd graph per step: with the engine's worker threads every step pays a cross-thread
        # hand-over (measured on the GPU box's host: 127 us per step for float32, 220 us for bfloat16, against
        # 100 us on the calling thread -- tools/gpu_cpu_overhead.py), which a real training step pays once per
        # ITERATION, not once per operator.  The bench therefore runs the backward on the calling thread.
        torch.autograd.set_multithreading_enabled(False)
        bf16 = v.dtype == torch.bfloat16
        vg, lg, ag = (t.detach().clone().requires_grad_() for t in (v, loc, attn))
        if inp["kind"] == "box":
            fn = boxer_amd.BoxAttnBF16Function if bf16 else boxer_amd.BoxAttnFunction

            def step():
                vg.grad = lg.grad = ag.grad = None
                out = fn.apply(vg, sh, ls, lg, ag, 64)
                out.backward(go)
                return out, [vg.grad, lg.grad, ag.grad]
        else:
            fn = boxer_amd.InstanceAttnBF16Function if bf16 else boxer_amd.InstanceAttnFunction
            wg = inp["level_w"].detach().clone().requires_grad_()
            gm = inp["grad_mask"]
            ms = int(round(math.sqrt(inp["dims"]["P"])))

            def step():
                vg.grad = lg.grad = ag.grad = wg.grad = None
                out, mask = fn.apply(vg, sh, ls, lg, ag, wg, ms, 64)
                torch.autograd.backward([out, mask], [go, gm.view_as(mask)])
                return (out, mask.view(gm.shape)), [vg.grad, lg.grad, ag.grad, wg.grad]
        return step
    if inp["kind"] == "box":
        def step():      # training step: the forward also prepares the backward's plan
            out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
            grads = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)
            return out, grads
    else:
        lw, gm = inp["level_w"], inp["grad_mask"]

        def step():
            out, plan = ops.instance_attn_forward_train(v, sh, ls, loc, attn, lw, 64)
            grads = ops.instance_attn_backward(v, sh, ls, loc, attn, lw, go, gm, 64, plan=plan)
            return out, grads
    return step


def graph_step(step):
    """The same step as one HIP graph launch (capture after a warm-up on a side stream)."""
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            keep = step()                  # outputs stay alive with the graph
    torch.cuda.current_stream().wait_stream(side)
    graph._keep = keep
    return graph.replay


def time_phases(inp, iters=20):
    """fwd-only and bwd-only device time (ms, median) with events on the op's stream."""
    from boxer_amd import ops
    v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn",
                                                  "grad_out"))
    if inp["kind"] == "box":
        fwd = lambda: ops.box_attn_forward(v, sh, ls, loc, attn, 64)
        bwd = lambda: ops.box_attn_backward(v, sh, ls, loc, attn, go, 64)
    else:
        lw, gm = inp["level_w"], inp["grad_mask"]
        fwd = lambda: ops.instance_attn_forward(v, sh, ls, loc, attn, lw, 64)
        bwd = lambda: ops.instance_attn_backward(v, sh, ls, loc, attn, lw, go, gm, 64)
    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[name] = ts[len(ts) // 2]
    return res


def kernel_profile(step, steps, variant):
    """Average duration of the op's main kernels, measured with HIP events that the library
    records around them on the launch stream (boxattn_profile_*; see include/boxattn.h).

    Same schedule as the timed region: one stream, so a kernel's duration is its own
    (`--variant 6` forks the point-gradient kernel onto the library's helper stream, which
    stretches the overlapped kernels)."""
    from boxer_amd import _lib
    if not hasattr(_lib, "profile_begin"):
        return None
    _lib.set_variant(variant)
    for _ in range(3):
        step()
    _lib.profile_begin()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    res = _lib.profile_end()
    _lib.set_variant(variant)
    return res


# --------------------------------------------------------------------------------------
# parity gate: nothing is timed unless the step's tensors match the CPU oracle
# --------------------------------------------------------------------------------------
PARITY_INFO = {}       # filled by parity_report: how many points the grad_loc comparison leaves out


def parity_report(inp, out, grads):
    """Compare every tensor of one step with the CPU oracle (oracle/boxattn_oracle.c on the same
    -- for bf16: the rounded -- inputs) -> [(name, worst ratio, tol)] with
    ratio = |got - want| / (max(1, rms(want)) + |want|) per element, tol = 1e-4 for fp32 tensors
    and 1e-2 for bf16 ones (BASELINE.json north_star).

    The oracle runs in float64, except for grad_loc: the bilinear fractions come from the pixel
    coordinate ``loc * size - 0.5`` evaluated in float32 by the operator (and by the reference's
    CUDA kernels, box_attn_kernel.cuh:318-319); at coordinates ~100 its rounding moves the
    fractions by ~1e-5, which the float64 evaluation does not see and which enters grad_loc
    multiplied by W_l * a * |difference of corner sums|.  grad_loc is therefore checked against
    the oracle's float32 build (same coordinate rounding, other summation order), and not for
    points within 1e-4 px of a bilinear cell edge (it is discontinuous there)."""
    import numpy as np
    from oracle import boxattn_oracle as oc
    f64 = lambda t: t.detach().double().cpu().numpy()
    a = {k: (f64(v) if v.is_floating_point() else v.cpu().numpy())
         for k, v in inp.items() if isinstance(v, torch.Tensor)}
    a32 = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in a.items()}
    oc.set_num_threads(max(1, min(os.cpu_count() or 1, inp["dims"]["B"] * inp["dims"]["H"])))
    if inp["kind"] == "box":
        args = lambda d: (d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"])
        want = [oc.box_attn_forward(*args(a))] + list(oc.box_attn_backward(*args(a), a["grad_out"]))
        want[2] = oc.box_attn_backward(*args(a32), a32["grad_out"])[1].astype(np.float64)
        got = [out] + list(grads)
        names = ["out", "grad_value", "grad_loc", "grad_attn"]
    else:
        args = lambda d: (d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], d["level_w"])
        want = list(oc.instance_attn_forward(*args(a)))
        want += list(oc.instance_attn_backward(*args(a), a["grad_out"], a["grad_mask"]))
        want[3] = oc.instance_attn_backward(*args(a32), a32["grad_out"],
                                            a32["grad_mask"])[1].astype(np.float64)
        got = list(out) + list(grads)
        names = ["out", "mask_out", "grad_value", "grad_loc", "grad_spatial", "grad_level"]
    size = a["shapes"].astype(np.float64)[None, None, None, :, None, ::-1]       # (W, H)
    pix = a["loc"] * size - 0.5
    edge = (np.abs(pix - np.round(pix)) < 1e-4).any(-1, keepdims=True)
    PARITY_INFO.update(points=int(edge.size), edge_points=int(edge.sum()))     # reported with the bench line
    report = []
    for name, g, w in zip(names, got, want):
        tol = 1e-2 if g.dtype == torch.bfloat16 else 1e-4
        g = f64(g).reshape(w.shape)
        if name == "grad_loc":
            g, w = g * ~edge, w * ~edge
        if not np.isfinite(g).all():
            report.append((name, float("inf"), tol))
            continue
        scale = max(1.0, float(np.sqrt(np.mean(w * w)))) if w.size else 1.0
        ratio = np.abs(g - w) / (scale + np.abs(w))
        report.append((name, float(ratio.max()) if w.size else 0.0, tol))
    return report


def parity_gate(inp, step):
    """Run ``step`` once and check its tensors (parity_report).  -> None when everything
    matches, else a description of the failures."""
    out, grads = step()
    torch.cuda.synchronize()
    bad = ["%s: worst |err| / (max(1, rms) + |want|) = %.3e > %.0e" % r
           for r in parity_report(inp, out, grads) if not r[1] <= r[2]]
    return "; ".join(bad) if bad else None


# --------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1), both bounded
# --------------------------------------------------------------------------------------
def cpu_baseline(workload, budget_s=15.0):
    """Leg 1 (the line's ``cpu_baseline``, kind "port"): the C restatement of the reference
    kernels (oracle/boxattn_oracle.c, OpenMP over image x head, fp32) on the bench workload
    itself, forward + backward, ~budget_s seconds.
    Leg 2 (``pytorch_fallback`` inside it): BASELINE.json's north-star wording -- the repo's
    pure-PyTorch formulation (oracle/torch_fallback.py: per level grid_sample + weighted sum,
    autograd backward; the counterpart of the reference's tests/box_attn_test.py:9-42) on the
    host cores at BASELINE configs[0] (C1), core count stated."""
    from oracle import boxattn_oracle as oc
    levels, lq, P, kind = WORKLOADS[workload]
    kind = "box" if kind.startswith("box3d") else kind
    batch = 1 if workload == "C1" else BATCH
    inp = make_inputs(workload, torch.float32, "cpu", family="model", batch=batch, seed=0)
    a = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    np_ = n_points(inp["dims"])
    pairs = inp["dims"]["B"] * inp["dims"]["H"]               # the restatement's parallel axis
    cores = max(1, min(os.cpu_count() or 1, pairs))
    oc.set_num_threads(cores)

    def once():
        if kind == "box":
            oc.box_attn_forward(a["value"], a["shapes"], a["lsi"], a["loc"], a["attn"])
            oc.box_attn_backward(a["value"], a["shapes"], a["lsi"], a["loc"], a["attn"],
                                 a["grad_out"])
        else:
            oc.instance_attn_forward(a["value"], a["shapes"], a["lsi"], a["loc"], a["attn"],
                                     a["level_w"])
            oc.instance_attn_backward(a["value"], a["shapes"], a["lsi"], a["loc"], a["attn"],
                                      a["level_w"], a["grad_out"], a["grad_mask"])

    t0 = time.perf_counter()
    once()                                            # warm-up (also sizes the loop)
    first = time.perf_counter() - t0
    iters = max(1, min(400, int(budget_s / max(first, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    dt = (time.perf_counter() - t0) / iters
    res = {"value": np_ / dt / 1e9, "unit": "Gsample-points/s", "cores": cores,
           "kind": "port",
           "sample": "%s fp32, B=%d images (%d points), fwd+bwd of the C restatement of the "
                     "reference kernels (oracle/boxattn_oracle.c), %d iterations, %.3f s/iter, "
                     "%d OpenMP threads (one per image x head)" % (workload, batch, np_, iters,
                                                                   dt, cores)}
    res["pytorch_fallback"] = pytorch_fallback("C1")
    return res


def pytorch_fallback(workload="C1", budget_s=8.0, iters=None):
    """The north star's comparator: box attention as pure PyTorch on the CPU (grid_sample formulation, fwd + autograd
    bwd; the reference's own oracle, tests/box_attn_test.py:9-42).  "C1" = BASELINE.json configs[0] (N=1, 1 level 64x64,
    100 queries, 8 heads, 2x2 grid: bounded by `budget_s`); the headline shape (C2, B = 2) runs `iters` = 3 iterations on
    all host cores."""
    from oracle import torch_fallback as tf
    batch = 1 if workload == "C1" else BATCH
    inp = make_inputs(workload, torch.float32, "cpu", family="model", batch=batch, seed=0)
    # (C1 is tiny: beyond ~16 threads the intra-op fork/join dominates -- 256 threads on the GPU box's host: 870 ms
    # per iteration instead of a few ms; the headline shape takes every core)
    cores = min(os.cpu_count() or 1, 16) if workload == "C1" else (os.cpu_count() or 1)
    old_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    v = inp["value"].clone().requires_grad_()
    loc = inp["loc"].clone().requires_grad_()
    attn = inp["attn"].clone().requires_grad_()
    shapes = [tuple(int(x) for x in r) for r in inp["shapes"]]

    def once():
        out = tf.box_attn(v, shapes, loc, attn)
        v.grad = loc.grad = attn.grad = None
        out.backward(inp["grad_out"])

    once()
    if iters is None:
        t0 = time.perf_counter()
        once()
        first = time.perf_counter() - t0
        iters = max(3, min(2000, int(budget_s / max(first, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    dt = (time.perf_counter() - t0) / iters
    torch.set_num_threads(old_threads)
    np_ = n_points(inp["dims"])
    return {"value": np_ / dt / 1e9, "unit": "Gsample-points/s", "cores": cores,
            "kind": "pure-PyTorch grid_sample formulation (oracle/torch_fallback.py)",
            "sample": "%s fp32: B=%d, levels %s, %d queries, 8 heads, C=32, %d points a level (%d sample "
                      "points), fwd + autograd bwd, %d iterations, %.3f ms/iter, torch threads=%d"
                      % (workload, batch, "/".join("%dx%d" % hw for hw in WORKLOADS[workload][0]), inp["dims"]["Lq"],
                         inp["dims"]["P"], np_, iters, dt * 1e3, cores)}


# --------------------------------------------------------------------------------------
# timing protocol (shared with tests/test_dist_gloo.py, which runs it on CPU with gloo)
# --------------------------------------------------------------------------------------
def run_timed(step, steps, warmup, sync, dist=None, device=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on
    both sides; returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def throughput(elapsed, points_per_rank_step, world, steps):
    """Whole-job Gpoints/s (every rank processes its own images: weak scaling) and ms/step."""
    total = points_per_rank_step * world * steps
    return total / elapsed / 1e9, elapsed / steps * 1e3



# --------------------------------------------------------------------------------------
# self-launch: python bench.py --gpus N without a launcher
# --------------------------------------------------------------------------------------
def spawn_ranks(n, cmd=None):
    """Start n fresh ``python bench.py`` children (one per GPU) with the launcher environment
    set, BEFORE this process has touched the GPU (no re-exec of a GPU process: plain children);
    rank 0's stdout is ours.  Returns the exit code.  (``cmd``: another command line, for the
    CPU test of the launcher.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def resident_leg(args, step, np_rank, device):
    """The same step replaying ONE input set (the parity-gated one): inputs, outputs and bin records of a C2 step are
    ~250 MB, about the size of the 256 MiB Infinity Cache, so part of its traffic is served on-die -- reported beside the
    headline, which cycles N sets (inputs regenerated outside the timed region)."""
    n_timed = max(1, min(args.steps, 1000))
    elapsed = run_timed(step, n_timed, min(max(args.warmup, 8), 50), torch.cuda.synchronize, None, device)
    value, ms = throughput(elapsed, np_rank, 1, n_timed)
    return {"steps": n_timed, "ms_per_step": round(ms, 4), "value": round(value, 4), "unit": "Gsample-points/s",
            "note": "one input set replayed (cache-resident): this rank alone, no barrier"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)        # ~0.5-1 s timed at the default workload
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--inputs", default="model", choices=["model", "test"])
    ap.add_argument("--batch", type=int, default=None,
                    help="images per GPU (the headline line uses the default: 2; C1: 1)")
    ap.add_argument("--entry", default="reference", choices=["ops", "function", "reference"],
                    help="what a step calls: Functions in the reference's own shape on the compiled drop-in module -- "
                         "its four functions only (default: the API the reference's model code calls), the e2edet.ops "
                         "boundary of boxer_amd.ops with the plan hand-over, or boxer_amd's autograd Functions "
                         "(.apply + .backward)")
    ap.add_argument("--rotate", type=int, default=8, metavar="N",
                    help="the timed steps cycle N input / upstream-gradient sets generated outside the timed region "
                         "(cache-cold: the headline); the one-set replay is reported beside it as `resident` "
                         "(0 / 1: the headline replays one set)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step in a HIP graph and time replays (not the headline run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the parity gate")
    ap.add_argument("--preheat-s", type=float, default=PREHEAT_SECONDS,
                    help="untimed pre-heat before the protocol, seconds (profiling runs pass 0)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant override (A/B)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="library tuning option (boxattn_set_option), e.g. 11=2: dense encoder kernels on")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 1 if args.workload == "C1" else BATCH

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus"
    device = torch.device("cuda", torch.cuda.current_device())

    from boxer_amd import _lib
    _lib.set_variant(args.variant)
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.load().boxattn_set_option(int(k), int(v))
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    # every rank owns its own images (different seed): data-parallel shard, no exchange
    inp = make_inputs(args.workload, dtype, device, family=args.inputs, batch=args.batch,
                      seed=rank)
    step0 = make_step(inp, args.entry)

    # parity gate on the tensors of the step that is about to be timed (every rank its own)
    if not args.no_check:
        failure = parity_gate(inp, step0)
        if failure is not None:
            print("bench.py: parity gate FAILED on rank %d (%s %s %s): %s" % (
                rank, args.workload, args.dtype, args.inputs, failure), file=sys.stderr, flush=True)
            sys.exit(3)
    # The HEADLINE cycles N input / upstream-gradient sets (generated here, outside the timed region: 544 MB at C2
    # bf16): consecutive steps share nothing but the level tables, as consecutive layers / iterations of a training
    # run do.  (One replayed set is cache-resident; it is timed too and reported as `resident`.)
    n_sets = args.rotate if args.rotate > 1 else 1
    sets = [inp] + [make_inputs(args.workload, dtype, device, family=args.inputs, batch=args.batch,
                                seed=1000 * (rank + 1) + i) for i in range(1, n_sets)]
    eager_fns = [step0] + [make_step(x, args.entry) for x in sets[1:]]
    fns = [graph_step(f) for f in eager_fns] if args.graph else eager_fns
    turn = [0]

    def cycle(fs):
        def step():
            turn[0] += 1
            return fs[turn[0] % len(fs)]()
        return step
    step, eager_step = cycle(fns), cycle(eager_fns)

    # Device pre-heat: the first ~100 steps after start-up run up to ~8 % slower than the steady state
    # (clock ramp, tools/gpu_ramp.py).  With a short --warmup the gap is filled here, outside
    # the W warm-up + K timed steps of the protocol, so that the number is the steady-state one.
    preheat, t0 = 0, time.perf_counter()
    while preheat < max(0, PREHEAT_STEPS - args.warmup) or \
            time.perf_counter() - t0 < args.preheat_s:
        step()
        preheat += 1
        if preheat % 64 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    elapsed = run_timed(step, args.steps, args.warmup, torch.cuda.synchronize, dist, device)
    step = eager_step                      # the per-kernel profile needs the launches themselves

    np_rank = n_points(inp["dims"])
    value, ms_per_step = throughput(elapsed, np_rank, world, args.steps)
    resident = resident_leg(args, fns[0], np_rank, device) if n_sets > 1 else None
    per_rank = None
    if dist is not None:                   # per-rank spread (clock / power variance between GPUs)
        t0 = time.perf_counter()           # every rank alone, no barrier: its own rate
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        mine = torch.tensor([np_rank * args.steps / (time.perf_counter() - t0) / 1e9],
                            device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        vals = sorted(float(t.item()) for t in allr)
        per_rank = {"min": round(vals[0], 3), "median": round(vals[len(vals) // 2], 3),
                    "max": round(vals[-1], 3), "unit": "Gsample-points/s per GPU (each rank re-timed alone, no barrier)"}

    phases = time_phases(inp)
    prof = kernel_profile(step, min(args.steps, 20), args.variant)
    elem = 2 if dtype == torch.bfloat16 else 4
    b_fwd, b_bwd, b_kernel = algorithmic_bytes(inp["dims"], inp["kind"], elem)

    if rank == 0:
        # dominant kernel = the longest of the op's kernels (HIP events recorded by the library
        # around each launch, on the launch stream)
        kern = {k: v for k, v in (prof or {}).items() if v["ms"]}
        if kern and any(k in b_kernel for k in kern):
            dom = max((k for k in kern if k in b_kernel), key=lambda k: kern[k]["ms"])
            dom_ms, dom_bytes = kern[dom]["ms"], b_kernel[dom]
            src = ("HIP events around every launch of the kernel (boxattn_profile_*), same schedule as the "
                   "timed region; the bracketed pass runs ~10 % slower than the free-running step, so the "
                   "per-kernel averages add up to more than ms_per_step")
        else:
            dom, dom_ms, dom_bytes = "bwd (whole call)", phases["bwd"], b_bwd
            src = "HIP events around the backward call"
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        step_gbs = (b_fwd + b_bwd) / (ms_per_step * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as fh:
                t = json.load(fh)
            key = "%s/%s/%s" % (args.workload, args.dtype, args.inputs)
            traffic = t.get(key, {}).get(dom)
            if traffic is not None:
                traffic_src = ("profiles/hbm_traffic.json (builder's rocprofv3 --pmc run of this "
                               "command, %s; not measured in this run)" % t.get("_round", "r02"))
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": dom_bytes,
                    "avg_launch_ms": round(dom_ms, 4), "timing": src,
                    # the whole step against the roofline: north_star's ">= 60 %" is THIS number
                    "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
                    "fwd_bwd": {"algorithmic_bytes": b_fwd + b_bwd,
                                "achieved_GBs": round(step_gbs, 1),
                                "frac": round(step_gbs / HBM_PEAK_GBS, 4)},
                    "fwd_ms": round(phases["fwd"], 4), "bwd_ms": round(phases["bwd"], 4),
                    "kernels": {k: {"avg_ms": round(v["ms"], 4), "launches": v["launches"],
                                    "algorithmic_bytes": b_kernel.get(k)}
                                for k, v in kern.items()}}
        line = {
            "metric": "box-attn fwd+bwd Gsample-points/s + achieved HBM GB/s, BoxeR-R50 COCO shapes",
            "value": round(value, 4), "unit": "Gsample-points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "launch": "hip-graph replay" if args.graph else "eager",
            "entry": {"ops": "boxer_amd.ops.*_forward_train + *_backward(plan=...) (the e2edet.ops boundary as "
                             "boxer_amd's Functions call it)",
                      "function": "autograd Function .apply + .backward (the reference's call site, "
                                  "box_attention.py:234)",
                      "reference": "Functions in the reference's own shape (box_attention_func.py:10-64) on the "
                                   "compiled drop-in module: box_attn_forward / box_attn_backward only"}[args.entry],
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: %s-attn fwd+bwd, levels %s, Lq=%d, H=%d, C=%d, P=%d, "
                                   "B=%d images per GPU, inputs=%s" % (
                                       args.workload, inp["kind"],
                                       "/".join("%dx%d" % hw for hw in WORKLOADS[args.workload][0]),
                                       inp["dims"]["Lq"], H_HEADS, C_HEAD, inp["dims"]["P"], args.batch,
                                       args.inputs),
                       "points_per_step_per_gpu": np_rank, "parallelism": "dp%d" % world,
                       "preheat_steps": preheat,
                       "parity_gate": "skipped" if args.no_check else
                       "passed (all tensors vs CPU oracle; grad_loc: %d of %d points within 1e-4 px of a bilinear "
                       "cell edge not compared)" % (PARITY_INFO.get("edge_points", -1), PARITY_INFO.get("points", -1))},
            "roofline": roofline,
        }
        line["inputs"] = ("%d input / upstream-gradient sets (%.0f MB) cycled from step to step: cache-cold" % (
            n_sets, sum(t.numel() * t.element_size() for x in sets for t in x.values()
                        if isinstance(t, torch.Tensor)) / 1e6)) if n_sets > 1 else "one input set replayed (cache-resident)"
        if resident is not None:
            line["resident"] = resident
        if per_rank is not None:
            line["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload)
            if inp["kind"] == "box" and not WORKLOADS[args.workload][3].startswith("box3d"):
                # north_star's comparator at the shape of this line: the pure-PyTorch fallback on the host cores,
                # 3 iterations -- a top-level key
                line["pytorch_fallback_cpu"] = pytorch_fallback(args.workload, iters=3)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
