#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run only in the build container (needs /root/reference; the GPU box has no reference):

    python tests/golden/make_goldens.py

What is executed from the reference (nothing of it is copied into this repo -- the
functions are pulled out of the reference files at run time and exec'd in memory):

* ``PlainBoxAttnFunction``       tests/box_attn_test.py:9-42
* ``PlainInstanceAttnFunction``  tests/instance_attn_test.py:11-63
* ``view_with_shape``            e2edet/utils/general.py:289-324
* ``e2edet/module/box_attention.py`` (BoxAttention / InstanceAttention / Box3dAttention),
  loaded with stub ``e2edet.module.ops`` Functions that forward to the two oracles above.

These are the reference's own test oracles for its CUDA op (the op has no CPU build, and
the package itself cannot be imported here: torchvision / torch._six are missing), run in
fp64 on CPU.  Each fixture is an .npz of inputs, expected outputs and expected gradients
for stored upstream gradients.  Fixture ids follow SURVEY.md section 8(c): G1..G7; G8 (layers) and G9
(modules at the model's head geometry, with gradients) were added in rounds 2 and 5.

Storage notes: all expected outputs are float64.  Inputs are float64 except where a
``*_q8`` int8 array + ``*_scale`` is stored (value = q8 * scale, exactly representable in
bf16/fp32/fp64 -- keeps fixtures small and makes bf16 input rounding a non-issue).
"""
import ast
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)


def _extract(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing


NS = {"torch": torch, "F": F, "math": math}
_extract(os.path.join(REF, "e2edet/utils/general.py"), ["view_with_shape"], NS)
_extract(os.path.join(REF, "tests/box_attn_test.py"), ["PlainBoxAttnFunction"], NS)
_extract(os.path.join(REF, "tests/instance_attn_test.py"), ["PlainInstanceAttnFunction"], NS)
ref_box = NS["PlainBoxAttnFunction"]
ref_inst = NS["PlainInstanceAttnFunction"]


def lsi_of(shapes):
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    return torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))


def run_box(value, shapes, loc, attn, gout):
    """value (B,S,H,C) etc. in fp64 -> dict of expected out + grads (reference oracle)."""
    B, S, H, C = value.shape
    v = value.clone().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    out = ref_box(v.view(B, S, H * C), shapes, 2 * l - 1, a)
    out.backward(gout)
    return dict(out=out.detach(), grad_value=v.grad, grad_loc=l.grad, grad_attn=a.grad)


def run_inst(value, shapes, loc, sw, lw, ms, gout, gmask):
    B, S, H, C = value.shape
    v = value.clone().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    s = sw.clone().requires_grad_(True)
    w = lw.clone().requires_grad_(True)
    out, mask = ref_inst(v.view(B, S, H * C), shapes, 2 * l - 1, s, w, ms)
    # mask: (B, Lq, ms, ms, H*C)
    (out * gout).sum().add((mask * gmask).sum()).backward()
    return dict(out=out.detach(), mask_out=mask.detach(), grad_value=v.grad, grad_loc=l.grad,
                grad_spatial=s.grad, grad_level=w.grad)


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def q8_value(gen, shape, scale=1.0 / 64):
    q = torch.randint(-127, 128, shape, generator=gen, dtype=torch.int64)
    return q.to(torch.int8), scale, q.double() * scale


def box_weights(gen, B, Lq, H, L, P):
    a = torch.rand(B, Lq, H, L, P, generator=gen) + 1e-5
    return a / a.sum(-1, keepdim=True).sum(-2, keepdim=True)


def inst_weights(gen, B, Lq, H, L, ms):
    a = torch.rand(B, Lq, H, L, ms, ms, generator=gen) + 1e-5
    sw = a / a.sum(-1, keepdim=True).sum(-2, keepdim=True).sum(-3, keepdim=True)
    lw = a / a.sum(-3, keepdim=True)
    return sw, lw


# --------------------------------------------------------------------------------------
# G1: the reference's own box test shape (tests/box_attn_test.py:45-60), seed 3
# --------------------------------------------------------------------------------------
def g1():
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = torch.ones(N, Lq, M * D)          # the reference test uses output.sum().backward()
    exp = run_box(value, shapes, loc, attn, gout)
    save("G1_box_reftest", value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc, attn=attn,
         grad_out=gout, **exp)


# --------------------------------------------------------------------------------------
# G2: the reference's own instance test shape (tests/instance_attn_test.py:66-90), seed 3
# --------------------------------------------------------------------------------------
def g2():
    torch.manual_seed(3)
    N, M, D, Lq, L, P, MS = 1, 2, 2, 2, 2, 4, 2
    shapes = torch.tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    a = torch.rand(N, Lq, M, L, MS, MS) + 1e-5
    sw = a / a.sum(-1, keepdim=True).sum(-2, keepdim=True).sum(-3, keepdim=True)
    lw = a / a.sum(-3, keepdim=True)
    gen = torch.Generator().manual_seed(32)
    gout = torch.randn(N, Lq, M * D, generator=gen)
    gmask = torch.randn(N, Lq, MS, MS, M * D, generator=gen)
    exp = run_inst(value, shapes, loc, sw, lw, MS, gout, gmask)
    save("G2_inst_reftest", value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc,
         spatial_w=sw, level_w=lw, mask_size=MS, grad_out=gout, grad_mask=gmask, **exp)


# --------------------------------------------------------------------------------------
# G3: channel sweep of the reference gradcheck (tests/box_attn_test.py:194): every
#     backward-kernel family of the reference (30,32,64,71; 1025 kept tiny with Lq=1)
# --------------------------------------------------------------------------------------
def g3():
    shapes = torch.tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    for D in (30, 32, 64, 71, 1025):
        gen = torch.Generator().manual_seed(300 + D)
        N, M, L, P = 1, 2, 2, 2
        Lq = 1 if D > 1000 else 2
        value = torch.rand(N, S, M, D, generator=gen) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2, generator=gen)
        attn = box_weights(gen, N, Lq, M, L, P)
        gout = torch.randn(N, Lq, M * D, generator=gen)
        exp = run_box(value, shapes, loc, attn, gout)
        save("G3_box_C%d" % D, value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc,
             attn=attn, grad_out=gout, **exp)
        # instance flavour of the same sweep (instance_attn_test.py has the same loop)
        if D in (30, 32, 71):
            sw, lw = inst_weights(gen, N, Lq, M, L, 2)
            loc4 = torch.rand(N, Lq, M, L, 4, 2, generator=gen)
            gm = torch.randn(N, Lq, 2, 2, M * D, generator=gen)
            exp = run_inst(value, shapes, loc4, sw, lw, 2, gout, gm)
            save("G3_inst_C%d" % D, value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc4,
                 spatial_w=sw, level_w=lw, mask_size=2, grad_out=gout, grad_mask=gm, **exp)


# --------------------------------------------------------------------------------------
# G4: out-of-range and edge locations: U(-0.2, 1.2) plus exact {0, 1, 0.5/W, 1-0.5/W, ...}
#     on power-of-two maps (both "x*W-0.5" and grid_sample's "((2x-1+1)*W-1)/2" are exact).
#     NOT included: a pixel coordinate of exactly -1 (loc = -0.5/W).  There the reference's
#     CUDA kernel and its own grid_sample test oracle disagree on grad_loc: the kernel's
#     window test `h_im > -1 && w_im > -1` (box_attn_kernel.cuh:325-328) skips the point
#     (all grads 0) while grid_sample differentiates the zero-weight valid corner.  The
#     kernel is the behaviour we follow; tests/test_oracle_golden.py pins it separately.
# --------------------------------------------------------------------------------------
def g4():
    gen = torch.Generator().manual_seed(4)
    shapes = torch.tensor([(8, 4), (4, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    B, H, C, Lq, L, P = 2, 3, 5, 7, 2, 4
    value = torch.randn(B, S, H, C, generator=gen)
    loc = torch.rand(B, Lq, H, L, P, 2, generator=gen) * 1.4 - 0.2
    # exact edge values, per level so the pixel arithmetic is exact
    for l, (hl, wl) in enumerate(shapes.tolist()):
        xs = torch.tensor([0.0, 1.0, 0.5 / wl, 1 - 0.5 / wl, 1.5 / wl, -0.25 / wl, 1 + 0.5 / wl])
        ys = torch.tensor([0.0, 1.0, 0.5 / hl, 1 - 0.5 / hl, 1.5 / hl, -0.25 / hl, 1 + 0.5 / hl])
        for q in range(Lq):
            loc[0, q, 0, l, 0, 0] = xs[q % len(xs)]
            loc[0, q, 0, l, 1, 1] = ys[q % len(ys)]
            loc[0, q, 1, l, 2, 0] = xs[(q + 3) % len(xs)]
            loc[0, q, 1, l, 2, 1] = ys[q % len(ys)]
    attn = box_weights(gen, B, Lq, H, L, P)
    gout = torch.randn(B, Lq, H * C, generator=gen)
    exp = run_box(value, shapes, loc, attn, gout)
    save("G4_box_edges", value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc, attn=attn,
         grad_out=gout, **exp)
    sw, lw = inst_weights(gen, B, Lq, H, L, 2)
    gm = torch.randn(B, Lq, 2, 2, H * C, generator=gen)
    exp = run_inst(value, shapes, loc, sw, lw, 2, gout, gm)
    save("G4_inst_edges", value=value, shapes=shapes, lsi=lsi_of(shapes), loc=loc, spatial_w=sw,
         level_w=lw, mask_size=2, grad_out=gout, grad_mask=gm, **exp)


# --------------------------------------------------------------------------------------
# G5: BASELINE.json configs[0] (C1): 1 level 64x64, 100 queries, 8 heads, C=32, 2x2 grid
# --------------------------------------------------------------------------------------
def g5():
    gen = torch.Generator().manual_seed(5)
    shapes = torch.tensor([(64, 64)], dtype=torch.long)
    B, S, H, C, Lq, L, P = 1, 64 * 64, 8, 32, 100, 1, 4
    q8, scale, value = q8_value(gen, (B, S, H, C))
    loc = (torch.rand(B, Lq, H, L, P, 2, generator=gen) * 1.1 - 0.05).float().double()
    attn = box_weights(gen, B, Lq, H, L, P).float().double()
    gout = (torch.randint(-64, 65, (B, Lq, H * C), generator=gen).double() / 32)
    exp = run_box(value, shapes, loc, attn, gout)
    gv = exp.pop("grad_value")
    # grad_value is 8 MB dense in fp64 but only ~4*P*Lq*H rows are non-zero: store sparsely
    rows = gv.view(B * S * H, C).abs().sum(-1).nonzero().squeeze(1)
    save("G5_box_C1", value_q8=q8, value_scale=scale, shapes=shapes, lsi=lsi_of(shapes), loc=loc,
         attn=attn, grad_out=gout, grad_value_rows=rows, grad_value_vals=gv.view(-1, C)[rows],
         **exp)


# --------------------------------------------------------------------------------------
# G6: small multi-level maps with the BoxeR head geometry (H=8, C=32)
# --------------------------------------------------------------------------------------
def g6():
    shapes = torch.tensor([(13, 17), (7, 9), (4, 5), (2, 3)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    gen = torch.Generator().manual_seed(6)
    B, H, C, L = 2, 8, 32, 4
    q8, scale, value = q8_value(gen, (B, S, H, C))

    Lq, P = 64, 4                                            # encoder-like box attention
    loc = (torch.rand(B, Lq, H, L, P, 2, generator=gen) * 1.2 - 0.1).float().double()
    attn = box_weights(gen, B, Lq, H, L, P).float().double()
    gout = torch.randint(-64, 65, (B, Lq, H * C), generator=gen).double() / 32
    exp = run_box(value, shapes, loc, attn, gout)
    save("G6_box_ml", value_q8=q8, value_scale=scale, shapes=shapes, lsi=lsi_of(shapes), loc=loc,
         attn=attn, grad_out=gout, **exp)

    Lq, ms = 10, 4                                           # instance attention, 4x4 grid
    loc = (torch.rand(B, Lq, H, L, ms * ms, 2, generator=gen) * 1.2 - 0.1).float().double()
    sw, lw = inst_weights(gen, B, Lq, H, L, ms)
    sw, lw = sw.float().double(), lw.float().double()
    gout = torch.randint(-64, 65, (B, Lq, H * C), generator=gen).double() / 32
    gm = torch.randint(-64, 65, (B, Lq, ms, ms, H * C), generator=gen).double() / 32
    exp = run_inst(value, shapes, loc, sw, lw, ms, gout, gm)
    save("G6_inst_ms4", value_q8=q8, value_scale=scale, shapes=shapes, lsi=lsi_of(shapes),
         loc=loc, spatial_w=sw, level_w=lw, mask_size=ms, grad_out=gout, grad_mask=gm, **exp)

    Lq, ms, H2 = 3, 14, 2                                    # reference-actual 14x14 grid
    q8b, scale, value2 = q8_value(gen, (1, S, H2, C))
    loc = (torch.rand(1, Lq, H2, L, ms * ms, 2, generator=gen) * 1.2 - 0.1).float().double()
    sw, lw = inst_weights(gen, 1, Lq, H2, L, ms)
    sw, lw = sw.float().double(), lw.float().double()
    gout = torch.randint(-64, 65, (1, Lq, H2 * C), generator=gen).double() / 32
    gm = torch.randint(-64, 65, (1, Lq, ms, ms, H2 * C), generator=gen).double() / 32
    exp = run_inst(value2, shapes, loc, sw, lw, ms, gout, gm)
    save("G6_inst_ms14", value_q8=q8b, value_scale=scale, shapes=shapes, lsi=lsi_of(shapes),
         loc=loc, spatial_w=sw, level_w=lw, mask_size=ms, grad_out=gout, grad_mask=gm, **exp)


# --------------------------------------------------------------------------------------
# G7: module level -- the reference nn.Modules (geometry code + projections) run on CPU
#     with the reference oracles standing in for the CUDA Functions
# --------------------------------------------------------------------------------------
def load_reference_modules():
    class _Box:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, im2col_step):
            B, S, H, C = value.shape
            b, l1, nh, nl = attn.shape[:4]
            return ref_box(value.reshape(B, S, H * C), shapes, 2 * loc - 1,
                           attn.reshape(b, l1, nh, nl, -1))

    class _Inst:
        @staticmethod
        def apply(value, shapes, lsi, loc, sw, lw, mask_size, im2col_step):
            B, S, H, C = value.shape
            return ref_inst(value.reshape(B, S, H * C), shapes, 2 * loc - 1, sw, lw, mask_size)

    for name in ("e2edet", "e2edet.module", "e2edet.module.ops"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["e2edet.module.ops"].BoxAttnFunction = _Box
    sys.modules["e2edet.module.ops"].InstanceAttnFunction = _Inst
    spec = importlib.util.spec_from_file_location(
        "_ref_box_attention", os.path.join(REF, "e2edet/module/box_attention.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def g7():
    ref = load_reference_modules()
    shapes = torch.tensor([(6, 5), (3, 4)], dtype=torch.long)
    lsi = lsi_of(shapes)
    S = int(shapes.prod(1).sum())
    d, nl, nh, B, Lq = 32, 2, 4, 2, 5

    def randomise(m, seed):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def common(seed, ref_dim):
        g = torch.Generator().manual_seed(seed)
        query = torch.randn(B, Lq, d, generator=g)
        value = torch.randn(B, S, d, generator=g)
        v_mask = torch.rand(B, S, generator=g) < 0.1
        ratios = 0.7 + 0.3 * torch.rand(B, 1, 1, nl, 1, 2, generator=g)
        rw = torch.rand(B, Lq, ref_dim, generator=g)
        rw[..., 2:4] = 0.1 + 0.4 * rw[..., 2:4]
        return query, value, v_mask, ratios, rw

    def pack(m, extra):
        sd = {"sd." + k: v for k, v in m.state_dict().items()}
        sd.update(extra)
        return sd

    # BoxAttention
    m = ref.BoxAttention(d, nl, nh, kernel_size=2).double()
    randomise(m, 70)
    query, value, v_mask, ratios, rw = common(71, 4)
    grid = m._where_to_attend(query, ratios, rw)
    out, attn = m(query, value, shapes, v_mask, lsi, ratios, rw)
    save("G7_module_box", **pack(m, dict(query=query, value=value, v_mask=v_mask, ratios=ratios,
         ref_windows=rw, shapes=shapes, lsi=lsi, grid=grid, attn=attn, out=out)))

    # BoxAttention with per-head reference windows (B, Lq, H, 4) and no valid ratios
    rw_h = torch.rand(B, Lq, nh, 4, generator=torch.Generator().manual_seed(72))
    grid = m._where_to_attend(query, None, rw_h)
    out, attn = m(query, value, shapes, None, lsi, None, rw_h)
    save("G7_module_box_perhead", **pack(m, dict(query=query, value=value, ref_windows=rw_h,
         shapes=shapes, lsi=lsi, grid=grid, attn=attn, out=out)))

    # InstanceAttention, training branch (kernel 4) and inference branch
    for ks in (4, 14):
        m = ref.InstanceAttention(d, nl, nh, kernel_size=ks).double()
        randomise(m, 73 + ks)
        m.inferencing = False
        grid = m._where_to_attend(query, ratios, rw)
        out, mask_out, (sw, lw) = m(query, value, shapes, v_mask, lsi, ratios, rw)
        m.inferencing = True
        out_inf, none_mask, (sw_inf,) = m(query, value, shapes, v_mask, lsi, ratios, rw)
        assert none_mask is None
        save("G7_module_inst_k%d" % ks, **pack(m, dict(query=query, value=value, v_mask=v_mask,
             ratios=ratios, ref_windows=rw, shapes=shapes, lsi=lsi, grid=grid, spatial_w=sw,
             level_w=lw, out=out, mask_out=mask_out, out_inferencing=out_inf)))

    # Box3dAttention: learned rotation with (B, Lq, 7) windows; fixed per-head angles with
    # (B, Lq, H, 5) windows (the 3D encoder case, box3d_transformer.py:62-77)
    m = ref.Box3dAttention(d, nl, nh, with_rotation=True, kernel_size=2).double()
    randomise(m, 80)
    query, value, v_mask, ratios, rw7 = common(81, 7)
    grid = m._where_to_attend(query, ratios, rw7)
    out, attn = m(query, value, shapes, v_mask, lsi, ratios, rw7)
    save("G7_module_box3d_rot", **pack(m, dict(query=query, value=value, v_mask=v_mask,
         ratios=ratios, ref_windows=rw7, shapes=shapes, lsi=lsi, grid=grid, attn=attn, out=out)))

    m = ref.Box3dAttention(d, nl, nh, with_rotation=False, kernel_size=3).double()
    randomise(m, 82)
    g = torch.Generator().manual_seed(83)
    rw5 = torch.rand(B, Lq, nh, 5, generator=g)
    grid = m._where_to_attend(query, None, rw5)
    out, attn = m(query, value, shapes, None, lsi, None, rw5)
    save("G7_module_box3d_fixed", **pack(m, dict(query=query, value=value, ref_windows=rw5,
         shapes=shapes, lsi=lsi, grid=grid, attn=attn, out=out)))


# --------------------------------------------------------------------------------------
# G8: transformer LAYERS around the modules: the reference's
# BoxTransformerEncoderLayer / DecoderLayer (box_transformer.py:316-465) and the BoxeR-3D
# Box3dTransformerEncoderLayer / DecoderLayer with the encoder's reference windows of 8 fixed
# per-head angles (box3d_transformer.py:62-109, 230-322), run on small maps in fp64.
# --------------------------------------------------------------------------------------
def load_reference_transformers():
    import copy
    box_attention = load_reference_modules()
    sys.modules["e2edet.module.box_attention"] = box_attention
    sys.modules["e2edet.module"].__path__ = []            # a package: relative imports resolve
    general = types.ModuleType("e2edet.utils.general")
    general.__dict__.update({"torch": torch, "F": F, "math": math, "nn": torch.nn, "copy": copy})
    _extract(os.path.join(REF, "e2edet/utils/general.py"),
             ["flatten_with_shape", "inverse_sigmoid", "get_clones", "get_activation_fn",
              "get_proposal_pos_embed", "normalize_period"], general.__dict__)
    sys.modules.setdefault("e2edet.utils", types.ModuleType("e2edet.utils"))
    sys.modules["e2edet.utils.general"] = general
    mods = []
    for name in ("box_transformer", "box3d_transformer"):
        spec = importlib.util.spec_from_file_location(
            "e2edet.module." + name, os.path.join(REF, "e2edet/module/%s.py" % name))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["e2edet.module." + name] = mod
        spec.loader.exec_module(mod)
        mods.append(mod)
    return mods


def g8():
    tr2d, tr3d = load_reference_transformers()
    d, nh, ff, B, Lq = 32, 4, 48, 2, 6

    def randomise(m, seed):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)

    def pack(m, extra):
        sd = {"sd." + k: v for k, v in m.state_dict().items()}
        sd.update(extra)
        return sd

    # ---- 2D: encoder layer over a 2-level map with padding mask + valid ratios
    levels = [(7, 6), (4, 3)]
    shapes = torch.tensor(levels, dtype=torch.long)
    lsi = lsi_of(shapes)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(90)
    src = torch.randn(B, S, d, generator=g)
    pos = 0.3 * torch.randn(B, S, d, generator=g)
    mask = torch.rand(B, S, generator=g) < 0.1
    ratios = 0.7 + 0.3 * torch.rand(B, 1, 1, len(levels), 1, 2, generator=g)
    enc_ref = torch.rand(B, S, 4, generator=g)
    enc_ref[..., 2:] = 0.1 + 0.3 * enc_ref[..., 2:]
    enc = tr2d.BoxTransformerEncoderLayer(d, nh, len(levels), ff, 0.0, "relu").double()
    randomise(enc, 91)
    memory = enc(src, pos, shapes, mask, lsi, ratios, enc_ref)
    save("G8_layer_enc2d", **pack(enc, dict(src=src, pos=pos, v_mask=mask, ratios=ratios,
         ref_windows=enc_ref, shapes=shapes, lsi=lsi, out=memory)))

    # ---- 2D: decoder layers (detection: BoxAttention; instance segmentation: InstanceAttention
    #      14x14, training branch, both residual modes)
    tgt = torch.randn(B, Lq, d, generator=g)
    qpos = 0.3 * torch.randn(B, Lq, d, generator=g)
    dec_ref = torch.rand(B, Lq, 4, generator=g)
    dec_ref[..., 2:] = 0.1 + 0.4 * dec_ref[..., 2:]
    dec = tr2d.BoxTransformerDecoderLayer(d, nh, len(levels), ff, 0.0, "relu", False, "v1").double()
    randomise(dec, 92)
    out, roi = dec(tgt, qpos, memory.detach(), shapes, mask, lsi, ratios, dec_ref)
    assert roi is None
    save("G8_layer_dec2d", **pack(dec, dict(tgt=tgt, query_pos=qpos, memory=memory, v_mask=mask,
         ratios=ratios, ref_windows=dec_ref, shapes=shapes, lsi=lsi, out=out)))
    for mode in ("v1", "v2"):
        dec = tr2d.BoxTransformerDecoderLayer(d, nh, len(levels), ff, 0.0, "relu", True, mode).double()
        randomise(dec, 93)
        dec.inferencing = False
        dec.multihead_attn.inferencing = False
        out, roi = dec(tgt, qpos, memory.detach(), shapes, mask, lsi, ratios, dec_ref)
        save("G8_layer_dec2d_mask_%s" % mode, **pack(dec, dict(
            tgt=tgt, query_pos=qpos, memory=memory, v_mask=mask, ratios=ratios,
            ref_windows=dec_ref, shapes=shapes, lsi=lsi, out=out, roi=roi)))

    # ---- 2D: the encoder's reference windows, with and without padding masks
    masks = []
    for (h, w) in levels:
        m = torch.zeros(B, h, w, dtype=torch.bool)
        m[0, :, w - 2:] = True                    # image 0: two padded columns
        m[1, h - 1:, :] = True                    # image 1: one padded row
        masks.append(m)
    maps2 = [torch.zeros(B, d, h, w) for h, w in levels]
    holder2 = types.SimpleNamespace(ref_size=4)
    rw_masked = tr2d.BoxTransformer._create_ref_windows(holder2, maps2, masks)
    rw_plain = tr2d.BoxTransformer._create_ref_windows(holder2, maps2, None)
    save("G8_refwin2d", shapes=shapes, masks=torch.cat([m.flatten(1) for m in masks], 1),
         ref_masked=rw_masked, ref_plain=rw_plain)

    # ---- 3D (BEV): the encoder's reference windows (8 fixed per-head angles) + one encoder
    #      layer; one decoder layer with learned rotation
    nh3 = 8
    levels3 = [(6, 6), (3, 3)]
    shapes3 = torch.tensor(levels3, dtype=torch.long)
    lsi3 = lsi_of(shapes3)
    S3 = int(shapes3.prod(1).sum())
    maps = [torch.zeros(B, d, h, w) for h, w in levels3]
    holder = types.SimpleNamespace(ref_size=4)
    ref3 = tr3d.Box3dTransformer._create_ref_windows(holder, maps)             # (B, S, 8, 5)
    src3 = torch.randn(B, S3, d, generator=g)
    pos3 = 0.3 * torch.randn(B, S3, d, generator=g)
    enc3 = tr3d.Box3dTransformerEncoderLayer(d, nh3, len(levels3), ff, 0.0, "relu").double()
    randomise(enc3, 94)
    mem3 = enc3(src3, pos3, shapes3, lsi3, ref3)
    save("G8_layer_enc3d", **pack(enc3, dict(src=src3, pos=pos3, ref_windows=ref3, shapes=shapes3,
         lsi=lsi3, out=mem3)))
    dec_ref3 = torch.rand(B, Lq, 7, generator=g)
    dec_ref3[..., 2:4] = 0.1 + 0.4 * dec_ref3[..., 2:4]
    dec3 = tr3d.Box3dTransformerDecoderLayer(d, nh3, len(levels3), ff, 0.0, "relu").double()
    randomise(dec3, 95)
    out3 = dec3(tgt, qpos, mem3.detach(), shapes3, lsi3, dec_ref3)
    save("G8_layer_dec3d", **pack(dec3, dict(tgt=tgt, query_pos=qpos, memory=mem3,
         ref_windows=dec_ref3, shapes=shapes3, lsi=lsi3, out=out3)))


# --------------------------------------------------------------------------------------
# G9: the modules at the MODEL's head geometry -- d = 256, 8 heads (32 channels per head), 4 levels, 2 x 2
#     points -- with GRADIENTS: outputs and d(loss)/d(query, value, ref_windows, every parameter) for a stored
#     upstream gradient, differentiated by torch in fp64 through the reference's own module code
#     (box_attention.py:63-81, 196-239, 304-363).  G7 / G8 use d = 32 with 4 heads (8 channels per head), which
#     only the generic kernels take; these shapes are the ones the window-staged / gather / binned kernel
#     families run (encoder shape: one query per pixel, pixel-centre reference windows).
#     Storage: inputs and parameters are int8 / 64 resp. int8 / 2048 (exact in bf16 / fp16: stored as float16),
#     expected outputs and gradients float32 (the GPU paths compared with them are float32 / bf16).
# --------------------------------------------------------------------------------------
def g9():
    ref = load_reference_modules()
    levels = [(8, 12), (4, 6), (2, 3), (1, 2)]
    shapes = torch.tensor(levels, dtype=torch.long)
    lsi = lsi_of(shapes)
    S = int(shapes.prod(1).sum())
    d, nl, nh, B, Lq_dec = 256, 4, 8, 2, 12

    def q8(gen, shape, scale):
        return torch.randint(-127, 128, shape, generator=gen, dtype=torch.int64).double() * scale

    def randomise(m, seed):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(q8(g, tuple(p.shape), 1.0 / 2048))

    def pixel_windows(extra=()):
        """encoder reference windows: one per pixel, centred on it, 4 pixels wide and high (box_transformer.py:70-116)"""
        rows = []
        for h, w in levels:
            ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64),
                                    indexing="ij")
            r = torch.stack([(xs + 0.5) / w, (ys + 0.5) / h, torch.full_like(xs, 4.0 / w),
                             torch.full_like(ys, 4.0 / h)], -1).reshape(-1, 4)
            rows.append(r)
        rw = torch.cat(rows, 0)[None].repeat(B, 1, 1)
        return rw

    def run(name, m, query, value, v_mask, ratios, rw, n_out, seed):
        g = torch.Generator().manual_seed(seed)
        m.zero_grad(set_to_none=True)                  # (a module serves several fixtures)
        q = query.clone().requires_grad_(True)
        v = value.clone().requires_grad_(True)
        r = rw.clone().requires_grad_(True)
        outs = m(q, v, shapes, v_mask, lsi, ratios, r)
        outs = [o for o in outs[:n_out]]
        gouts = [q8(g, tuple(o.shape), 1.0 / 64) for o in outs]
        sum((o * go).sum() for o, go in zip(outs, gouts)).backward()
        arrs = {"sd." + k: t.to(torch.float16) for k, t in m.state_dict().items()}
        for k, t in m.state_dict().items():            # (the float16 copies are exact)
            assert torch.equal(arrs["sd." + k].double(), t.double()), k
        arrs.update(shapes=shapes, lsi=lsi, query=query.to(torch.float16), value=value.to(torch.float16),
                    ref_windows=rw, grad_query=q.grad.float(), grad_value=v.grad.float(),
                    grad_ref_windows=r.grad.float())
        if v_mask is not None:
            arrs["v_mask"] = v_mask
        if ratios is not None:
            arrs["ratios"] = ratios
        for i, (o, go) in enumerate(zip(outs, gouts)):
            arrs["out%d" % i] = o.detach().float()
            arrs["gout%d" % i] = go.to(torch.float16)
        for k, p in m.named_parameters():
            arrs["grad." + k] = p.grad.float()
        save(name, **arrs)

    def inputs(seed, Lq):
        g = torch.Generator().manual_seed(seed)
        return (q8(g, (B, Lq, d), 1.0 / 64), q8(g, (B, S, d), 1.0 / 64), torch.rand(B, S, generator=g) < 0.1,
                0.7 + 0.3 * torch.rand(B, 1, 1, nl, 1, 2, generator=g), g)

    # BoxAttention, encoder shape (one query per pixel, pixel-centre windows, no mask / ratios as in box_transformer.py:330)
    m = ref.BoxAttention(d, nl, nh, kernel_size=2).double()
    randomise(m, 100)
    query, value, v_mask, ratios, g = inputs(101, S)
    run("G9_module_box_enc", m, query, value, None, None, pixel_windows(), 1, 102)
    # ... and with padded pixels + valid ratios
    run("G9_module_box_enc_masked", m, query, value, v_mask, ratios, pixel_windows(), 1, 103)
    # BoxAttention, decoder shape
    query, value, v_mask, ratios, g = inputs(104, Lq_dec)
    rw = torch.rand(B, Lq_dec, 4, generator=g)
    rw[..., 2:4] = 0.1 + 0.4 * rw[..., 2:4]
    run("G9_module_box_dec", m, query, value, v_mask, ratios, rw, 1, 105)
    # Box3dAttention, encoder shape with the 8 fixed per-head angles (box3d_transformer.py:62-77)
    m = ref.Box3dAttention(d, nl, nh, with_rotation=False, kernel_size=2).double()
    randomise(m, 106)
    query, value, v_mask, ratios, g = inputs(107, S)
    ang = (torch.arange(nh, dtype=torch.float64) / nh)[None, None, :, None].expand(B, S, nh, 1)
    rw5 = torch.cat([pixel_windows()[:, :, None, :].expand(B, S, nh, 4), ang], -1).contiguous()
    run("G9_module_box3d_fixed_enc", m, query, value, None, None, rw5, 1, 108)
    # Box3dAttention, decoder shape with learned rotation
    m = ref.Box3dAttention(d, nl, nh, with_rotation=True, kernel_size=2).double()
    randomise(m, 109)
    query, value, v_mask, ratios, g = inputs(110, Lq_dec)
    rw7 = torch.rand(B, Lq_dec, 7, generator=g)
    rw7[..., 2:4] = 0.1 + 0.4 * rw7[..., 2:4]
    run("G9_module_box3d_rot_dec", m, query, value, v_mask, ratios, rw7, 1, 111)
    # InstanceAttention (training branch: output + mask output), decoder shape
    m = ref.InstanceAttention(d, nl, nh, kernel_size=4).double()
    randomise(m, 112)
    m.inferencing = False
    query, value, v_mask, ratios, g = inputs(113, Lq_dec)
    rw = torch.rand(B, Lq_dec, 4, generator=g)
    rw[..., 2:4] = 0.1 + 0.4 * rw[..., 2:4]
    run("G9_module_inst_k4", m, query, value, v_mask, ratios, rw, 2, 114)


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference checkout not found (run in the build container)"
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9):
        if not only or fn.__name__ in only:
            fn()
