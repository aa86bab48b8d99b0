"""The callers of the operator (boxer_amd.layers) against the G8 goldens,
which tests/golden/make_goldens.py:g8 produced by running the reference's own
BoxTransformer{Encoder,Decoder}Layer / Box3dTransformer{Encoder,Decoder}Layer and
_create_ref_windows (e2edet/module/box_transformer.py:70-116, 316-465; box3d_transformer.py:62-109,
230-322).  CPU part: reference-window builders, pillar scatter, checkpoint compatibility.  GPU part
(the layers call the HIP operator): layer outputs in float64."""
import numpy as np
import pytest
import torch

import golden_io
from boxer_amd import layers

LAYER_CASES = {
    "G8_layer_enc2d": lambda d, nh, nl, ff: layers.BoxTransformerEncoderLayer(d, nh, nl, ff, 0.0, "relu"),
    "G8_layer_dec2d": lambda d, nh, nl, ff: layers.BoxTransformerDecoderLayer(d, nh, nl, ff, 0.0, "relu", False, "v1"),
    "G8_layer_dec2d_mask_v1": lambda d, nh, nl, ff: layers.BoxTransformerDecoderLayer(d, nh, nl, ff, 0.0, "relu", True, "v1"),
    "G8_layer_dec2d_mask_v2": lambda d, nh, nl, ff: layers.BoxTransformerDecoderLayer(d, nh, nl, ff, 0.0, "relu", True, "v2"),
    "G8_layer_enc3d": lambda d, nh, nl, ff: layers.Box3dTransformerEncoderLayer(d, nh, nl, ff, 0.0, "relu"),
    "G8_layer_dec3d": lambda d, nh, nl, ff: layers.Box3dTransformerDecoderLayer(d, nh, nl, ff, 0.0, "relu"),
}


def build(name):
    g = golden_io.load(name)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    d = sd["linear1.weight"].shape[1]
    ff = sd["linear1.weight"].shape[0]
    nl = int(g["shapes"].shape[0])
    nh = 8 if "3d" in name else 4
    layer = LAYER_CASES[name](d, nh, nl, ff).double()
    layer.load_state_dict(sd, strict=True)          # same parameter / buffer names as the reference
    return layer, g


@pytest.mark.parametrize("name", sorted(LAYER_CASES))
def test_reference_checkpoints_load(name):
    layer, g = build(name)
    assert sum(p.numel() for p in layer.parameters()) > 0


def test_encoder_reference_windows_2d():
    g = golden_io.load("G8_refwin2d")
    levels = [tuple(int(x) for x in r) for r in g["shapes"]]
    B = g["ref_plain"].shape[0]
    got = layers.encoder_ref_windows_2d(levels, B, dtype=torch.float64)
    np.testing.assert_allclose(got.numpy(), g["ref_plain"], rtol=0, atol=1e-12)
    flat = torch.from_numpy(g["masks"])
    masks, o = [], 0
    for (h, w) in levels:
        masks.append(flat[:, o:o + h * w].view(B, h, w))
        o += h * w
    got = layers.encoder_ref_windows_2d(levels, B, dtype=torch.float64, masks=masks)
    np.testing.assert_allclose(got.numpy(), g["ref_masked"], rtol=0, atol=1e-12)


def test_encoder_reference_windows_3d():
    g = golden_io.load("G8_layer_enc3d")
    levels = [tuple(int(x) for x in r) for r in g["shapes"]]
    got = layers.encoder_ref_windows_3d(levels, g["ref_windows"].shape[0], dtype=torch.float64)
    assert got.shape == g["ref_windows"].shape == (2, sum(h * w for h, w in levels), 8, 5)
    # the reference builds the angle table in float32 (torch.FloatTensor)
    np.testing.assert_allclose(got.numpy(), g["ref_windows"], rtol=0, atol=1e-7)


def test_pillar_scatter_builds_the_bev_canvas():
    """PointPillarsScatter semantics (point_pillar.py:20-67): canvas[b, :, y, x] = features of
    the pillar at (b, ., y, x), zeros elsewhere; then the BoxeR-3D level shapes follow by
    striding the 468 x 468 canvas (base_boxer3d_detection.yaml:27-37, neck strides 2, 2)."""
    g = torch.Generator().manual_seed(0)
    B, C, nx, ny, N = 2, 5, 12, 9, 40
    cells = torch.randperm(B * ny * nx, generator=g)[:N]
    coords = torch.stack([cells // (ny * nx), torch.zeros_like(cells), (cells // nx) % ny,
                          cells % nx], 1)
    feats = torch.randn(N, C, generator=g)
    canvas = layers.pillar_scatter(feats, coords, B, nx, ny)
    assert canvas.shape == (B, C, ny, nx)
    want = torch.zeros(B, C, ny, nx)
    for i in range(N):
        b, _, y, x = (int(v) for v in coords[i])
        want[b, :, y, x] = feats[i]
    assert torch.equal(canvas, want)
    assert int((canvas.abs().sum(1) > 0).sum()) == N


# ------------------------------------------------------------------------------------ GPU
def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(LAYER_CASES))
def test_layer_matches_reference_golden(name):
    layer, g = build(name)
    layer = layer.cuda()
    shapes, lsi = _dev(g["shapes"]), _dev(g["lsi"])
    ref = _dev(g["ref_windows"])
    if "enc2d" in name:
        out = layer(_dev(g["src"]), _dev(g["pos"]), shapes, _dev(g["v_mask"]), lsi,
                    _dev(g["ratios"]), ref)
    elif "enc3d" in name:
        out = layer(_dev(g["src"]), _dev(g["pos"]), shapes, lsi, ref)
    elif "dec3d" in name:
        out = layer(_dev(g["tgt"]), _dev(g["query_pos"]), _dev(g["memory"]), shapes, lsi, ref)
    else:
        if "mask" in name:
            layer.inferencing = False
            layer.multihead_attn.inferencing = False
        out, roi = layer(_dev(g["tgt"]), _dev(g["query_pos"]), _dev(g["memory"]), shapes,
                         _dev(g["v_mask"]), lsi, _dev(g["ratios"]), ref)
        if "mask" in name:
            np.testing.assert_allclose(roi.detach().cpu().numpy(), g["roi"], rtol=0, atol=1e-9)
        else:
            assert roi is None
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=0, atol=1e-9)


def _run_layer(name, layer, g, cast=lambda t: t):
    shapes, lsi = _dev(g["shapes"]), _dev(g["lsi"])
    f = lambda k: cast(_dev(g[k]))
    ref = f("ref_windows")
    if "enc2d" in name:
        return layer(f("src"), f("pos"), shapes, _dev(g["v_mask"]), lsi, f("ratios"), ref)
    if "enc3d" in name:
        return layer(f("src"), f("pos"), shapes, lsi, ref)
    if "dec3d" in name:
        return layer(f("tgt"), f("query_pos"), f("memory"), shapes, lsi, ref)
    if "mask" in name:
        layer.inferencing = False
        layer.multihead_attn.inferencing = False
    return layer(f("tgt"), f("query_pos"), f("memory"), shapes, _dev(g["v_mask"]), lsi, f("ratios"), ref)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("fused_grid,fused_pointwise,native_bf16",
                         [(1, False, False), (0, True, False), (1, True, False), (1, True, True)],
                         ids=["grid", "pointwise", "grid+pointwise", "grid+pointwise+bf16"])
@pytest.mark.parametrize("name", sorted(LAYER_CASES))
def test_layer_with_fused_paths_matches_reference_golden(name, fused_grid, fused_pointwise, native_bf16):
    """The opt-in paths of SURVEY.md 8(f) N1 / N3 (grid kernels,
    one-pass softmax / mask-fill + cast, bf16 storage) inside the reference's encoder / decoder layers,
    against the goldens the reference's own layer classes produced (G8) -- float32 / bf16 runs, so at
    those types' tolerances."""
    from boxer_amd.modules import _BoxAttentionBase as _AttentionBase
    layer, g = build(name)
    layer = layer.cuda().float()
    for m in layer.modules():
        if isinstance(m, _AttentionBase):
            m.fused_grid, m.fused_pointwise, m.native_bf16 = fused_grid, fused_pointwise, native_bf16
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=native_bf16):
        out = _run_layer(name, layer, g, cast=lambda t: t.float() if t.is_floating_point() else t)
    want = g["out"]
    err = float(np.abs(out.detach().double().cpu().numpy() - want).max()) / max(1.0, float(np.abs(want).max()))
    assert err <= (3e-2 if native_bf16 else 2e-4), (name, err)


@pytest.mark.gpu
def test_bev_encoder_layer_runs_on_pillar_features():
    """pillar features -> BEV canvas -> two levels -> one BoxeR-3D encoder layer
    (8 fixed per-head angles) forward + backward in the bf16 storage mode; checked against the
    float64 run of the same layer (the float64 path is pinned by G8_layer_enc3d)."""
    torch.manual_seed(0)
    B, d, nx, ny = 2, 64, 48, 48
    n_pillars = 900
    cells = torch.randperm(B * ny * nx)[:n_pillars]
    coords = torch.stack([cells // (ny * nx), torch.zeros_like(cells), (cells // nx) % ny,
                          cells % nx], 1).cuda()
    feats = torch.randn(n_pillars, d, device="cuda")
    canvas = layers.pillar_scatter(feats, coords, B, nx, ny)                 # (B, d, 48, 48)
    maps = [torch.nn.functional.avg_pool2d(canvas, 2), torch.nn.functional.avg_pool2d(canvas, 4)]
    levels = [tuple(m.shape[-2:]) for m in maps]                             # 24 x 24, 12 x 12
    src = torch.cat([m.flatten(2).transpose(1, 2) for m in maps], 1)
    shapes = torch.tensor(levels, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    ref = layers.encoder_ref_windows_3d(levels, B, device="cuda")
    pos = 0.1 * torch.randn_like(src)
    layer = layers.Box3dTransformerEncoderLayer(d, 8, 2, 128, 0.0, "relu").cuda()
    with torch.no_grad():
        layer.self_attn.linear_box_weight.normal_(0, 0.05)
        layer.self_attn.linear_attn_weight.normal_(0, 0.05)
    want_layer = layers.Box3dTransformerEncoderLayer(d, 8, 2, 128, 0.0, "relu").cuda().double()
    want_layer.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    want = want_layer(src.double(), pos.double(), shapes, lsi, ref.double())
    layer.self_attn.native_bf16 = True
    x = src.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = layer(x, pos, shapes, lsi, ref)
    out.float().square().mean().backward()
    assert torch.isfinite(x.grad).all()
    err = (out.double() - want).abs().max().item()
    assert err <= 5e-2 * max(1.0, want.abs().max().item()), err


def test_split_k_weight_gradient_matches_linear():
    """boxer_amd.dense: the batched split-K weight gradient (and the bias gradient taken in the
    same backward) equal F.linear's, ragged tail included."""
    import torch.nn.functional as F
    from boxer_amd import dense
    torch.manual_seed(0)
    x = torch.randn(3, 1111, 24, dtype=torch.float64, requires_grad=True)
    w = torch.randn(40, 24, dtype=torch.float64, requires_grad=True)
    b = torch.randn(40, dtype=torch.float64, requires_grad=True)
    g = torch.randn(3, 1111, 40, dtype=torch.float64)
    want = torch.autograd.grad(F.linear(x, w, b), (x, w, b), g)
    old = dense.CHUNK_ROWS
    dense.CHUNK_ROWS = 256                          # 3333 rows: 13 chunks of 256 + a tail of 5
    try:
        y = dense._SplitKLinear.apply(x, w, b)
        got = torch.autograd.grad(y, (x, w, b), g)
    finally:
        dense.CHUNK_ROWS = old
    assert torch.allclose(y, F.linear(x, w, b))
    for a, c in zip(got, want):
        assert torch.allclose(a, c, rtol=1e-10, atol=1e-10)
    # off (the default), on CPU or with few rows: plain F.linear
    assert "SplitK" not in dense.linear(x, w, b).grad_fn.name()


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [False, True])
def test_split_k_linear_on_gpu(amp):
    """Same outputs; weight / bias gradients equal F.linear's to the rounding of the partial
    products (bf16 under autocast), measured against a float64 reference."""
    import torch.nn.functional as F
    from boxer_amd import dense
    torch.manual_seed(1)
    x = torch.randn(2, 9000, 256, device="cuda", requires_grad=True)
    w = (0.05 * torch.randn(128, 256, device="cuda")).requires_grad_()
    b = torch.randn(128, device="cuda", requires_grad=True)
    g = torch.randn(2, 9000, 128, device="cuda")
    ref = torch.autograd.grad(F.linear(x.double(), w.double(), b.double()), (x, w, b), g.double())
    old = dense.set_split_k(True)
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y = dense.linear(x, w, b)
            y0 = F.linear(x, w, b)
        assert "SplitK" in y.grad_fn.name()
        got = torch.autograd.grad(y, (x, w, b), g.to(y.dtype))
        base = torch.autograd.grad(y0, (x, w, b), g.to(y0.dtype))
    finally:
        dense.set_split_k(old)
    assert y.dtype == y0.dtype and torch.equal(y, y0)
    for a, c, r in zip(got, base, ref):
        assert a.dtype == c.dtype
        scale = float(r.abs().max())
        err_new, err_lib = float((a.double() - r).abs().max()), float((c.double() - r).abs().max())
        assert err_new <= max(2.0 * err_lib, (4e-3 if amp else 2e-5) * scale), (err_new, err_lib)
