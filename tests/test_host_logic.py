"""CPU: host-side logic of the drop-in layer -- argument checks of boxer_amd.ops (same error
behaviour as the reference host code) and the nn.Module geometry / state_dict surface pinned
to the reference goldens G7 (the op itself is replaced by the CPU oracle here; the GPU run
of the same fixtures is tests/test_gpu_parity.py::test_modules_match_reference_goldens)."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import torch_fallback as tf


def test_ops_reject_cpu_tensors_like_the_reference():
    from boxer_amd import ops
    shapes = torch.tensor([[3, 2]])
    lsi = torch.tensor([0])
    value = torch.zeros(1, 6, 2, 4)
    loc = torch.zeros(1, 1, 2, 1, 2, 2)
    attn = torch.zeros(1, 1, 2, 1, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):   # box_attn.h:53
        ops.box_attn_forward(value, shapes, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.box_attn_backward(value, shapes, lsi, loc, attn, torch.zeros(1, 1, 8), 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.instance_attn_forward(value, shapes, lsi, loc, attn, attn, 64)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from boxer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


class _OracleBox:
    @staticmethod
    def apply(value, shapes, lsi, loc, attn, im2col_step):
        return tf.box_attn(value, shapes, loc, attn)


class _OracleInst:
    @staticmethod
    def apply(value, shapes, lsi, loc, sw, lw, mask_size, im2col_step):
        out, mask = tf.instance_attn(value, shapes, loc, sw, lw)
        b, l, _, c = mask.shape
        return out, mask.view(b, l, mask_size, mask_size, c)


@pytest.fixture
def oracle_functions(monkeypatch):
    from boxer_amd import modules
    monkeypatch.setattr(modules, "BoxAttnFunction", _OracleBox)
    monkeypatch.setattr(modules, "InstanceAttnFunction", _OracleInst)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _load(cls, g, **kw):
    m = cls(**kw).double()
    sd = {k[3:]: _t(v) for k, v in g.items() if k.startswith("sd.")}
    missing = m.load_state_dict(sd, strict=True)        # reference checkpoint keys load as-is
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


def _args(g, mask=True, ratio=True):
    return (_t(g["query"]), _t(g["value"]), _t(g["shapes"]), _t(g["v_mask"]) if mask else None,
            _t(g["lsi"]), _t(g["ratios"]) if ratio else None, _t(g["ref_windows"]))


TOL = dict(rtol=1e-10, atol=1e-11)


def test_box_attention_module(oracle_functions):
    from boxer_amd import BoxAttention
    g = golden_io.load("G7_module_box")
    m = _load(BoxAttention, g, d_model=32, num_level=2, num_head=4, kernel_size=2)
    a = _args(g)
    grid = m._where_to_attend(a[0], a[5], a[6])
    np.testing.assert_allclose(grid.detach().numpy(), g["grid"], **TOL)
    out, attn = m(*a)
    np.testing.assert_allclose(attn.detach().numpy(), g["attn"], **TOL)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)

    g = golden_io.load("G7_module_box_perhead")
    a = _args(g, False, False)
    grid = m._where_to_attend(a[0], None, a[6])
    np.testing.assert_allclose(grid.detach().numpy(), g["grid"], **TOL)
    out, _ = m(*a)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)


@pytest.mark.parametrize("ks", [4, 14])
def test_instance_attention_module(oracle_functions, ks):
    from boxer_amd import InstanceAttention
    g = golden_io.load("G7_module_inst_k%d" % ks)
    m = _load(InstanceAttention, g, d_model=32, num_level=2, num_head=4, kernel_size=ks)
    a = _args(g)
    np.testing.assert_allclose(m._where_to_attend(a[0], a[5], a[6]).detach().numpy(), g["grid"],
                               **TOL)
    with pytest.raises(AttributeError):       # `inferencing` is injected by the model
        m(*a)
    m.inferencing = False
    out, mask_out, (sw, lw) = m(*a)
    np.testing.assert_allclose(sw.detach().numpy(), g["spatial_w"], **TOL)
    np.testing.assert_allclose(lw.detach().numpy(), g["level_w"], **TOL)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    np.testing.assert_allclose(mask_out.detach().numpy(), g["mask_out"], **TOL)
    m.inferencing = True
    out, mask_out, weights = m(*a)
    assert mask_out is None and len(weights) == 1
    np.testing.assert_allclose(out.detach().numpy(), g["out_inferencing"], **TOL)


def test_box3d_attention_module(oracle_functions):
    from boxer_amd import Box3dAttention
    g = golden_io.load("G7_module_box3d_rot")
    m = _load(Box3dAttention, g, d_model=32, num_level=2, num_head=4, with_rotation=True,
              kernel_size=2)
    a = _args(g)
    np.testing.assert_allclose(m._where_to_attend(a[0], a[5], a[6]).detach().numpy(), g["grid"],
                               **TOL)
    out, attn = m(*a)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)

    g = golden_io.load("G7_module_box3d_fixed")
    m = _load(Box3dAttention, g, d_model=32, num_level=2, num_head=4, with_rotation=False,
              kernel_size=3)
    a = _args(g, False, False)
    np.testing.assert_allclose(m._where_to_attend(a[0], None, a[6]).detach().numpy(), g["grid"],
                               **TOL)
    out, attn = m(*a)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)


def test_module_surface_matches_reference():
    """Constructor args, parameter / buffer names and shapes, init (box_attention.py:186-194)."""
    from boxer_amd import Box3dAttention, BoxAttention, InstanceAttention
    torch.manual_seed(0)
    d, L, H = 256, 4, 8
    m = BoxAttention(d, L, H)
    sd = m.state_dict()
    assert sorted(sd) == sorted([
        "linear_box_weight", "linear_box_bias", "linear_attn_weight", "linear_attn_bias",
        "value_proj.weight", "value_proj.bias", "out_proj.weight", "out_proj.bias",
        "kernel_indices"])
    assert sd["linear_box_weight"].shape == (L * H * 4, d)
    assert sd["linear_attn_weight"].shape == (H * L * 4, d)
    assert sd["kernel_indices"].tolist() == [[-0.25, -0.25], [0.25, -0.25], [-0.25, 0.25],
                                             [0.25, 0.25]]
    assert m.im2col_step == 64 and m.num_point == 4 and m.head_dim == 32
    assert not sd["linear_box_weight"].any() and not sd["linear_attn_weight"].any()
    assert (sd["linear_box_bias"] >= 0).all() and (sd["linear_box_bias"] <= 1).all()
    assert not sd["value_proj.bias"].any()

    m = InstanceAttention(d, L, H, 14)
    assert m.state_dict()["linear_attn_weight"].shape == (H * L * 4, d)
    assert m.state_dict()["kernel_indices"].shape == (196, 2)
    assert not hasattr(m, "inferencing")

    m = Box3dAttention(d, 2, H, with_rotation=True)
    assert m.state_dict()["linear_box_weight"].shape == (2 * H * 5, d)
    assert m.num_variable == 5
    assert m.state_dict()["kernel_indices"].tolist() == [[-0.25, -0.25], [0.25, -0.25],
                                                         [-0.25, 0.25], [0.25, 0.25]]
    m = Box3dAttention(d, 2, H, with_rotation=False, kernel_size=3)
    assert m.state_dict()["linear_box_weight"].shape == (2 * H * 4, d)
    assert m.state_dict()["kernel_indices"][0].tolist() == [-0.5, -0.5]     # /2, not /k


# ------------------------------------------------------------------ box -> grid op (opt-in)
def test_box_grid_op_rejects_bad_arguments():
    from boxer_amd import ops
    """Host-side validation of ops.box_grid_forward / _backward (no GPU needed: CPU tensors are
    refused first, exactly like the four reference entry points)."""
    ref = torch.rand(2, 5, 4)
    off = torch.randn(2, 5, 3, 2, 4)
    kidx = torch.rand(4, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.box_grid_forward(ref, off, kidx)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.box_grid_backward(ref, off, kidx, None, 0, torch.rand(2, 5, 3, 2, 4, 2))


def test_modules_fused_grid_flag_is_inert_on_cpu(oracle_functions):
    """`fused_grid = True` only takes effect for CUDA tensors; everything else silently keeps the
    reference's torch path (same outputs as without the flag)."""
    from boxer_amd import BoxAttention
    torch.manual_seed(3)
    m = BoxAttention(32, 2, 4, 2)
    with torch.no_grad():
        m.linear_box_weight.normal_(0, 0.05)
    shapes = torch.tensor([[6, 5], [3, 2]])
    lsi = torch.tensor([0, 30])
    q, v = torch.randn(2, 7, 32), torch.randn(2, 36, 32)
    ref_w = torch.rand(2, 7, 4)
    out0 = m(q, v, shapes, None, lsi, None, ref_w)[0]
    m.fused_grid = True
    out1 = m(q, v, shapes, None, lsi, None, ref_w)[0]
    assert torch.equal(out0, out1)


def test_bench_algorithmic_bytes():
    """60 B (bf16) / 76 B (fp32) per sample point at C2."""
    import bench
    dims = dict(B=2, S=13294, H=8, C=32, L=4, Lq=13294, P=4)
    npts = bench.n_points(dims)
    assert npts == 3403264
    for elem, per_point in ((2, 60.0), (4, 76.0)):
        fwd, bwd, per_kernel = bench.algorithmic_bytes(dims, "box", elem)
        assert abs((fwd + bwd) / npts - per_point) < 0.01
        assert set(per_kernel) >= {"fwd", "bwd_points", "bwd_accumulate"}
