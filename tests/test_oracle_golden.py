"""CPU: pin the oracles (C restatement + grid_sample fallback) to the reference goldens.

The goldens are fp64 outputs of the reference's own test oracle (see
tests/golden/make_goldens.py), so a pass here means "oracle == reference on these inputs".
"""
import numpy as np
import pytest
import torch

import golden_io
from oracle import boxattn_oracle as oc
from oracle import torch_fallback as tf

F64_TOL = dict(rtol=1e-11, atol=1e-12)
F32_TOL = dict(rtol=2e-4, atol=2e-5)     # float flavour of the oracle vs fp64 goldens


def _mask_native(g, key):
    """goldens store mask as (B,Lq,ms,ms,HC); native layout is (B,Lq,P,HC)."""
    m = g[key]
    return m.reshape(m.shape[0], m.shape[1], -1, m.shape[-1])


@pytest.mark.parametrize("name", golden_io.BOX)
def test_c_oracle_box_f64(name):
    g = golden_io.load(name)
    out = oc.box_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    np.testing.assert_allclose(out, g["out"], **F64_TOL)
    gv, gl, ga = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                      g["grad_out"])
    np.testing.assert_allclose(gv, g["grad_value"], **F64_TOL)
    np.testing.assert_allclose(gl, g["grad_loc"], **F64_TOL)
    np.testing.assert_allclose(ga, g["grad_attn"], **F64_TOL)


@pytest.mark.parametrize("name", golden_io.INST)
def test_c_oracle_instance_f64(name):
    g = golden_io.load(name)
    out, mask = oc.instance_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"],
                                         g["spatial_w"], g["level_w"])
    np.testing.assert_allclose(out, g["out"], **F64_TOL)
    np.testing.assert_allclose(mask, _mask_native(g, "mask_out"), **F64_TOL)
    gv, gl, gs, glw = oc.instance_attn_backward(
        g["value"], g["shapes"], g["lsi"], g["loc"], g["spatial_w"], g["level_w"],
        g["grad_out"], _mask_native(g, "grad_mask"))
    np.testing.assert_allclose(gv, g["grad_value"], **F64_TOL)
    np.testing.assert_allclose(gl, g["grad_loc"], **F64_TOL)
    np.testing.assert_allclose(gs, g["grad_spatial"], **F64_TOL)
    np.testing.assert_allclose(glw, g["grad_level"], **F64_TOL)


@pytest.mark.parametrize("name", ["G1_box_reftest", "G3_box_C32", "G6_box_ml"])
def test_c_oracle_box_f32(name):
    g = golden_io.load(name)
    f = lambda k: g[k].astype(np.float32)
    out = oc.box_attn_forward(f("value"), g["shapes"], g["lsi"], f("loc"), f("attn"))
    assert out.dtype == np.float32
    scale = max(1.0, np.abs(g["out"]).max())
    np.testing.assert_allclose(out / scale, g["out"] / scale, **F32_TOL)
    gv, gl, ga = oc.box_attn_backward(f("value"), g["shapes"], g["lsi"], f("loc"), f("attn"),
                                      f("grad_out"))
    for got, key in ((gv, "grad_value"), (gl, "grad_loc"), (ga, "grad_attn")):
        s = max(1.0, np.abs(g[key]).max())
        np.testing.assert_allclose(got / s, g[key] / s, **F32_TOL)


@pytest.mark.parametrize("name", ["G2_inst_reftest", "G6_inst_ms4"])
def test_c_oracle_instance_f32(name):
    g = golden_io.load(name)
    f = lambda k: g[k].astype(np.float32)
    out, mask = oc.instance_attn_forward(f("value"), g["shapes"], g["lsi"], f("loc"),
                                         f("spatial_w"), f("level_w"))
    for got, want in ((out, g["out"]), (mask, _mask_native(g, "mask_out"))):
        s = max(1.0, np.abs(want).max())
        np.testing.assert_allclose(got / s, want / s, **F32_TOL)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", golden_io.BOX)
def test_torch_fallback_box(name):
    g = golden_io.load(name)
    v = _t(g["value"]).requires_grad_(True)
    l = _t(g["loc"]).requires_grad_(True)
    a = _t(g["attn"]).requires_grad_(True)
    out = tf.box_attn(v, g["shapes"], l, a)
    out.backward(_t(g["grad_out"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **F64_TOL)
    np.testing.assert_allclose(v.grad.numpy(), g["grad_value"], **F64_TOL)
    np.testing.assert_allclose(l.grad.numpy(), g["grad_loc"], **F64_TOL)
    np.testing.assert_allclose(a.grad.numpy(), g["grad_attn"], **F64_TOL)


@pytest.mark.parametrize("name", golden_io.INST)
def test_torch_fallback_instance(name):
    g = golden_io.load(name)
    v = _t(g["value"]).requires_grad_(True)
    l = _t(g["loc"]).requires_grad_(True)
    s = _t(g["spatial_w"]).requires_grad_(True)
    w = _t(g["level_w"]).requires_grad_(True)
    out, mask = tf.instance_attn(v, g["shapes"], l, s, w)
    gm = _t(_mask_native(g, "grad_mask"))
    ((out * _t(g["grad_out"])).sum() + (mask * gm).sum()).backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **F64_TOL)
    np.testing.assert_allclose(mask.detach().numpy(), _mask_native(g, "mask_out"), **F64_TOL)
    np.testing.assert_allclose(v.grad.numpy(), g["grad_value"], **F64_TOL)
    np.testing.assert_allclose(l.grad.numpy(), g["grad_loc"], **F64_TOL)
    np.testing.assert_allclose(s.grad.numpy(), g["grad_spatial"], **F64_TOL)
    np.testing.assert_allclose(w.grad.numpy(), g["grad_level"], **F64_TOL)


def test_oracle_zero_sized_and_all_outside():
    """Ragged / degenerate inputs: no queries, and every point outside the map."""
    shapes = np.array([[3, 2]], dtype=np.int64)
    lsi = np.array([0], dtype=np.int64)
    value = np.random.default_rng(0).standard_normal((1, 6, 2, 3))
    loc = np.full((1, 4, 2, 1, 2, 2), 7.5)
    attn = np.full((1, 4, 2, 1, 2), 0.5)
    out = oc.box_attn_forward(value, shapes, lsi, loc, attn)
    assert out.shape == (1, 4, 6) and not out.any()
    gv, gl, ga = oc.box_attn_backward(value, shapes, lsi, loc, attn, np.ones_like(out))
    assert not gv.any() and not gl.any() and not ga.any()
    out0 = oc.box_attn_forward(value, shapes, lsi, loc[:, :0], attn[:, :0])
    assert out0.shape == (1, 0, 6)


def test_oracle_window_test_at_exactly_minus_one():
    """A pixel coordinate of exactly -1 is skipped by the reference kernel's window test
    (box_attn_kernel.cuh:325-328: `h_im > -1 && w_im > -1 && h_im < H && w_im < W`), so all
    gradients of that point are zero.  (The reference's grid_sample test oracle yields a
    non-zero grad_loc there; the CUDA kernel is the behaviour the op follows.)"""
    shapes = np.array([[4, 4]], dtype=np.int64)
    lsi = np.array([0], dtype=np.int64)
    value = np.random.default_rng(1).standard_normal((1, 16, 1, 2))
    loc = np.array([-0.5 / 4, 0.5]).reshape(1, 1, 1, 1, 1, 2)       # w_im = -1 exactly
    attn = np.ones((1, 1, 1, 1, 1))
    out = oc.box_attn_forward(value, shapes, lsi, loc, attn)
    gv, gl, ga = oc.box_attn_backward(value, shapes, lsi, loc, attn, np.ones_like(out))
    assert not out.any() and not gv.any() and not gl.any() and not ga.any()
    loc[..., 0] = 1 + 0.5 / 4                                        # w_im = W exactly
    gv, gl, ga = oc.box_attn_backward(value, shapes, lsi, loc, attn, np.ones_like(out))
    assert not gv.any() and not gl.any() and not ga.any()


def test_g9_fixtures_are_complete():
    """G9 (tests/golden/make_goldens.py g9): the reference's modules at d = 256 / 8 heads / 4 levels with gradients.  The
    GPU tests compare every opt-in path of boxer_amd.modules with them; here only that each fixture holds what those
    tests read: a state dict, inputs, an upstream gradient per output, and a gradient for every input and parameter."""
    import golden_io
    names = golden_io.names("G9_")
    assert len(names) == 6, names
    for name in names:
        g = golden_io.load(name)
        params = [k[3:] for k in g if k.startswith("sd.") and k[3:] != "kernel_indices"]
        assert params and all("grad." + k in g and g["grad." + k].shape == g["sd." + k].shape for k in params), name
        for k in ("query", "value", "ref_windows"):
            assert g["grad_" + k].shape == g[k].shape, (name, k)
        n_out = 2 if "inst" in name else 1
        assert all(g["out%d" % i].shape == g["gout%d" % i].shape for i in range(n_out)), name
        assert g["value"].shape[-1] == 256 and g["shapes"].shape == (4, 2)
        for k in g:                                   # finite numbers only
            if g[k].dtype.kind == "f":
                assert np.isfinite(g[k]).all(), (name, k)
