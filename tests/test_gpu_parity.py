"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP operator, called through the
C ABI (boxer_amd.ops -> libboxattn_hip.so), against

* the committed golden vectors of the reference's own oracle (tests/golden), and
* the CPU oracle (oracle/boxattn_oracle.c) on seeded inputs,

Tolerances (BASELINE.json north_star): fp64 1e-10, fp32 1e-4, bf16 1e-2 -- absolute on
O(1) data, scaled by the magnitude of the expected tensor when that exceeds 1.
"""
import math

import numpy as np
import pytest
import torch

import golden_io
from oracle import boxattn_oracle as oc

pytestmark = pytest.mark.gpu

TOL = {torch.float64: 1e-10, torch.float32: 1e-4, torch.bfloat16: 1e-2}
VARIANTS = {"auto": 0, "generic": 1, "atomic": 2, "binned": 3}


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def close(got, want, dtype, what, ignore=None):
    """``ignore``: boolean mask (broadcastable to ``want``) of entries excluded from the check."""
    got = got.detach().double().cpu().numpy()
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if ignore is not None:
        keep = ~np.broadcast_to(ignore, want.shape)
        got, want = got * keep, want * keep
    scale = max(1.0, float(np.abs(want).max()) if want.size else 1.0)
    err = float(np.abs(got - want).max()) / scale if want.size else 0.0
    assert err <= TOL[dtype], "%s: max scaled err %.3e > %.1e" % (what, err, TOL[dtype])


@pytest.fixture(autouse=True)
def _reset_variant():
    from boxer_amd import _lib
    yield
    _lib.set_variant(0)


def _cdt(dtype):
    return torch.float32 if dtype == torch.bfloat16 else dtype


def run_box(g, dtype, variant):
    from boxer_amd import _lib, ops
    _lib.set_variant(VARIANTS[variant])
    cdt = _cdt(dtype)
    value, loc, attn = dev(g["value"], dtype), dev(g["loc"], cdt), dev(g["attn"], cdt)
    shapes, lsi = dev(g["shapes"]), dev(g["lsi"])
    gout = dev(g["grad_out"], dtype)
    out = ops.box_attn_forward(value, shapes, lsi, loc, attn, 64)
    gv, gl, ga = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64)
    torch.cuda.synchronize()
    return out, gv, gl, ga


def run_inst(g, dtype, variant):
    from boxer_amd import _lib, ops
    _lib.set_variant(VARIANTS[variant])
    cdt = _cdt(dtype)
    value, loc = dev(g["value"], dtype), dev(g["loc"], cdt)
    sw, lw = dev(g["spatial_w"], cdt), dev(g["level_w"], cdt)
    shapes, lsi = dev(g["shapes"]), dev(g["lsi"])
    gout = dev(g["grad_out"], dtype)
    gm = g["grad_mask"]
    gmask = dev(gm.reshape(gm.shape[0], gm.shape[1], -1, gm.shape[-1]), dtype)
    out, mask = ops.instance_attn_forward(value, shapes, lsi, loc, sw, lw, 64)
    gv, gl, gs, glw = ops.instance_attn_backward(value, shapes, lsi, loc, sw, lw, gout, gmask, 64)
    torch.cuda.synchronize()
    return out, mask, gv, gl, gs, glw


# ------------------------------------------------------------------ goldens, fp64 / fp32
@pytest.mark.parametrize("variant", ["auto", "generic"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", golden_io.BOX)
def test_box_golden(name, dtype, variant):
    g = golden_io.load(name)
    out, gv, gl, ga = run_box(g, dtype, variant)
    close(out, g["out"], dtype, "out")
    close(gv, g["grad_value"], dtype, "grad_value")
    close(gl, g["grad_loc"], dtype, "grad_loc")
    close(ga, g["grad_attn"], dtype, "grad_attn")


@pytest.mark.parametrize("variant", ["auto", "generic"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", golden_io.INST)
def test_instance_golden(name, dtype, variant):
    g = golden_io.load(name)
    out, mask, gv, gl, gs, glw = run_inst(g, dtype, variant)
    m = g["mask_out"]
    close(out, g["out"], dtype, "out")
    close(mask, m.reshape(m.shape[0], m.shape[1], -1, m.shape[-1]), dtype, "mask_out")
    close(gv, g["grad_value"], dtype, "grad_value")
    close(gl, g["grad_loc"], dtype, "grad_loc")
    close(gs, g["grad_spatial"], dtype, "grad_spatial")
    close(glw, g["grad_level"], dtype, "grad_level")


# ------------------------------------------------------------------ goldens, bf16 storage
# q8 fixtures: value and upstream grads are exactly representable in bf16, so the only bf16
# error is the rounding of the bf16 outputs (out, mask_out, grad_value).
@pytest.mark.parametrize("variant", ["auto", "generic"])
@pytest.mark.parametrize("name", ["G5_box_C1", "G6_box_ml"])
def test_box_golden_bf16(name, variant):
    g = golden_io.load(name)
    out, gv, gl, ga = run_box(g, torch.bfloat16, variant)
    assert out.dtype == torch.bfloat16 and gv.dtype == torch.bfloat16
    assert gl.dtype == torch.float32 and ga.dtype == torch.float32
    close(out, g["out"], torch.bfloat16, "out")
    close(gv, g["grad_value"], torch.bfloat16, "grad_value")
    close(gl, g["grad_loc"], torch.float32, "grad_loc")      # fp32 math on exact inputs
    close(ga, g["grad_attn"], torch.float32, "grad_attn")


@pytest.mark.parametrize("variant", ["auto", "generic"])
@pytest.mark.parametrize("name", ["G6_inst_ms4", "G6_inst_ms14"])
def test_instance_golden_bf16(name, variant):
    g = golden_io.load(name)
    out, mask, gv, gl, gs, glw = run_inst(g, torch.bfloat16, variant)
    m = g["mask_out"]
    close(out, g["out"], torch.bfloat16, "out")
    close(mask, m.reshape(m.shape[0], m.shape[1], -1, m.shape[-1]), torch.bfloat16, "mask_out")
    close(gv, g["grad_value"], torch.bfloat16, "grad_value")
    close(gl, g["grad_loc"], torch.float32, "grad_loc")
    close(gs, g["grad_spatial"], torch.float32, "grad_spatial")
    close(glw, g["grad_level"], torch.float32, "grad_level")


# ------------------------------------------------------------------ seeded, vs the C oracle
def _seeded(shapes, B, H, C, Lq, P, seed, lo=-0.1, hi=1.1):
    rng = np.random.default_rng(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    sizes = shapes.prod(1)
    lsi = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    S, L = int(sizes.sum()), len(shapes)
    value = rng.integers(-127, 128, (B, S, H, C)).astype(np.float64) / 64
    loc = rng.uniform(lo, hi, (B, Lq, H, L, P, 2)).astype(np.float32).astype(np.float64)
    a = rng.uniform(1e-5, 1, (B, Lq, H, L, P))
    attn = (a / a.sum((-1, -2), keepdims=True)).astype(np.float32).astype(np.float64)
    lvl = (a / a.sum(-2, keepdims=True)).astype(np.float32).astype(np.float64)
    gout = rng.integers(-64, 65, (B, Lq, H * C)).astype(np.float64) / 32
    gmask = rng.integers(-64, 65, (B, Lq, P, H * C)).astype(np.float64) / 32
    return dict(value=value, shapes=shapes, lsi=lsi, loc=loc, attn=attn, spatial_w=attn,
                level_w=lvl, grad_out=gout, grad_mask=gmask, on_edge=on_cell_edge(loc, shapes))


def on_cell_edge(loc, shapes, eps=1e-4):
    """(B,Lq,H,L,P,1) mask of points whose pixel coordinate is within eps of an integer.
    grad_loc is discontinuous there (the bilinear cell changes), so float32 arithmetic and the
    float64 oracle may legitimately pick different cells; such points are excluded from the
    grad_loc comparison (their other outputs are continuous and stay checked)."""
    size = np.asarray(shapes, dtype=np.float64)[None, None, None, :, None, ::-1]   # (W, H)
    pix = loc * size - 0.5
    return (np.abs(pix - np.round(pix)) < eps).any(-1, keepdims=True)


SEEDED = [
    # shapes, B, H, C, Lq, P
    ([(20, 30), (10, 15), (5, 8), (3, 4)], 2, 8, 32, 333, 4),     # BoxeR encoder geometry
    ([(20, 30), (10, 15)], 1, 8, 32, 7, 16),                      # ragged tail wave
    ([(9, 7)], 3, 4, 16, 50, 4),                                  # G=4 path
    ([(9, 7), (4, 3)], 2, 2, 64, 21, 9),                          # G=16 path, odd P
    ([(9, 7), (4, 3)], 2, 3, 12, 5, 4),                           # C%4==0 but no fast variant
    ([(6, 5)], 1, 1, 1, 3, 1),                                    # minimum everything
    # odd map sizes (uneven destination blocks), one-pixel-wide / -high levels, enough points
    # per block for chunked work items
    ([(25, 25), (13, 13), (7, 5), (1, 3), (2, 1)], 1, 2, 32, 700, 4),
    ([(13, 21), (5, 4)], 2, 2, 64, 300, 9),                       # 8 channels per lane, G=8 (bf16)
    # few queries x many points: one wave per (query, head) pair in the instance forward;
    # 3 and 5 levels do not fill the lane groups evenly
    ([(11, 9), (6, 5), (3, 2)], 1, 4, 32, 6, 49),
    ([(11, 9), (6, 5), (3, 2), (2, 2), (1, 1)], 2, 2, 32, 3, 36),
    # ... two waves per pair (head = workgroup mod 8), an odd number of (image, query) rows: the last workgroup's second
    # pair does not exist
    ([(11, 9), (6, 5)], 1, 8, 32, 7, 40),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", SEEDED, ids=[str(i) for i in range(len(SEEDED))])
def test_box_vs_oracle(cfg, dtype):
    g = _seeded(*cfg, seed=11)
    want_out = oc.box_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                g["grad_out"])
    out, gv, gl, ga = run_box(g, dtype, "auto")
    close(out, want_out, dtype, "out")
    close(gv, want[0], dtype, "grad_value")
    close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
    close(ga, want[2], torch.float32, "grad_attn")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", SEEDED, ids=[str(i) for i in range(len(SEEDED))])
def test_instance_vs_oracle(cfg, dtype):
    g = _seeded(*cfg, seed=12)
    want_out, want_mask = oc.instance_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"],
                                                   g["spatial_w"], g["level_w"])
    want = oc.instance_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"],
                                     g["spatial_w"], g["level_w"], g["grad_out"],
                                     g["grad_mask"])
    out, mask, gv, gl, gs, glw = run_inst(g, dtype, "auto")
    close(out, want_out, dtype, "out")
    close(mask, want_mask, dtype, "mask_out")
    close(gv, want[0], dtype, "grad_value")
    close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
    close(gs, want[2], torch.float32, "grad_spatial")
    close(glw, want[3], torch.float32, "grad_level")


# every backward algorithm of the fast family on the BoxeR geometry (C = 16 / 32 / 64):
# "atomic" = hardware fp atomics per contribution, "binned" = destination-binned, no atomics
FAST_CFGS = [SEEDED[0], SEEDED[1], SEEDED[2], SEEDED[3],
             ([(37, 53), (19, 27), (10, 14), (5, 7)], 2, 8, 32, 700, 4),   # several blocks / level
             ([(16, 24)], 1, 8, 32, 3000, 4)]                              # chunked heavy bins


@pytest.mark.parametrize("variant", ["atomic", "binned"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", FAST_CFGS, ids=[str(i) for i in range(len(FAST_CFGS))])
def test_box_backward_algorithms(cfg, dtype, variant):
    g = _seeded(*cfg, seed=21, lo=-0.2, hi=1.2)
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                g["grad_out"])
    out, gv, gl, ga = run_box(g, dtype, variant)
    close(gv, want[0], dtype, "grad_value")
    close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
    close(ga, want[2], torch.float32, "grad_attn")


OPT_ACC_F32 = 19        # boxattn_set_option: float32 accumulate -- 0 default: bf16 matrix cores on exact three-term splits,
                        # 1: VALU list walk, 2: v_mfma_f32_32x32x2_f32


@pytest.mark.parametrize("with_plan", [False, True], ids=["own_binning", "forward_plan"])
@pytest.mark.parametrize("cfg", [FAST_CFGS[0], FAST_CFGS[1], FAST_CFGS[4], FAST_CFGS[5], SEEDED[6]],
                         ids=["0", "1", "4", "5", "6"])
def test_float32_matrix_core_accumulate(cfg, with_plan):
    """float32 grad_value (32 channels per head) from the matrix cores -- bf16 MFMAs on exact three-term splits (the
    default) and float32 MFMAs -- against the oracle at the float32 tolerance and against the VALU kernel to
    summation-order rounding."""
    from boxer_amd import _lib, ops
    g = _seeded(*cfg, seed=37, lo=-0.2, hi=1.2)
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"], g["grad_out"])
    value, loc, attn = dev(g["value"], torch.float32), dev(g["loc"], torch.float32), dev(g["attn"], torch.float32)
    shapes, lsi, gout = dev(g["shapes"]), dev(g["lsi"]), dev(g["grad_out"], torch.float32)
    res = {}
    for mode in (0, 2, 1):
        old = _lib.load().boxattn_set_option(OPT_ACC_F32, mode)
        try:
            if with_plan:
                _, plan = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
                gv = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan)[0]
            else:
                gv = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64)[0]
            torch.cuda.synchronize()
        finally:
            _lib.load().boxattn_set_option(OPT_ACC_F32, old)
        res[mode] = gv
        close(gv, want[0], torch.float32, "grad_value (float32 accumulate %d)" % mode)
    assert (res[2] - res[1]).abs().max().item() <= 1e-5 * max(1.0, res[1].abs().max().item())
    assert (res[0] - res[1]).abs().max().item() <= 1e-5 * max(1.0, res[1].abs().max().item())


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["split_bf16_mfma", "valu", "f32_mfma"])
def test_float32_accumulate_single_terms(mode):
    """One point per (query, head) on a map with one query: every grad_value element is ONE product (hh hw a) g.  The
    default float32 accumulate runs on the bf16 matrix cores with rows and weights split into three bf16 terms each
    (exact splits; the six partial products above 2^-24 of the product): its result must be the float32 product to
    within 2 ulp -- float32 arithmetic, not a 16-bit approximation of it (a two-term split of the weights alone
    would be off by 2^-17, i.e. 64 ulp)."""
    from boxer_amd import _lib
    rng = np.random.default_rng(4321)
    C = 32
    # a power-of-two map and locations on a 2^-10 grid: pixel coordinates, bilinear fractions and their products are
    # exact in float32 (the kernels form them in float32, the oracle in float64 -- on a general map the two differ by
    # the rounding of loc * size - 0.5, hundreds of ulp of a small fraction, whatever the accumulate does)
    shapes = np.asarray([(8, 16)], dtype=np.int64)
    B, H, Lq, P = 2, 8, 1, 1
    S = int(shapes.prod(1).sum())
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    g = dict(shapes=shapes, lsi=np.zeros(1, dtype=np.int64),
             value=f32(rng.standard_normal((B, S, H, C))),
             loc=rng.integers(128, 896, (B, Lq, H, 1, P, 2)).astype(np.float64) / 1024 + 1.0 / 4096,
             attn=f32(rng.uniform(0.2, 1.0, (B, Lq, H, 1, P))),
             grad_out=f32(rng.standard_normal((B, Lq, H * C)) * rng.choice([1e-3, 1.0, 1e3], (B, Lq, H * C))))
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"], g["grad_out"])[0]
    old = _lib.load().boxattn_set_option(OPT_ACC_F32, mode)
    try:
        _, gv, _, _ = run_box(g, torch.float32, "binned")
    finally:
        _lib.load().boxattn_set_option(OPT_ACC_F32, old)
    got = gv.double().cpu().numpy()
    nz = want != 0
    assert nz.sum() == B * H * 4 * C and (got[~nz] == 0).all()
    ulp = 2.0 ** (np.floor(np.log2(np.abs(want[nz]))) - 23)
    worst = (np.abs(got[nz] - want[nz]) / ulp).max()
    # (weight = (hh hw) a: one float32 rounding in the kernels, none in the oracle; then the product with g)
    assert worst <= 3.0, "worst error %.2f ulp" % worst


@pytest.mark.parametrize("variant", ["atomic", "binned"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", FAST_CFGS[:5], ids=[str(i) for i in range(5)])
def test_instance_backward_algorithms(cfg, dtype, variant):
    g = _seeded(*cfg, seed=22, lo=-0.2, hi=1.2)
    want = oc.instance_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"],
                                     g["spatial_w"], g["level_w"], g["grad_out"],
                                     g["grad_mask"])
    out, mask, gv, gl, gs, glw = run_inst(g, dtype, variant)
    close(gv, want[0], dtype, "grad_value")
    close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
    close(gs, want[2], torch.float32, "grad_spatial")
    close(glw, want[3], torch.float32, "grad_level")


# (the library takes the matrix-core kernel for instance attention from 65 536 points per (image, head) slice up)
INST_SPLIT_CFGS = [([(20, 30), (10, 15)], 1, 8, 32, 300, 112),                       # 14 x 8 points, two levels
                   ([(25, 25), (13, 13), (7, 5), (1, 3), (2, 1)], 1, 2, 32, 2000, 8),    # uneven blocks, one-pixel levels
                   ([(16, 24)], 2, 8, 32, 1100, 64),                                 # heavy bins: chunked work items
                   ([(20, 30), (10, 15)], 2, 8, 32, 700, 49),                        # odd P: one point per fill thread and group
                   FAST_CFGS[0]]                                                     # (below the bound: the VALU kernel)


@pytest.mark.parametrize("cfg", INST_SPLIT_CFGS, ids=["two_levels", "uneven", "heavy", "odd_p", "small"])
def test_float32_instance_matrix_core_accumulate(cfg):
    """Instance attention, float32, 32 channels per head: grad_value from the bf16 matrix cores -- the two upstream rows of
    a record (a_s grad_out + a_l grad_mask, instance_attn_kernel.cuh:139) combined in float32 and split into three exact
    bf16 terms -- against the oracle at the float32 tolerance and against the VALU kernel to summation-order rounding."""
    from boxer_amd import _lib
    g = _seeded(*cfg, seed=38, lo=-0.2, hi=1.2)
    want = oc.instance_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["spatial_w"], g["level_w"],
                                     g["grad_out"], g["grad_mask"])
    res = {}
    for mode in (0, 1):
        old = _lib.load().boxattn_set_option(OPT_ACC_F32, mode)
        try:
            _, _, gv, gl, gs, glw = run_inst(g, torch.float32, "binned")
        finally:
            _lib.load().boxattn_set_option(OPT_ACC_F32, old)
        res[mode] = gv
        close(gv, want[0], torch.float32, "grad_value (float32 accumulate %d)" % mode)
        close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
        close(gs, want[2], torch.float32, "grad_spatial")
        close(glw, want[3], torch.float32, "grad_level")
    assert (res[0] - res[1]).abs().max().item() <= 1e-5 * max(1.0, res[1].abs().max().item())


@pytest.mark.parametrize("chunk", [64, 192, 1024, 4096])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_binned_backward_records_per_item(dtype, chunk):
    """boxattn_set_option(10): the records per work item only change how a bin is cut into work
    items (and how many partial tiles the combine pass sums), never the result -- heavy bins
    (3 000 queries on a 16 x 24 map) and a multi-level map with a one-block level."""
    from boxer_amd import _lib
    for cfg in (FAST_CFGS[-1], ([(25, 25), (13, 13), (7, 5), (1, 3), (2, 1)], 1, 2, 32, 700, 4)):
        g = _seeded(*cfg, seed=31, lo=-0.2, hi=1.2)
        want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                    g["grad_out"])
        old = _lib.set_option("bin_chunk", chunk)
        try:
            out, gv, gl, ga = run_box(g, dtype, "binned")
        finally:
            _lib.set_option("bin_chunk", old)
        close(gv, want[0], dtype, "grad_value")
        close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
        close(ga, want[2], torch.float32, "grad_attn")


@pytest.mark.parametrize("levels", [[(128, 256)],               # 32 x 32 = 1 024 blocks: still one workgroup
                                    [(128, 264)],               # 1 056: just over
                                    [(128, 512)],               # 2 048: exactly two segments
                                    [(200, 180), (20, 18)]],    # 1 125 + 15, two levels
                         ids=["1024", "1056", "2048", "2lv"])
def test_block_scan_over_several_workgroups(levels):
    """More than 1 024 blocks per (image, head) slice: the block scan runs as bin_scan_seg_kernel +
    bin_scan_emit_kernel.  150 queries: a SPARSE map (BinPlan::min_items == 0) -- the blocks without records get
    no work item, zero workers in the accumulate launch store their zeros (into poisoned memory here);
    3 000 queries: every block has its item."""
    for n_queries in (150, 3000):
        g = _seeded(levels, 1, 2, 32, n_queries, 4, seed=32)
        want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                    g["grad_out"])
        for dtype in (torch.float32, torch.bfloat16):
            junk = [torch.full(g["value"].shape, float("nan"), device="cuda", dtype=dtype) for _ in range(3)]
            del junk                                   # poison what the allocator hands out next
            out, gv, gl, ga = run_box(g, dtype, "binned")
            assert torch.isfinite(gv.float()).all()
            close(gv, want[0], dtype, "grad_value")
            close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
            close(ga, want[2], torch.float32, "grad_attn")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_instance_backward_on_a_sparse_map(dtype):
    """Instance attention on a map of more than 1 024 blocks with few queries (the mask decoder on a BEV-sized
    map): a sparse plan -- no work items for the blocks without records, their zeros come from the zero workers
    of the (VALU) accumulate kernel; into poisoned memory."""
    g = _seeded([(136, 264), (17, 33)], 1, 2, 32, 40, 4, seed=33)
    want = oc.instance_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["spatial_w"], g["level_w"],
                                     g["grad_out"], g["grad_mask"])
    junk = [torch.full(g["value"].shape, float("nan"), device="cuda", dtype=dtype) for _ in range(3)]
    del junk
    out, mask, gv, gl, gs, glw = run_inst(g, dtype, "binned")
    assert torch.isfinite(gv.float()).all()
    close(gv, want[0], dtype, "grad_value")
    close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
    close(gs, want[2], torch.float32, "grad_spatial")
    close(glw, want[3], torch.float32, "grad_level")


def test_binned_backward_clustered_points():
    """All sample points of a head on one pixel: one bin gets every record (many chunks),
    all other bins are empty."""
    g = _seeded([(12, 20), (6, 10)], 2, 8, 32, 500, 4, seed=23)
    g["loc"][...] = 0.5 + (g["loc"] - 0.5) * 1e-3
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                g["grad_out"])
    out, gv, gl, ga = run_box(g, torch.float32, "binned")
    close(gv, want[0], torch.float32, "grad_value")
    close(gl, want[1], torch.float32, "grad_loc", ignore=g["on_edge"])
    close(ga, want[2], torch.float32, "grad_attn")


# ------------------------------------------------------------------ edge cases
def test_empty_queries_and_all_outside():
    from boxer_amd import ops
    g = _seeded([(6, 5)], 2, 8, 32, 4, 4, seed=3)
    value, shapes, lsi = dev(g["value"], torch.float32), dev(g["shapes"]), dev(g["lsi"])
    loc = torch.full((2, 4, 8, 1, 4, 2), 5.0, device="cuda")
    attn = dev(g["attn"], torch.float32)
    out = ops.box_attn_forward(value, shapes, lsi, loc, attn, 64)
    gv, gl, ga = ops.box_attn_backward(value, shapes, lsi, loc, attn, torch.ones_like(out), 64)
    assert not out.any() and not gv.any() and not gl.any() and not ga.any()
    out0 = ops.box_attn_forward(value, shapes, lsi, loc[:, :0].contiguous(),
                                attn[:, :0].contiguous(), 64)
    assert out0.shape == (2, 0, 256)
    gv0, gl0, ga0 = ops.box_attn_backward(value, shapes, lsi, loc[:, :0].contiguous(),
                                          attn[:, :0].contiguous(), out0, 64)
    assert gv0.shape == value.shape and not gv0.any() and gl0.numel() == 0


def test_outputs_do_not_depend_on_previous_buffer_contents():
    """Outputs are fully defined by the call (no reliance on pre-zeroed memory)."""
    from boxer_amd import ops
    g = _seeded([(8, 8), (4, 4)], 1, 8, 32, 16, 4, seed=5, lo=-0.5, hi=1.5)
    args = [dev(g["value"], torch.float32), dev(g["shapes"]), dev(g["lsi"]),
            dev(g["loc"], torch.float32), dev(g["attn"], torch.float32)]
    gout = dev(g["grad_out"], torch.float32)
    first = ops.box_attn_backward(*args, gout, 64)
    junk = [torch.full_like(t, float("nan")) for t in first]      # poison the allocator cache
    del junk
    second = ops.box_attn_backward(*args, gout, 64)
    for a, b in zip(first, second):
        assert torch.isfinite(b).all()
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


def test_error_behaviour_matches_reference():
    from boxer_amd import ops
    g = _seeded([(6, 5)], 2, 2, 4, 3, 2, seed=1)
    value, shapes, lsi = dev(g["value"], torch.float32), dev(g["shapes"]), dev(g["lsi"])
    loc, attn = dev(g["loc"], torch.float32), dev(g["attn"], torch.float32)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        ops.box_attn_forward(value.transpose(2, 3), shapes, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.box_attn_forward(value.cpu(), shapes, lsi, loc, attn, 64)
    with pytest.raises(AssertionError, match="must divide"):
        ops.box_attn_forward(torch.cat([value, value[:1]]), shapes, lsi,
                             torch.cat([loc, loc[:1]]), torch.cat([attn, attn[:1]]), 2)
    # im2col_step is numerically irrelevant (pure batch chunking in the reference)
    a = ops.box_attn_forward(value, shapes, lsi, loc, attn, 1)
    b = ops.box_attn_forward(value, shapes, lsi, loc, attn, 64)
    assert torch.equal(a, b)


# ------------------------------------------------------------------ autograd Functions
def test_function_gradcheck_fp64():
    """The reference's check_gradient_numerical (tests/box_attn_test.py:162-189)."""
    from torch.autograd import gradcheck
    from boxer_amd import BoxAttnFunction, InstanceAttnFunction
    shapes = torch.tensor([(6, 4), (3, 2)], dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for C in (30, 32, 64, 71):
        value = (torch.rand(1, S, 2, C, device="cuda", dtype=torch.double) * 0.01).requires_grad_()
        loc = torch.rand(1, 2, 2, 2, 2, 2, device="cuda", dtype=torch.double).requires_grad_()
        attn = torch.rand(1, 2, 2, 2, 2, device="cuda", dtype=torch.double) + 1e-5
        attn = (attn / attn.sum((-1, -2), keepdim=True)).requires_grad_()
        assert gradcheck(BoxAttnFunction.apply, (value, shapes, lsi, loc, attn, 2))
    value = (torch.rand(1, S, 2, 8, device="cuda", dtype=torch.double) * 0.01).requires_grad_()
    loc = torch.rand(1, 2, 2, 2, 4, 2, device="cuda", dtype=torch.double).requires_grad_()
    a = torch.rand(1, 2, 2, 2, 2, 2, device="cuda", dtype=torch.double) + 1e-5
    sw = (a / a.sum((-1, -2, -3), keepdim=True)).requires_grad_()
    lw = (a / a.sum(-3, keepdim=True)).requires_grad_()
    assert gradcheck(InstanceAttnFunction.apply, (value, shapes, lsi, loc, sw, lw, 2, 2))


def test_function_autocast_contract():
    """Under autocast the parity Functions compute in fp32 and return fp32
    (custom_fwd(cast_inputs=float32), box_attention_func.py:11)."""
    from boxer_amd import BoxAttnBF16Function, BoxAttnFunction
    g = _seeded([(8, 8), (4, 4)], 2, 8, 32, 10, 4, seed=9)
    shapes, lsi = dev(g["shapes"]), dev(g["lsi"])
    value = dev(g["value"], torch.float32).requires_grad_()
    loc = dev(g["loc"], torch.float32).requires_grad_()
    attn = dev(g["attn"], torch.float32).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = BoxAttnFunction.apply(value.bfloat16(), shapes, lsi, loc, attn, 64)
    assert out.dtype == torch.float32
    out.sum().backward()
    assert value.grad.dtype == torch.float32 and loc.grad is not None and attn.grad is not None
    want = oc.box_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    close(out, want, torch.float32, "autocast out")
    out_bf = BoxAttnBF16Function.apply(value, shapes, lsi, loc, attn, 64)
    assert out_bf.dtype == torch.bfloat16
    out_bf.float().sum().backward()
    close(out_bf, want, torch.bfloat16, "bf16 out")


# ------------------------------------------------------------------ nn.Modules (G7)
def _load_module(cls, g, **kw):
    m = cls(**kw).double().cuda()
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    m.load_state_dict(sd, strict=True)
    return m


def test_modules_match_reference_goldens():
    from boxer_amd import Box3dAttention, BoxAttention, InstanceAttention
    d, nl, nh = 32, 2, 4

    def args(g, with_mask=True, with_ratio=True):
        return (dev(g["query"]), dev(g["value"]), dev(g["shapes"]),
                dev(g["v_mask"]) if with_mask else None, dev(g["lsi"]),
                dev(g["ratios"]) if with_ratio else None, dev(g["ref_windows"]))

    g = golden_io.load("G7_module_box")
    m = _load_module(BoxAttention, g, d_model=d, num_level=nl, num_head=nh, kernel_size=2)
    out, attn = m(*args(g))
    close(out, g["out"], torch.float64, "BoxAttention out")
    close(attn, g["attn"], torch.float64, "BoxAttention attn")

    g = golden_io.load("G7_module_box_perhead")
    out, attn = m(*args(g, False, False))
    close(out, g["out"], torch.float64, "BoxAttention per-head out")

    for ks in (4, 14):
        g = golden_io.load("G7_module_inst_k%d" % ks)
        m = _load_module(InstanceAttention, g, d_model=d, num_level=nl, num_head=nh,
                         kernel_size=ks)
        m.inferencing = False
        out, mask_out, (sw, lw) = m(*args(g))
        close(out, g["out"], torch.float64, "InstanceAttention out")
        close(mask_out, g["mask_out"], torch.float64, "InstanceAttention mask_out")
        close(sw, g["spatial_w"], torch.float64, "spatial_w")
        close(lw, g["level_w"], torch.float64, "level_w")
        m.inferencing = True
        out, none_mask, _ = m(*args(g))
        assert none_mask is None
        close(out, g["out_inferencing"], torch.float64, "InstanceAttention inferencing out")

    g = golden_io.load("G7_module_box3d_rot")
    m = _load_module(Box3dAttention, g, d_model=d, num_level=nl, num_head=nh,
                     with_rotation=True, kernel_size=2)
    out, _ = m(*args(g))
    close(out, g["out"], torch.float64, "Box3dAttention(rot) out")

    g = golden_io.load("G7_module_box3d_fixed")
    m = _load_module(Box3dAttention, g, d_model=d, num_level=nl, num_head=nh,
                     with_rotation=False, kernel_size=3)
    out, _ = m(*args(g, False, False))
    close(out, g["out"], torch.float64, "Box3dAttention(fixed) out")


# The opt-in paths of SURVEY.md 8(f) N1 / N3 against the SAME reference goldens:
# fused_grid (grid kernels), fused_pointwise (one-pass
# softmax / mask-fill + cast), native bf16 storage.  They are float32 / bf16 paths, so the modules run in
# float32 here and are compared with the float64 goldens at the float32 (bf16) tolerance.
@pytest.mark.parametrize("native_bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("fused_pointwise", [False, True], ids=["torch_pointwise", "fused_pointwise"])
@pytest.mark.parametrize("fused_grid", [0, 1])
def test_fused_module_paths_match_reference_goldens(fused_grid, fused_pointwise, native_bf16):
    from boxer_amd import Box3dAttention, BoxAttention, InstanceAttention
    d, nl, nh = 32, 2, 4
    tol = 2e-2 if native_bf16 else 1e-4

    def f32(a):
        t = dev(a)
        return t.float() if t.is_floating_point() else t

    def args(g, with_mask=True, with_ratio=True):
        return (f32(g["query"]), f32(g["value"]), dev(g["shapes"]),
                dev(g["v_mask"]) if with_mask else None, dev(g["lsi"]),
                f32(g["ratios"]) if with_ratio else None, f32(g["ref_windows"]))

    def make(cls, g, **kw):
        m = _load_module(cls, g, **kw).float()
        m.fused_grid, m.fused_pointwise, m.native_bf16 = fused_grid, fused_pointwise, native_bf16
        return m

    def run(m, a):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=native_bf16):
            return m(*a)

    def check(got, want, what):
        got = got.detach().double().cpu().numpy()
        err = float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))
        assert err <= tol, "%s (fused_grid=%s pointwise=%s bf16=%s): %.3e > %.0e" % (
            what, fused_grid, fused_pointwise, native_bf16, err, tol)

    g = golden_io.load("G7_module_box")
    m = make(BoxAttention, g, d_model=d, num_level=nl, num_head=nh, kernel_size=2)
    out, attn = run(m, args(g))
    check(out, g["out"], "BoxAttention out")
    check(attn, g["attn"], "BoxAttention attn")
    g = golden_io.load("G7_module_box_perhead")
    check(run(m, args(g, False, False))[0], g["out"], "BoxAttention per-head out")
    for ks in (4, 14):
        g = golden_io.load("G7_module_inst_k%d" % ks)
        m = make(InstanceAttention, g, d_model=d, num_level=nl, num_head=nh, kernel_size=ks)
        m.inferencing = False
        out, mask_out, (sw, lw) = run(m, args(g))
        check(out, g["out"], "InstanceAttention out")
        check(mask_out, g["mask_out"], "InstanceAttention mask_out")
        check(sw, g["spatial_w"], "spatial_w")
        check(lw, g["level_w"], "level_w")
    g = golden_io.load("G7_module_box3d_rot")
    m = make(Box3dAttention, g, d_model=d, num_level=nl, num_head=nh, with_rotation=True, kernel_size=2)
    check(run(m, args(g))[0], g["out"], "Box3dAttention(rot) out")
    g = golden_io.load("G7_module_box3d_fixed")
    m = make(Box3dAttention, g, d_model=d, num_level=nl, num_head=nh, with_rotation=False, kernel_size=3)
    check(run(m, args(g, False, False))[0], g["out"], "Box3dAttention(fixed) out")


# ------------------------------------------------------------------ nn.Modules at the model's head geometry (G9)
# d = 256, 8 heads (32 channels per head), 4 levels, 2 x 2 points: the shapes the window-staged / gather / binned
# kernels run (G7 / G8: d = 32 with 4 heads = 8 channels per head, generic kernels only).  Outputs AND gradients --
# of query, value, reference windows and every parameter -- against what torch's autograd gives for the reference's
# own module code in float64 (tests/golden/make_goldens.py g9), for every opt-in path, with the path that ran asserted.
G9 = {
    "G9_module_box_enc": ("BoxAttention", dict(kernel_size=2), 1),
    "G9_module_box_enc_masked": ("BoxAttention", dict(kernel_size=2), 1),
    "G9_module_box_dec": ("BoxAttention", dict(kernel_size=2), 1),
    "G9_module_box3d_fixed_enc": ("Box3dAttention", dict(with_rotation=False, kernel_size=2), 1),
    "G9_module_box3d_rot_dec": ("Box3dAttention", dict(with_rotation=True, kernel_size=2), 1),
    "G9_module_inst_k4": ("InstanceAttention", dict(kernel_size=4), 2),
}


def _g9_run(name, dtype, fused_grid=0, fused_pointwise=False, native_bf16=False):
    """-> (module, outputs, {name: gradient}, golden, spies)"""
    import boxer_amd
    from boxer_amd import ops
    cls_name, kw, n_out = G9[name]
    g = golden_io.load(name)
    m = _load_module(getattr(boxer_amd, cls_name), g, d_model=256, num_level=4, num_head=8, **kw).to(dtype)
    m.fused_grid, m.fused_pointwise, m.native_bf16 = fused_grid, fused_pointwise, native_bf16
    if cls_name == "InstanceAttention":
        m.inferencing = False

    def t(key, grad=False):
        if key not in g:
            return None
        x = dev(g[key])
        x = x.to(dtype) if x.is_floating_point() else x
        return x.requires_grad_() if grad else x
    query, value, rw = t("query", True), t("value", True), t("ref_windows", True)
    spies = {"plans": []}
    orig_train = ops._forward_train

    def spy_train(*a, **k):
        spies["plans"].append(orig_train(*a, **k))
        return spies["plans"][-1]
    ops._forward_train = spy_train
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=native_bf16):
            outs = m(query, value, dev(g["shapes"]), t("v_mask"), dev(g["lsi"]), t("ratios"), rw)[:n_out]
        loss = sum((o.double() * dev(g["gout%d" % i]).double()).sum() for i, o in enumerate(outs))
        loss.backward()
    finally:
        ops._forward_train = orig_train
    grads = {"query": query.grad, "value": value.grad, "ref_windows": rw.grad}
    grads.update({"param." + k: p.grad for k, p in m.named_parameters()})
    return m, outs, grads, g, spies


def _g9_check(name, outs, grads, g, tol, what, rms_only=()):
    """Every output and gradient: max |got - want| / max(1, max |want|) <= tol; for the tensors named in `rms_only`
    the relative root-mean-square error instead (bf16 autocast: the box offsets are bf16 numbers, a few sample
    points per image land on the other side of a bilinear cell edge than in the float64 run, and the gradients
    that flow through the LOCATIONS -- reference windows, box projection, query -- jump there; the reference under
    autocast has the same sensitivity)."""
    def errs(got, want):
        got = got.detach().double().cpu().numpy()
        want = np.asarray(want, dtype=np.float64).reshape(got.shape)
        return (float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max())),
                float(np.sqrt(((got - want) ** 2).mean()) / max(1e-30, np.sqrt((want ** 2).mean()))))
    worst = {}
    for i, o in enumerate(outs):
        worst["out%d" % i] = errs(o, g["out%d" % i])[0]
    for k, v in grads.items():
        key = "grad_" + k if not k.startswith("param.") else "grad." + k[6:]
        assert v is not None, "%s: no gradient reached %s" % (name, k)
        e_max, e_rms = errs(v, g[key])
        worst[k] = e_rms / 2 if any(r in k for r in rms_only) else e_max       # (rms criterion: 2 tol)
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, "%s %s: errors above %.0e: %s" % (name, what, tol, bad)


@pytest.mark.parametrize("name", sorted(G9))
def test_g9_modules_float64(name):
    """float64 (the generic kernels): the module glue -- projections, box decoding, softmax, masks, rotation -- and
    every gradient through it agree with the reference's module code to float32 storage precision."""
    _, outs, grads, g, _ = _g9_run(name, torch.float64)
    _g9_check(name, outs, grads, g, 2e-6, "float64")


@pytest.mark.parametrize("native_bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("fused_pointwise", [False, True], ids=["torch_pointwise", "fused_pointwise"])
@pytest.mark.parametrize("fused_grid", [0, 1])
@pytest.mark.parametrize("name", sorted(G9))
def test_g9_modules_fast_paths(name, fused_grid, fused_pointwise, native_bf16):
    """float32 / bf16 storage on the fast kernel families, every opt-in path (fused_grid: grid kernels; fused_pointwise;
    native bf16) -- and it is ASSERTED that the fast path ran: the training forward went through the training entry
    (32 channels per head: the binned backward is eligible), and the bf16 encoder shapes ran the window-staged forward
    (it is the only kernel that feeds the locality counters of the state buffer)."""
    from boxer_amd import ops
    ops.release_workspaces()
    m, outs, grads, g, spies = _g9_run(name, torch.float32, fused_grid, fused_pointwise, native_bf16)
    _g9_check(name, outs, grads, g, 4e-2 if native_bf16 else 2e-4,
              "fused_grid=%s pointwise=%s bf16=%s" % (fused_grid, fused_pointwise, native_bf16),
              rms_only=("ref_windows", "linear_box", "query") if native_bf16 else ())
    assert spies["plans"] and all(p is not None for p in spies["plans"]), "no plan: not on the binned fast path"
    if native_bf16 and name.endswith(("_enc", "_enc_masked")):
        torch.cuda.synchronize()
        counted = sum(int(st[:1024].view(torch.int64).sum().item()) for st in ops._STATE.values())
        assert counted > 0, "the window-staged forward did not run (no locality counts)"


# ------------------------------------------------------------------ box -> grid (opt-in)
def _torch_grid(ref, off, kidx, vr, angle_mode):
    """The reference modules' _where_to_attend after the offset projection
    (box_attention.py:63-81, 304-338)."""
    r = ref[:, :, None, None] if ref.dim() == 3 else ref[:, :, :, None]
    wh = r[..., 2:4]
    boxes = r[..., :4] + off[..., :4] / 8 * torch.cat([wh, wh], dim=-1)
    boxes = boxes.unsqueeze(-2)
    c, size = boxes[..., :2], boxes[..., 2:]
    local = kidx * torch.relu(size)
    if angle_mode:
        ang = (r[..., 4:5] + off[..., 4:5] / 16) * 2 * math.pi if angle_mode == 1 \
            else r[..., 4:5].expand(off.shape[:4] + (1,))
        cos, sin = torch.cos(ang), torch.sin(ang)
        lx, ly = local[..., 0], local[..., 1]
        local = torch.stack([lx * cos - ly * sin, lx * sin + ly * cos], dim=-1)
    grid = c + local
    return grid * vr if vr is not None else grid


@pytest.mark.parametrize("angle_mode", [0, 1, 2])
@pytest.mark.parametrize("P", [1, 4, 9, 196])
@pytest.mark.parametrize("per_head,with_ratio", [(False, False), (False, True), (True, True)])
def test_box_grid_op_matches_torch(angle_mode, P, per_head, with_ratio):
    from boxer_amd import BoxGridFunction
    gen = torch.Generator(device="cuda").manual_seed(100 + 10 * angle_mode + P)
    B, Lq, H, L = 2, 37, 3, 2
    V, D = (5 if angle_mode == 1 else 4), (7 if angle_mode else 4)
    ref = torch.rand((B, Lq, H, D) if per_head else (B, Lq, D), device="cuda", generator=gen)
    off = torch.randn(B, Lq, H, L, V, device="cuda", generator=gen) * 4    # some sizes <= 0
    k = int(round(P ** 0.5))
    kidx = (torch.rand(P, 2, device="cuda", generator=gen) - 0.5) if k * k != P else \
        torch.stack(torch.meshgrid(torch.linspace(-0.5, 0.5, k, device="cuda"),
                                   torch.linspace(-0.5, 0.5, k, device="cuda"),
                                   indexing="ij")[::-1], -1).reshape(P, 2).contiguous()
    vr = (0.5 + 0.5 * torch.rand(B, 1, 1, L, 1, 2, device="cuda", generator=gen)) \
        if with_ratio else None
    w = torch.randn(B, Lq, H, L, P, 2, device="cuda", generator=gen)

    r1, o1 = ref.clone().requires_grad_(), off.clone().requires_grad_()
    want = _torch_grid(r1, o1, kidx, vr, angle_mode)
    (want * w).sum().backward()
    r2, o2 = ref.clone().requires_grad_(), off.clone().requires_grad_()
    got = BoxGridFunction.apply(r2, o2, kidx, vr, angle_mode)
    (got * w).sum().backward()
    assert got.shape == want.shape
    assert (1 + off[..., 2:4] / 8 <= 0).any(), "the case should contain collapsed boxes (relu)"
    if angle_mode == 0:
        assert torch.equal(got, want.detach()), "grid is not bit-identical to the torch ops"
    else:
        torch.testing.assert_close(got, want.detach(), rtol=0, atol=3e-6)
    tol = 2e-5 * max(1.0, P ** 0.5)
    torch.testing.assert_close(o2.grad, o1.grad, rtol=1e-4, atol=tol)
    torch.testing.assert_close(r2.grad, r1.grad, rtol=1e-4, atol=tol * (L * (1 if per_head else H)))
    # offsets only: no reference-window gradient is computed or returned
    o3 = off.clone().requires_grad_()
    got3 = BoxGridFunction.apply(ref, o3, kidx, vr, angle_mode)
    (got3 * w).sum().backward()
    assert torch.equal(o3.grad, o2.grad)


def test_modules_with_fused_grid():
    """module.fused_grid = True changes neither outputs nor parameter gradients."""
    from boxer_amd import Box3dAttention, BoxAttention, InstanceAttention
    torch.manual_seed(5)
    d, nl, nh = 64, 2, 4
    shapes = torch.tensor([[9, 7], [5, 4]], device="cuda")
    lsi = torch.tensor([0, 63], device="cuda")
    S, B, Lq = 83, 2, 11
    value = torch.randn(B, S, d, device="cuda")
    ratios = 0.6 + 0.4 * torch.rand(B, 1, 1, nl, 1, 2, device="cuda")
    cases = [
        (BoxAttention(d, nl, nh, 2), torch.rand(B, Lq, 4, device="cuda"), ratios),
        (BoxAttention(d, nl, nh, 3), torch.rand(B, Lq, nh, 4, device="cuda"), None),
        (InstanceAttention(d, nl, nh, 4), torch.rand(B, Lq, 4, device="cuda"), ratios),
        (Box3dAttention(d, nl, nh, True, 2), torch.rand(B, Lq, 7, device="cuda"), ratios),
        (Box3dAttention(d, nl, nh, False, 2), torch.rand(B, Lq, nh, 5, device="cuda"), None),
    ]
    for m, ref, vr in cases:
        m = m.cuda()
        with torch.no_grad():                                # make the box branch non-trivial
            m.linear_box_weight.normal_(0, 0.05)
            m.linear_attn_weight.normal_(0, 0.05)
        if isinstance(m, InstanceAttention):
            m.inferencing = False
        query = torch.randn(B, Lq, d, device="cuda")
        res = []
        for fused in (False, True):
            m.fused_grid = fused
            m.zero_grad()
            q = query.clone().requires_grad_()
            out = m(q, value, shapes, None, lsi, vr, ref)
            loss = sum((o * torch.linspace(0.5, 1.5, o.numel(), device="cuda").view_as(o)).sum()
                       for o in out if isinstance(o, torch.Tensor))
            loss.backward()
            res.append((out[0].detach(), q.grad.clone(), m.linear_box_weight.grad.clone(),
                        m.linear_box_bias.grad.clone()))
        name = type(m).__name__
        for a, b_, what in zip(res[0], res[1], ("out", "grad_query", "grad_box_w", "grad_box_b")):
            scale = max(1.0, float(a.abs().max()))
            torch.testing.assert_close(b_, a, rtol=0, atol=2e-4 * scale,
                                       msg=lambda s_, n=name, w=what: "%s %s: %s" % (n, w, s_))


# ------------------------------------------------------------------ randomised shape sweep
def _sweep_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        L = int(rng.choice([1, 2, 3, 5, 8]))
        shapes = [(int(rng.integers(1, 23)), int(rng.integers(1, 23))) for _ in range(L)]
        cases.append((shapes, int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 4, 8])),
                      int(rng.choice([16, 32, 64])), int(rng.choice([1, 7, 64, 257])),
                      int(rng.choice([1, 4, 9, 16]))))
    return cases


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("idx", range(12))
def test_random_shape_sweep(idx, dtype):
    """Random level sets (1..8 levels, maps from 1x1 to 22x22, so every block-edge and
    one-pixel-level case occurs), heads, channels, query and point counts: box and instance
    attention against the C oracle, locations partly outside [0, 1]."""
    cfg = _sweep_cases(12, seed=2024)[idx]
    g = _seeded(*cfg, seed=300 + idx, lo=-0.2, hi=1.2)
    want_out = oc.box_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                g["grad_out"])
    out, gv, gl, ga = run_box(g, dtype, "auto")
    close(out, want_out, dtype, "out %r" % (cfg,))
    close(gv, want[0], dtype, "grad_value %r" % (cfg,))
    close(gl, want[1], torch.float32, "grad_loc %r" % (cfg,), ignore=g["on_edge"])
    close(ga, want[2], torch.float32, "grad_attn %r" % (cfg,))
    want_out, want_mask = oc.instance_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"],
                                                   g["spatial_w"], g["level_w"])
    want = oc.instance_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"],
                                     g["spatial_w"], g["level_w"], g["grad_out"], g["grad_mask"])
    out, mask, gv, gl, gs, glw = run_inst(g, dtype, "auto")
    close(out, want_out, dtype, "inst out %r" % (cfg,))
    close(mask, want_mask, dtype, "inst mask %r" % (cfg,))
    close(gv, want[0], dtype, "inst grad_value %r" % (cfg,))
    close(gl, want[1], torch.float32, "inst grad_loc %r" % (cfg,), ignore=g["on_edge"])
    close(gs, want[2], torch.float32, "inst grad_spatial %r" % (cfg,))
    close(glw, want[3], torch.float32, "inst grad_level %r" % (cfg,))


# ------------------------------------------------------------------ HIP graph capture
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_backward_under_graph_capture(dtype):
    """The training forward + backward are capturable in a HIP graph (no hidden synchronisation,
    allocation or host read inside the library; the helper stream is forked / joined with events,
    which capture follows) and the replay reproduces the eager results on new input values."""
    from boxer_amd import ops
    g = _seeded([(20, 30), (10, 15), (5, 8), (3, 4)], 2, 8, 32, 333, 4, seed=21)
    dev_ = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dt) if dt else \
        torch.from_numpy(np.ascontiguousarray(a)).cuda()
    value = dev_(g["value"], dtype)
    shapes, lsi = dev_(g["shapes"]), dev_(g["lsi"])
    loc, attn = dev_(g["loc"], torch.float32), dev_(g["attn"], torch.float32)
    gout = dev_(g["grad_out"], dtype)

    def run():
        out, plan = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
        gv, gl, ga = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan)
        return out, gv, gl, ga

    for _ in range(3):                                       # warm-up: host tables, allocator
        eager = [t.clone() for t in run()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            captured = run()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    for a, b_, name in zip(captured, eager, ("out", "grad_value", "grad_loc", "grad_attn")):
        close(a, b_.double().cpu().numpy(), dtype if name in ("out", "grad_value") else torch.float32,
              "replay " + name)
    # new values in the same buffers: the graph recomputes (the captured forward rebuilds the plan)
    value.mul_(0.5)
    gout.mul_(2.0)
    graph.replay()
    torch.cuda.synchronize()
    want = run()
    torch.cuda.synchronize()
    for a, b_, name in zip(captured, want, ("out", "grad_value", "grad_loc", "grad_attn")):
        close(a, b_.double().cpu().numpy(), dtype if name in ("out", "grad_value") else torch.float32,
              "replay-2 " + name)


# ------------------------------------------------------------------ full-size properties
def test_full_size_linearity_and_checksums():
    """BASELINE configs[1] (C2) at full size, through size-independent properties:
    (1) sum_c out == box-attn of the channel-summed value is not available cheaply, so use
        linearity in value: op(a*v1 + v2) == a*op(v1) + op(v2);
    (2) sum(grad_value) == sum over points of (sum of valid corner weights * a * sum_c g);
        checked against an independent torch evaluation of the bilinear weights;
    (3) constant value field => out == (sum of in-range weights) * const, and grad_loc of
        interior points == 0."""
    from boxer_amd import ops
    torch.manual_seed(0)
    shapes = torch.tensor([(100, 100), (50, 50), (25, 25), (13, 13)], device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    B, H, C, L, P, Lq = 2, 8, 32, 4, 4, S
    v1 = torch.randn(B, S, H, C, device="cuda")
    v2 = torch.randn(B, S, H, C, device="cuda")
    loc = torch.rand(B, Lq, H, L, P, 2, device="cuda") * 1.1 - 0.05
    attn = torch.softmax(torch.randn(B, Lq, H, L * P, device="cuda"), -1).view(B, Lq, H, L, P)
    f = lambda v: ops.box_attn_forward(v, shapes, lsi, loc, attn, 64)
    lhs = f(2.5 * v1 + v2)
    rhs = 2.5 * f(v1) + f(v2)
    assert (lhs - rhs).abs().max().item() <= 1e-4 * max(1.0, rhs.abs().max().item())

    # constant field: out = const * sum_k (valid corner weight) * a
    const = torch.full((B, S, H, C), 0.75, device="cuda")
    out = f(const).view(B, Lq, H, C)
    wsum = torch.zeros(B, Lq, H, device="cuda")
    for l in range(L):
        hl, wl = [int(x) for x in shapes[l]]
        x = loc[:, :, :, l, :, 0] * wl - 0.5
        y = loc[:, :, :, l, :, 1] * hl - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        tot = torch.zeros_like(x)
        for dy, wy in ((0, 1 - ly), (1, ly)):
            for dx, wx in ((0, 1 - lx), (1, lx)):
                ok = ((y0 + dy >= 0) & (y0 + dy <= hl - 1) & (x0 + dx >= 0) & (x0 + dx <= wl - 1))
                tot = tot + torch.where(ok, wy * wx, torch.zeros_like(x))
        wsum = wsum + (tot * attn[:, :, :, l]).sum(-1)
    assert (out - 0.75 * wsum[..., None]).abs().max().item() <= 1e-4

    # backward checksum: sum(grad_value) == sum_q,h ( wsum * sum_c g )
    gout = torch.randn(B, Lq, H * C, device="cuda")
    gv, gl, ga = ops.box_attn_backward(const, shapes, lsi, loc, attn, gout, 64)
    want = (wsum * gout.view(B, Lq, H, C).sum(-1)).double().sum().item()
    got = gv.double().sum().item()
    assert abs(got - want) <= 1e-4 * max(1.0, abs(want))
    # grad_attn of a constant field: 0.75 * in-range weight of the point * sum_c g
    assert torch.isfinite(gl).all() and torch.isfinite(ga).all()


def test_host_table_cache_is_not_fooled_by_address_reuse():
    """The host copy of the level tables is cached per tensor object; a new tensor that lands
    on a recycled address with different shapes must not see the old copy."""
    from boxer_amd import ops
    outs = []
    for levels in ([(9, 7), (4, 3)], [(8, 8), (4, 4)], [(5, 6), (7, 2)]):
        g = _seeded(levels, 1, 8, 32, 12, 4, seed=2)
        shapes, lsi = dev(g["shapes"]), dev(g["lsi"])          # likely the same addresses again
        args = [dev(g["value"], torch.float32), shapes, lsi, dev(g["loc"], torch.float32),
                dev(g["attn"], torch.float32)]
        gv, gl, ga = ops.box_attn_backward(*args, dev(g["grad_out"], torch.float32), 64)
        want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                    g["grad_out"])
        close(gv, want[0], torch.float32, "grad_value %s" % (levels,))
        del shapes, lsi, args


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_binned_backward_run_to_run(dtype):
    """The binned backward uses no float atomics; run-to-run differences are limited to the
    fp32 summation order inside a bin (the record order comes from integer LDS atomics), and
    grad_loc / grad_weight are bit-identical."""
    from boxer_amd import _lib, ops
    g = _seeded([(40, 60), (20, 30), (10, 15), (5, 8)], 2, 8, 32, 3000, 4, seed=31)
    _lib.set_variant(3)
    cdt = _cdt(dtype)
    args = [dev(g["value"], dtype), dev(g["shapes"]), dev(g["lsi"]), dev(g["loc"], cdt),
            dev(g["attn"], cdt)]
    gout = dev(g["grad_out"], dtype)
    first = ops.box_attn_backward(*args, gout, 64)
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(4)]
    del junk
    for _ in range(3):
        again = ops.box_attn_backward(*args, gout, 64)
        assert torch.equal(first[1], again[1]) and torch.equal(first[2], again[2])
        scale = max(1.0, first[0].float().abs().max().item())
        tol = 1e-6 if dtype == torch.float32 else 8e-3        # bf16: one output ulp
        assert (first[0].float() - again[0].float()).abs().max().item() <= tol * scale


@pytest.mark.parametrize("C", [16, 32, 64])
def test_bf16_accumulate_single_terms_are_correctly_rounded(C):
    """bf16 box attention sums grad_value on the matrix cores, with every weight split into two
    bf16 terms (boxattn_binned_mfma.h).  One point per (query, head) on a map with one query:
    every grad_value element is a single product w*a*g (or zero), so the stored value must be
    that product rounded ONCE to bf16: within half a bf16 ulp (+2^-7 of slack for the 2^-17
    split error).  With a one-term split the error would reach a whole ulp."""
    rng = np.random.default_rng(1234 + C)
    shapes = np.asarray([(9, 11)], dtype=np.int64)
    B, H, Lq, P = 2, 8, 1, 1
    S = int(shapes.prod(1).sum())
    g = dict(shapes=shapes, lsi=np.zeros(1, dtype=np.int64),
             value=rng.integers(-127, 128, (B, S, H, C)).astype(np.float64) / 64,
             loc=rng.uniform(0.1, 0.9, (B, Lq, H, 1, P, 2)).astype(np.float32).astype(np.float64),
             attn=rng.uniform(0.2, 1.0, (B, Lq, H, 1, P)).astype(np.float32).astype(np.float64),
             grad_out=(rng.integers(1, 128, (B, Lq, H * C)) * rng.choice([-1, 1], (B, Lq, H * C))
                       ).astype(np.float64) / 32)
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                g["grad_out"])[0]
    _, gv, _, _ = run_box(g, torch.bfloat16, "binned")
    got = gv.double().cpu().numpy()
    nz = want != 0
    assert nz.sum() == B * H * 4 * C                       # four corners per point, all inside
    assert (got[~nz] == 0).all()
    half_ulp = 2.0 ** (np.floor(np.log2(np.abs(want[nz]))) - 8)
    worst = (np.abs(got[nz] - want[nz]) / half_ulp).max()
    assert worst <= 1.0 + 2.0 ** -7, "worst error %.4f half-ulps" % worst


@pytest.mark.parametrize("riders", [0, 4], ids=["one_pass", "two_pass"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_forward_plan(dtype, riders):
    """box_attn_forward_train counts and scans the sample points' destination bins for the backward
    inside the forward kernel's launch; the backward that receives the plan must give the same
    gradients, and a plan that no longer matches the locations (in-place update) must be ignored,
    not trusted."""
    from boxer_amd import _lib, ops
    _lib.set_option("riders", riders)     # 4: the forward counts and scans (a plan buffer); 0: the backward's one-pass fill
                                          # needs none -- the plan object then only carries the forward's hints
    # (1 000 queries: ~245 records a block of this map -- the one-pass fill takes shapes with >= 192)
    g = _seeded([(37, 53), (19, 27), (10, 14), (5, 7)], 2, 8, 32, 1000, 4, seed=41, lo=-0.2, hi=1.2)
    cdt = _cdt(dtype)
    value, loc, attn = dev(g["value"], dtype), dev(g["loc"], cdt), dev(g["attn"], cdt)
    shapes, lsi, gout = dev(g["shapes"]), dev(g["lsi"]), dev(g["grad_out"], dtype)
    want_out = oc.box_attn_forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"])
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                g["grad_out"])
    out, plan = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
    assert plan is not None and (plan.buf is not None) == (riders == 4)
    close(out, want_out, dtype, "out")
    gv, gl, ga = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan)
    close(gv, want[0], dtype, "grad_value (plan)")
    close(gl, want[1], torch.float32, "grad_loc (plan)", ignore=g["on_edge"])
    close(ga, want[2], torch.float32, "grad_attn (plan)")
    # stale plan: move every location, keep the tensor object
    loc.mul_(0.5).add_(0.25)
    g2 = dict(g, loc=(g["loc"].astype(np.float32) * np.float32(0.5) + np.float32(0.25)).astype(np.float64))
    want2 = oc.box_attn_backward(g2["value"], g2["shapes"], g2["lsi"], g2["loc"], g2["attn"],
                                 g2["grad_out"])
    gv2, _, _ = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan)
    close(gv2, want2[0], dtype, "grad_value (stale plan ignored)")
    # stale plan, weights this time (the bf16 flavour's records carry the attention weights)
    _, plan3 = ops.box_attn_forward_train(value, shapes, lsi, loc, attn, 64)
    attn.mul_(0.5)
    g3 = dict(g2, attn=(g["attn"].astype(np.float32) * np.float32(0.5)).astype(np.float64))
    want3 = oc.box_attn_backward(g3["value"], g3["shapes"], g3["lsi"], g3["loc"], g3["attn"],
                                 g3["grad_out"])
    gv3, _, _ = ops.box_attn_backward(value, shapes, lsi, loc, attn, gout, 64, plan=plan3)
    close(gv3, want3[0], dtype, "grad_value (plan with stale weights ignored)")


def test_training_forward_refuses_a_bad_state_buffer():
    """include/boxattn.h: a non-NULL state buffer must be 8-byte aligned and boxattn_state_bytes(...) bytes long -- a
    buffer that is not is an error (hipErrorInvalidValue = 1), not silently "no state" (which would cost the locality
    counters and a zero-fill launch per forward without anybody noticing)."""
    import ctypes
    from boxer_amd import _lib
    lib = _lib.load()
    g = _seeded([(20, 30), (10, 15)], 1, 8, 32, 800, 4, seed=59)     # (many records a block: a one-pass shape)
    value, loc, attn = dev(g["value"], torch.float32), dev(g["loc"], torch.float32), dev(g["attn"], torch.float32)
    shapes, lsi = dev(g["shapes"]), dev(g["lsi"])
    B, S, H, C = value.shape
    L, Lq, P = shapes.size(0), loc.size(1), loc.size(4)
    out = torch.empty(B, Lq, H * C, device="cuda")
    sh, ls = np.ascontiguousarray(g["shapes"]), np.ascontiguousarray(g["lsi"])
    nplan = int(lib.boxattn_plan_bytes(0, B, S, H, C, L, Lq, P, sh.ctypes.data, ls.ctypes.data))
    plan = torch.empty(nplan, dtype=torch.uint8, device="cuda")
    nstate = int(lib.boxattn_state_bytes(B, S, H, C, L, Lq, P, sh.ctypes.data, ls.ctypes.data))
    state = torch.zeros(nstate + 16, dtype=torch.uint8, device="cuda")
    lib.boxattn_set_option(15, 4)       # the two-pass riders: the training forward builds a plan (the default one-pass
                                        # fill of the backward needs none: *plan_built stays 0, asserted below)
    built = ctypes.c_int(0)
    stream = torch.cuda.current_stream().cuda_stream

    def call(state_ptr, state_bytes):
        return lib.boxattn_fwd_train_f32(value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
                                         attn.data_ptr(), B, S, H, C, L, Lq, P, out.data_ptr(), sh.ctypes.data,
                                         ls.ctypes.data, plan.data_ptr(), plan.numel(), state_ptr, state_bytes, 0,
                                         ctypes.addressof(built), stream)
    assert call(state.data_ptr(), nstate) == 0 and built.value == 1
    assert call(state.data_ptr() + 4, nstate) == 1, "misaligned state"
    assert call(state.data_ptr(), nstate - 8) == 1, "undersized state"
    assert call(0, 0) == 0 and built.value == 1, "no state at all is fine (tickets in the plan buffer)"
    lib.boxattn_set_option(15, 0)
    assert call(state.data_ptr(), nstate) == 0 and built.value == 0, "one-pass backward: nothing to hand over"
    assert call(state.data_ptr(), nstate - 8) == 1, "undersized state"
    assert call(0, 0) == 0 and built.value == 1, "without a state the backward is the two-pass one: a plan"
    torch.cuda.synchronize()


def _ref_style_function(mod):
    """An autograd Function in the reference's own shape (box_attention_func.py:10-64): forward saves the five
    tensors and calls ``box_attn_forward``; backward calls ``box_attn_backward`` with nothing but them."""
    from torch.autograd import Function
    from torch.autograd.function import once_differentiable

    class RefStyleBoxAttn(Function):
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
    return RefStyleBoxAttn


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reference_api_parks_the_plan(dtype):
    """The reference's four-function API gets the training step's fast path: a ``box_attn_forward`` whose inputs
    require a gradient builds the backward's plan and parks it; the ``box_attn_backward`` that follows -- called, as
    the reference's Function calls it, with the saved tensors only -- finds it: NO count / scan launches in the
    backward (the library's "bwd_binning" timing slot stays empty), same gradients as the oracle's.  A forward
    without gradient requirements parks nothing, and a backward that finds nothing plans for itself."""
    from boxer_amd import _lib, ops
    # (the two-pass riders of ABI 7, boxattn_set_option(15, 4): the default one-pass fill of the backward takes no plan
    # -- tests/test_gpu_onepass.py -- but instance attention and the shapes it does not take still park theirs)
    _lib.set_option("riders", 4)
    g = _seeded([(37, 53), (19, 27), (10, 14), (5, 7)], 2, 8, 32, 700, 4, seed=47)
    cdt = _cdt(dtype)
    shapes, lsi = dev(g["shapes"]), dev(g["lsi"])
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"], g["grad_out"])
    fn = _ref_style_function(ops)

    def step(requires_grad):
        v = dev(g["value"], dtype).requires_grad_(requires_grad)
        l = dev(g["loc"], cdt).requires_grad_(requires_grad)
        a = dev(g["attn"], cdt).requires_grad_(requires_grad)
        _lib.profile_begin()
        try:
            if requires_grad:
                out = fn.apply(v, shapes, lsi, l, a, 64)
                parked = len(ops._PARKED)
                out.backward(dev(g["grad_out"], dtype))
                grads = (v.grad, l.grad, a.grad)
            else:
                ops.box_attn_forward(v, shapes, lsi, l, a, 64)
                parked = len(ops._PARKED)
                grads = ops.box_attn_backward(v, shapes, lsi, l, a, dev(g["grad_out"], dtype), 64)
            torch.cuda.synchronize()
        finally:
            slots = _lib.profile_end()
        return grads, parked, slots

    grads, parked, slots = step(True)
    assert parked == 1 and len(ops._PARKED) == 0, "forward parks one plan, the backward takes it"
    assert slots["bwd_binning"]["launches"] == 0, slots
    close(grads[0], want[0], dtype, "grad_value (parked plan)")
    close(grads[1], want[1], torch.float32, "grad_loc (parked plan)", ignore=g["on_edge"])
    close(grads[2], want[2], torch.float32, "grad_attn (parked plan)")
    grads, parked, slots = step(False)
    assert parked == 0
    assert slots["bwd_binning"]["launches"] > 0, "no plan: the backward counts and scans itself"
    close(grads[0], want[0], dtype, "grad_value (own plan)")
    old = ops._PARK_PLANS
    ops.set_plan_parking(False)
    try:
        _, parked, slots = step(True)
        assert parked == 0 and slots["bwd_binning"]["launches"] > 0
    finally:
        ops.set_plan_parking(old)


def test_parked_plan_is_not_taken_by_other_tensors():
    """A parked plan names its location / weight tensors by address and version and holds them while it is parked:
    a backward with OTHER tensors of the same shape -- whatever their addresses -- must not get it, and a backward
    after an in-place update of the locations must not either."""
    from boxer_amd import _lib, ops
    _lib.set_option("riders", 4)          # (two-pass riders: the forward parks a plan)
    g = _seeded([(20, 30), (10, 15), (5, 8), (3, 4)], 2, 8, 32, 300, 4, seed=53)
    shapes, lsi = dev(g["shapes"]), dev(g["lsi"])
    v = dev(g["value"], torch.float32).requires_grad_()
    l = dev(g["loc"], torch.float32).requires_grad_()
    a = dev(g["attn"], torch.float32).requires_grad_()
    gout = dev(g["grad_out"], torch.float32)
    ops.box_attn_forward(v, shapes, lsi, l, a, 64)
    assert len(ops._PARKED) == 1
    # other tensors, same shapes: while the plan is parked their memory cannot be the parked tensors' memory
    g2 = _seeded([(20, 30), (10, 15), (5, 8), (3, 4)], 2, 8, 32, 300, 4, seed=54)
    l2, a2 = dev(g2["loc"], torch.float32), dev(g2["attn"], torch.float32)
    want2 = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g2["loc"], g2["attn"], g["grad_out"])
    gv2, _, _ = ops.box_attn_backward(v.detach(), shapes, lsi, l2, a2, gout, 64)
    close(gv2, want2[0], torch.float32, "grad_value (other tensors)")
    assert len(ops._PARKED) == 1, "the parked plan is still waiting for ITS backward"
    # in-place update of the parked locations: the version no longer matches
    with torch.no_grad():
        l.mul_(0.5).add_(0.25)
    g3 = dict(g, loc=(g["loc"].astype(np.float32) * np.float32(0.5) + np.float32(0.25)).astype(np.float64))
    want3 = oc.box_attn_backward(g3["value"], g3["shapes"], g3["lsi"], g3["loc"], g3["attn"], g3["grad_out"])
    gv3, _, _ = ops.box_attn_backward(v.detach(), shapes, lsi, l.detach(), a.detach(), gout, 64)
    close(gv3, want3[0], torch.float32, "grad_value (stale parked plan not taken)")
    ops.release_workspaces()
    assert len(ops._PARKED) == 0


@pytest.mark.parametrize("plan_in_forward", [False, True])
def test_functions_use_the_plan_and_match(plan_in_forward):
    """The autograd Functions with and without the forward preparing the backward's plan
    (functions.PLAN_IN_FORWARD, on by default)."""
    from boxer_amd import BoxAttnFunction, InstanceAttnFunction, functions
    old = functions.set_plan_in_forward(plan_in_forward)
    try:
        _functions_match()
    finally:
        functions.set_plan_in_forward(old)


def _functions_match():
    from boxer_amd import BoxAttnFunction, InstanceAttnFunction
    g = _seeded([(20, 30), (10, 15), (5, 8), (3, 4)], 2, 8, 32, 200, 4, seed=43)
    shapes, lsi = dev(g["shapes"]), dev(g["lsi"])
    v = dev(g["value"], torch.float32).requires_grad_()
    l = dev(g["loc"], torch.float32).requires_grad_()
    a = dev(g["attn"], torch.float32).requires_grad_()
    out = BoxAttnFunction.apply(v, shapes, lsi, l, a, 64)
    out.backward(dev(g["grad_out"], torch.float32))
    want = oc.box_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attn"],
                                g["grad_out"])
    close(v.grad, want[0], torch.float32, "grad_value")
    close(a.grad, want[2], torch.float32, "grad_attn")
    # no gradient requested -> plain forward, no plan work
    with torch.no_grad():
        out2 = BoxAttnFunction.apply(v.detach(), shapes, lsi, l.detach(), a.detach(), 64)
    assert torch.equal(out, out2)
    sw = dev(g["spatial_w"], torch.float32).view(2, 200, 8, 4, 2, 2).requires_grad_()
    lw = dev(g["level_w"], torch.float32).view(2, 200, 8, 4, 2, 2).requires_grad_()
    v.grad = None
    o, m = InstanceAttnFunction.apply(v, shapes, lsi, l, sw, lw, 2, 64)
    torch.autograd.backward([o, m], [dev(g["grad_out"], torch.float32),
                                     dev(g["grad_mask"], torch.float32).view_as(m)])
    wi = oc.instance_attn_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["spatial_w"],
                                   g["level_w"], g["grad_out"], g["grad_mask"])
    close(v.grad, wi[0], torch.float32, "instance grad_value")
    close(sw.grad, wi[2].reshape(sw.shape), torch.float32, "instance grad_spatial")


# ------------------------------------------------------------------ pointwise passes (N3)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [4, 16, 36, 64])
def test_softmax_passes_match_torch(n, dtype):
    from boxer_amd import LogitSoftmaxFunction
    g = torch.Generator(device="cuda").manual_seed(n)
    logits = (3 * torch.randn(3, 50, 8, n, device="cuda", generator=g)).to(dtype).requires_grad_()
    up = torch.randn(3, 50, 8, n, device="cuda", generator=g)
    got = LogitSoftmaxFunction.apply(logits)
    assert got.dtype == torch.float32
    got.backward(up)
    ref_in = logits.detach().double().requires_grad_()
    want = torch.softmax(ref_in, -1)
    want.backward(up.double())
    assert (got.double() - want).abs().max().item() <= 1e-6
    tol = 1e-6 if dtype == torch.float32 else 4e-3            # grad_logits rounded to bf16
    assert logits.grad.dtype == dtype
    assert (logits.grad.double() - ref_in.grad).abs().max().item() <= tol * max(
        1.0, ref_in.grad.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_value_mask_cast_matches_torch(dtype):
    from boxer_amd import ValueMaskCastFunction
    g = torch.Generator(device="cuda").manual_seed(3)
    value = torch.randn(2, 333, 256, device="cuda", generator=g).to(dtype).requires_grad_()
    mask = torch.rand(2, 333, device="cuda", generator=g) < 0.2
    for m in (mask, None):
        got = ValueMaskCastFunction.apply(value, m)
        want = value.detach().to(torch.bfloat16)
        if m is not None:
            want = want.masked_fill(m[..., None], 0)
        assert got.dtype == torch.bfloat16 and torch.equal(got, want)
        value.grad = None
        got.backward(torch.ones_like(got))
        want_g = torch.ones_like(value)
        if m is not None:
            want_g = want_g.masked_fill(m[..., None], 0)
        assert value.grad.dtype == dtype and torch.equal(value.grad, want_g)


def test_modules_with_fused_pointwise():
    """``module.fused_pointwise``: the softmax / mask + cast passes leave outputs and parameter
    gradients of the modules unchanged (float32 modules; bf16 storage mode for the value pass)."""
    from boxer_amd import Box3dAttention, BoxAttention, InstanceAttention
    torch.manual_seed(0)
    shapes = torch.tensor([(12, 9), (6, 5)], device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    B, Lq, d = 2, 17, 64
    query = torch.randn(B, Lq, d, device="cuda")
    value = torch.randn(B, S, d, device="cuda")
    v_mask = torch.rand(B, S, device="cuda") < 0.15
    ref = torch.rand(B, Lq, 5, device="cuda") * 0.5 + 0.2
    for cls, kw, native in ((BoxAttention, {}, False), (BoxAttention, {}, True),
                            (Box3dAttention, {"with_rotation": True}, False),
                            (InstanceAttention, {"kernel_size": 4}, False)):
        m = cls(d, 2, 8, **kw).cuda()
        m.native_bf16 = native
        m.inferencing = False
        with torch.no_grad():
            m.linear_box_weight.normal_(0, 0.3)
            m.linear_attn_weight.normal_(0, 0.3)
        rw = ref if cls is Box3dAttention else ref[..., :4]
        outs = []
        for fused in (False, True):
            m.fused_pointwise = fused
            m.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=native):
                res = m(query, value, shapes, v_mask, lsi, None, rw)
            res[0].float().square().sum().backward()
            outs.append((res[0].detach().float(),
                         [p.grad.detach().clone() for p in m.parameters()]))
        tol = 2e-2 if native else 1e-5
        scale = max(1.0, outs[0][0].abs().max().item())
        assert (outs[0][0] - outs[1][0]).abs().max().item() <= tol * scale, cls.__name__
        for ga, gb in zip(outs[0][1], outs[1][1]):
            assert (ga - gb).abs().max().item() <= tol * max(1.0, ga.abs().max().item()), cls.__name__


# ------------------------------------------------------------------ boxes straight into the kernels (N1, step 2)
