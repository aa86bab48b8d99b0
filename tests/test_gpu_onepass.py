"""GPU tests of the backward's one-pass fill (boxer_amd/csrc/boxattn_spec.h, C ABI 8): box attention whose accumulate
runs on the matrix cores bins its sample points in ONE pass, into record ranges that the previous call on the same
caller-owned state buffer planned from its own counts.  What is pinned here:

  * results never depend on what the state holds -- cold (zeroed), planned by the same data, planned by OTHER data (ranges
    too small: the redo workers recompute the blocks from the sampling locations), zeroed behind the library's back,
    handed over from another shape;
  * the steady state really is one pass: no count / scan launch, no plan, the chain of the slice's last rider ran (the
    state's counters say so);
  * every step of a long run over changing inputs matches the CPU oracle.

The oracle comparison is bench.parity_report (oracle/boxattn_oracle.c on the same -- for bf16: the rounded -- inputs;
the reference's forward / backward allclose tests, tests/box_attn_test.py:96-159, are the model).
"""
import ctypes

import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu

OPT_RIDERS = 15         # 0 default (one-pass fill where eligible), 1 launches of their own, 4 two-pass riders (ABI 7)
STAT_OFF = 1024         # the one-pass counters behind the locality counters: {calls, blocks redone} (include/boxattn.h)


def _lib():
    from boxer_amd import _lib
    return _lib.load()


def make_case(levels, lq, family="model", B=2, H=8, C=32, seed=0, dtype=torch.bfloat16):
    name = "_onepass_test"
    bench.WORKLOADS[name] = (list(levels), lq, 4, "box")
    old = bench.H_HEADS, bench.C_HEAD
    bench.H_HEADS, bench.C_HEAD = H, C
    try:
        return bench.make_inputs(name, dtype, "cuda", family=family, batch=B, seed=seed)
    finally:
        bench.H_HEADS, bench.C_HEAD = old
        del bench.WORKLOADS[name]


def step(inp, entry="train"):
    from boxer_amd import ops
    v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
    if entry == "train":
        out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
        grads = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)
    else:                      # the reference's two calls: nothing but the tensors goes from one to the other
        out = ops.box_attn_forward(v, sh, ls, loc, attn, 64)
        grads = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64)
    torch.cuda.synchronize()
    return out, grads


def check(inp, out, grads, what):
    for name, worst, tol in bench.parity_report(inp, out, grads):
        assert worst <= tol, "%s: %s worst |err| / (max(1, rms) + |want|) = %.3e > %.0e" % (what, name, worst, tol)


def counters():
    """{calls of the one-pass chain, blocks redone} summed over the state buffers boxer_amd.ops keeps."""
    from boxer_amd import ops
    tot = np.zeros(2, dtype=np.int64)
    for st in ops._STATE.values():
        tot += st[STAT_OFF:STAT_OFF + 16].view(torch.int64).cpu().numpy()
    return int(tot[0]), int(tot[1])


def binning_launches(inp, entry="train"):
    from boxer_amd import _lib
    _lib.profile_begin()
    try:
        out, grads = step(inp, entry)
    finally:
        slots = _lib.profile_end()
    return out, grads, slots["bwd_binning"]["launches"]


LEVELS4 = [(37, 53), (19, 27), (10, 14), (5, 7)]


@pytest.mark.parametrize("dtype,C", [(torch.bfloat16, 32), (torch.float32, 32), (torch.bfloat16, 16), (torch.bfloat16, 64)],
                         ids=["bf16", "f32", "bf16_c16", "bf16_c64"])
@pytest.mark.parametrize("lq", ["S", 900], ids=["encoder", "dense_decoder"])
@pytest.mark.parametrize("entry", ["train", "reference"])
def test_cold_then_one_pass_matches_oracle(dtype, C, lq, entry):
    """First call on a fresh state: the two-pass passes as launches (and the ranges planned from their scan).  From the
    second call on: no binning launch, the chain ran once per (image, head) slice, nothing had to be redone on the same
    data -- and every call's tensors are the oracle's.  Through both entries: the training entry points, and the
    reference's box_attn_forward / box_attn_backward pair (which needs no parked plan any more)."""
    from boxer_amd import ops
    ops.release_workspaces()
    inp = make_case(LEVELS4, lq, dtype=dtype, C=C, seed=3)
    if entry == "reference":
        inp["value"].requires_grad_()              # (the forward of a training step: an input requires a gradient)
    ns = inp["dims"]["B"] * inp["dims"]["H"]
    out, grads, launches = binning_launches(inp, entry)
    check(inp, out, grads, "cold call")
    assert launches > 0, "a cold state runs the two-pass passes"
    assert counters() == (0, 0)
    assert len(ops._PARKED) == 0, "the one-pass backward takes no plan: nothing is parked"
    for it in range(3):
        out, grads, launches = binning_launches(inp, entry)
        check(inp, out, grads, "one-pass call %d" % it)
        assert launches == 0, "steady state: no count / scan / fill launch"
        assert counters() == ((it + 1) * ns, 0), "one chain per slice and call, no block redone on the same data"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_ranges_planned_by_other_data_are_redone(dtype):
    """The ranges follow the PREVIOUS call's counts.  Plan them with every sample in the top-left corner of the maps,
    then sample the bottom-right corner: nearly every block that gets records has outgrown its (64-record) range and
    is recomputed by a redo worker from the sampling locations -- same tensors as the oracle's -- and the call after
    that, on ranges planned by the new data, redoes nothing."""
    from boxer_amd import ops
    ops.release_workspaces()
    a = make_case(LEVELS4, 900, family="test", dtype=dtype, seed=5)
    b = dict(a)
    a["loc"] = (a["loc"] * 0.3).contiguous()                      # top-left
    b["loc"] = (b["loc"] * 0.3 + 0.65).contiguous()               # bottom-right
    step(a)                                                       # cold: plans the ranges from a's counts
    out, grads = step(a)
    check(a, out, grads, "one-pass call on the data that planned the ranges")
    assert counters()[1] == 0
    out, grads, launches = binning_launches(b)
    assert launches == 0
    check(b, out, grads, "ranges planned by other data")
    redone = counters()[1]
    assert redone > 0, "no block outgrew its range?"
    out, grads = step(b)
    check(b, out, grads, "ranges re-planned")
    assert counters()[1] == redone, "the call before re-planned the ranges from ITS counts: nothing to redo"


def test_a_state_zeroed_behind_the_librarys_back_is_a_valid_state():
    """A zeroed buffer is a state whose every range is empty: the library believes it planned (its host-side note says
    so), every block with a record is recomputed by the redo workers, blocks without records are zero-filled by their
    items -- the oracle's tensors -- and the chain plans the ranges anew."""
    from boxer_amd import ops
    ops.release_workspaces()
    inp = make_case(LEVELS4, "S", dtype=torch.bfloat16, seed=7)
    step(inp); step(inp)
    for st in ops._STATE.values():
        st.zero_()
    out, grads = step(inp)
    check(inp, out, grads, "zeroed state")
    assert counters()[1] > 0
    redone = counters()[1]
    out, grads = step(inp)
    check(inp, out, grads, "after the zeroed state")
    assert counters()[1] == redone


def test_one_state_buffer_handed_from_shape_to_shape():
    """C ABI: a state buffer belongs to ONE geometry; the library notices a buffer that turns up with another one (its
    ranges and tickets mean nothing there), zeroes it and starts cold.  Alternate two shapes on one buffer through the
    raw entry points; every call's grad_value is the oracle's."""
    from boxer_amd import _lib, ops
    lib = _lib.load()
    cases = [make_case(LEVELS4, "S", dtype=torch.float32, seed=11),
             make_case([(20, 30), (10, 15), (5, 8)], 400, B=3, dtype=torch.float32, seed=12)]
    sizes = []
    for inp in cases:
        d = inp["dims"]
        dims = [d[k] for k in ("B", "S", "H", "C", "L", "Lq", "P")]
        sh, ls = inp["shapes"].cpu().numpy(), inp["lsi"].cpu().numpy()
        sizes.append((dims, sh, ls, int(lib.boxattn_state_bytes(*dims, sh.ctypes.data, ls.ctypes.data)),
                      int(lib.boxattn_bwd_workspace_bytes(0, *dims, sh.ctypes.data, ls.ctypes.data))))
    state = torch.zeros(max(s[3] for s in sizes), dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(s[4] for s in sizes), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rnd in range(3):
        for inp, (dims, sh, ls, _, _) in zip(cases, sizes):
            for rep in range(2):         # cold (or zeroed: the other shape was here), then one pass
                gv = torch.empty_like(inp["value"])
                gl, ga = torch.empty_like(inp["loc"]), torch.empty_like(inp["attn"])
                rc = lib.boxattn_bwd_ws_f32(inp["value"].data_ptr(), inp["shapes"].data_ptr(), inp["lsi"].data_ptr(),
                                            inp["loc"].data_ptr(), inp["attn"].data_ptr(), inp["grad_out"].data_ptr(),
                                            *dims, gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), sh.ctypes.data,
                                            ls.ctypes.data, ws.data_ptr(), ws.numel(), 0, 0, state.data_ptr(),
                                            state.numel(), 0, stream)
                assert rc == 0
                torch.cuda.synchronize()
                out = ops.box_attn_forward(inp["value"], inp["shapes"], inp["lsi"], inp["loc"], inp["attn"], 64)
                check(inp, out, [gv, gl, ga], "round %d rep %d" % (rnd, rep))
    # a state that is too small for the shape, or misaligned, is an error -- not "no state"
    inp, (dims, sh, ls, nstate, _) = cases[0], sizes[0]
    args = lambda ptr, n: (inp["value"].data_ptr(), inp["shapes"].data_ptr(), inp["lsi"].data_ptr(),
                           inp["loc"].data_ptr(), inp["attn"].data_ptr(), inp["grad_out"].data_ptr(), *dims,
                           gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), sh.ctypes.data, ls.ctypes.data, ws.data_ptr(),
                           ws.numel(), 0, 0, ptr, n, 0, stream)
    gv, gl, ga = torch.empty_like(inp["value"]), torch.empty_like(inp["loc"]), torch.empty_like(inp["attn"])
    assert lib.boxattn_bwd_ws_f32(*args(state.data_ptr(), nstate - 8)) == 1
    assert lib.boxattn_bwd_ws_f32(*args(state.data_ptr() + 4, nstate)) == 1
    assert lib.boxattn_bwd_ws_f32(*args(0, 0)) == 0              # no state: the backward plans for itself
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_soak_over_changing_inputs(dtype, monkeypatch):
    """A long run in which every step samples other locations (eight input sets of the model-like family and two of
    uniformly random locations, cycled): ranges planned by one set serve the next; whatever overflows is redone.
    Every step's grad_value is compared with the first step that ran on the same set (within the summation-order
    tolerance: the record order inside a bin follows atomics), a few steps fully with the oracle."""
    from boxer_amd import ops
    ops.release_workspaces()
    monkeypatch.setattr(ops._Locality, "enabled", False)       # (bit-equal outputs need the same kernels in every step)
    levels = [(50, 50), (25, 25), (13, 13), (7, 7)]
    sets = [make_case(levels, "S", family="model", dtype=dtype, seed=20 + i) for i in range(8)]
    sets += [make_case(levels, "S", family="test", dtype=dtype, seed=40 + i) for i in range(2)]
    first = {}
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-4
    for it in range(400):
        k = (it * 7) % len(sets)
        out, grads = step(sets[k])
        if k not in first:
            check(sets[k], out, grads, "set %d" % k)
            first[k] = (out, grads)
            continue
        ref_out, ref_grads = first[k]
        assert torch.equal(out, ref_out) and torch.equal(grads[1], ref_grads[1]) and torch.equal(grads[2], ref_grads[2])
        err = (grads[0].float() - ref_grads[0].float()).abs().max().item()
        assert err <= tol * max(1.0, ref_grads[0].float().abs().max().item()), (it, k, err)
        if it % 97 == 0:
            check(sets[k], out, grads, "step %d" % it)
    calls, redone = counters()
    assert calls > 0


def test_one_pass_step_in_a_hip_graph():
    """A captured step replays the one-pass fill against the live state buffer (its ranges are re-planned by every
    replay); results as eager."""
    from boxer_amd import ops
    ops.release_workspaces()
    inp = make_case(LEVELS4, "S", dtype=torch.bfloat16, seed=9)
    eager_out, eager_grads = step(inp)
    step(inp)
    replay = bench.graph_step(lambda: step_nosync(inp))
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    out, grads = replay.__self__._keep
    check(inp, out, grads, "graph replay")


def step_nosync(inp):
    from boxer_amd import ops
    v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
    out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
    return out, ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)


def test_sparse_decoder_shapes_keep_the_two_pass_riders():
    """The one-pass fill plans its ranges from the previous call's counts: it is for maps whose blocks see many records,
    about as many from call to call (an encoder).  300 decoder queries leave a handful of records per block, anywhere --
    such shapes keep the two-pass riders (the forward builds a plan) and run at the same speed on changing inputs."""
    from boxer_amd import ops
    ops.release_workspaces()
    sets = [make_case([(100, 167), (50, 84), (25, 42), (13, 21)], 300, B=2, dtype=torch.bfloat16, seed=60 + i) for i in range(3)]
    for it in range(6):
        inp = sets[it % 3]
        v, sh, ls, loc, attn, go = (inp[k] for k in ("value", "shapes", "lsi", "loc", "attn", "grad_out"))
        out, plan = ops.box_attn_forward_train(v, sh, ls, loc, attn, 64)
        assert plan is not None and plan.buf is not None, "the two-pass riders: a plan"
        grads = ops.box_attn_backward(v, sh, ls, loc, attn, go, 64, plan=plan)
        torch.cuda.synchronize()
        if it >= 3:
            check(inp, out, grads, "step %d" % it)
    assert counters() == (0, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_degenerate_record_distributions(dtype):
    """What the ranges are planned from can be anything: no record at all (every location outside the maps), then every
    point of a level in ONE block (a single range has to take them all: it does not -- redone -- and the call after that
    plans it: chunked, combined), then ordinary data again.  The oracle's tensors every time."""
    from boxer_amd import ops
    ops.release_workspaces()
    base = make_case(LEVELS4, "S", dtype=dtype, seed=13)
    outside = dict(base, loc=(base["loc"] * 0 + 7.5).contiguous())
    one_block = dict(base, loc=(base["loc"] * 0.02 + 0.4).contiguous())      # ~1 x 1 pixels on every level
    seq = [("ordinary (cold)", base), ("ordinary", base), ("no records", outside), ("no records again", outside),
           ("one block", one_block), ("one block, planned", one_block), ("one block, planned again", one_block),
           ("ordinary after one block", base), ("ordinary again", base)]
    redone = []
    for what, inp in seq:
        out, grads = step(inp)
        check(inp, out, grads, what)
        redone.append(counters()[1])
    assert redone[4] > redone[3], "every point in one block: its range was planned for none"
    assert redone[6] == redone[5], "planned by the same data: nothing to redo"
    assert redone[8] == redone[7], redone
