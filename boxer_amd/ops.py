"""``boxer_amd.ops`` -- drop-in for the reference's native module ``e2edet.ops``.

Same four functions, same argument order and return values as the pybind module of the
reference (e2edet/module/ops/src/vision.cpp:7-12; host code box_attn.cu:15-135,
instance_attn.cu:15-157), implemented as thin calls into the C-ABI HIP library
(include/boxattn.h).  Same error behaviour: CPU tensors raise "Not implemented on the CPU"
(box_attn.h:53), non-contiguous tensors raise "... must be contiguous" (box_attn.cu:9-11),
``batch % min(batch, im2col_step)`` is asserted (box_attn.cu:40-42).  Kernel launch errors
raise instead of being printed (box_attn_kernel.cuh:1118-1122).

Beyond the reference: ``value`` (and the upstream gradients) may be bfloat16; sampling
locations and attention weights are then taken in float32 (bf16 ones are upcast) and all
accumulation is float32.  float32 / float64 calls behave exactly like the reference.
"""
import os

import torch

from . import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}


def _check(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor: Not implemented on the CPU" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _prepare(value, shapes, lsi, loc, weights, extra=()):
    """Validate like the reference's CHECK_INPUT block and derive the dimensions."""
    _check(value, "value")
    _check(shapes, "spatial_shapes")
    _check(lsi, "level_start_index")
    _check(loc, "sampling_loc")
    for i, w in enumerate(weights):
        _check(w, "attn_weight" if len(weights) == 1 else ("spatial_attn_weight", "level_attn_weight")[i])
    for name, t in extra:
        _check(t, name)
    if value.dtype not in _SUFFIX:
        raise RuntimeError("box_attn: unsupported dtype %s (float32, float64, bfloat16)" % value.dtype)
    if shapes.dtype != torch.int64 or lsi.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    if value.dim() != 4 or loc.dim() != 6 or loc.size(-1) != 2:
        raise RuntimeError("expected value (B,S,H,C) and sampling_loc (B,Lq,H,L,P,2)")
    B, S, H, C = value.shape
    L = shapes.size(0)
    Lq, P = loc.size(1), loc.size(4)
    if loc.size(0) != B or loc.size(2) != H or loc.size(3) != L or lsi.numel() != L:
        raise RuntimeError("sampling_loc / spatial_shapes do not match value")
    n_w = B * Lq * H * L * P
    for w in weights:
        if w.numel() != n_w:
            raise RuntimeError("attention weights must have B*Lq*H*L*P elements")
    cdt = torch.float32 if value.dtype == torch.bfloat16 else value.dtype
    if value.dtype == torch.bfloat16:
        loc = loc.float() if loc.dtype != torch.float32 else loc
        weights = [w.float() if w.dtype != torch.float32 else w for w in weights]
    elif loc.dtype != cdt or any(w.dtype != cdt for w in weights):
        raise RuntimeError("value, sampling_loc and attention weights must share one dtype")
    for name, t in extra:
        if t.dtype != value.dtype:
            raise RuntimeError("%s must have the dtype of value" % name)
    return (B, S, H, C, L, Lq, P), loc, weights, cdt


def _chunk_assert(batch, im2col_step):
    step = min(batch, int(im2col_step))
    if batch > 0:
        assert step > 0 and batch % step == 0, \
            "batch(%d) must divide im2col_step(%d)" % (batch, step)


def _host_table(t):
    """Host copy of a small int64 device table (level shapes / start indices).  The copy is
    cached ON the tensor object together with its version counter, so only the first call with
    a given tensor synchronises (BoxeR hands the same two tensors to every layer of a step).
    Keying a global cache by data_ptr would be wrong: a freed tensor's address gets reused."""
    cached = getattr(t, "_boxattn_host", None)
    if cached is None or cached[0] != t._version:
        cached = (t._version, t.detach().cpu().contiguous().numpy().copy())
        try:
            t._boxattn_host = cached
        except Exception:          # objects that refuse attributes: just do not cache
            pass
    return cached[1]


class BackwardPlan:
    """Opaque hand-over from a training forward to the matching backward: the (small) device buffer in
    which the forward's launch already counted and scanned the sample points' destination bins (the
    binning only depends on the sampling locations), plus what it is valid for."""

    __slots__ = ("buf", "key", "hints", "keep")

    def __init__(self, buf, key, hints=0):
        self.buf, self.key, self.hints = buf, key, hints       # hints: as the forward ran (BOXATTN_HINT_*)
        self.keep = None                                       # parked plans: the tensors the key names (see _park)


def _plan_key(dims, loc, weights, dtype=None):
    # (the library's option switches -- records per item, riders, accumulate flavour, variant -- change the plan's
    # layout: a plan built under other settings is not this call's plan; so does the storage type: at 16 / 64 channels
    # per head bfloat16 bins contiguous query ranges and float32 interleaved ones)
    return (tuple(dims), loc.device.index, _lib.options_epoch(), dtype) + tuple((t.data_ptr(), t._version)
                                                                                for t in (loc,) + tuple(weights))


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_guard(device):
    """torch.cuda.device(device) -- unless it is the current device already (the usual case: the context manager
    costs several microseconds of a 55 us step)."""
    return _NO_GUARD if device.index == torch.cuda.current_device() else torch.cuda.device(device)


_SIZES = {}          # (query name, is_bf16, dims, level tables) -> bytes: the size queries are pure functions
_CACHE_CAP = 1024    # entries per shape-keyed cache: a detector trained on variably padded batches sees thousands of
                     # shapes; everything cached per shape is cheap to rebuild


def _bounded(cache):
    """Shape-keyed caches start over instead of growing without bound (entries still referenced elsewhere -- a
    state buffer a running kernel uses -- live on until the caching allocator gets them back, in stream order)."""
    if len(cache) >= _CACHE_CAP:
        cache.clear()
    return cache


def _sized_buffer(query, value, shapes, lsi, dims, minimum=256):
    """A scratch tensor of the size the library asks for (query: boxattn_plan_bytes /
    boxattn_bwd_workspace_bytes; None for 0 bytes when minimum is 0) + the host level tables."""
    sh, ls = _host_table(shapes), _host_table(lsi)
    is_bf16 = int(value.dtype == torch.bfloat16)
    key = (query.__name__, is_bf16, dims, sh.tobytes(), ls.tobytes(), _lib.options_epoch())
    nbytes = _SIZES.get(key)
    if nbytes is None:
        nbytes = _bounded(_SIZES)[key] = int(query(is_bf16, *dims, sh.ctypes.data, ls.ctypes.data))
    nbytes = max(nbytes, minimum)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=value.device) if nbytes else None
    return buf, sh, ls


_STATE = {}          # (device index, stream handle, dims, dtype) -> the library's persistent state buffer


def _state_buffer(key, device, nbytes):
    """The library's state buffer (include/boxattn.h, *_fwd_train_*: `state`): zeroed once and kept -- calls on
    one stream never overlap and every call leaves the tickets in it zero -- instead of a zero-fill launch in
    front of every training forward.  One per (stream, shape): its first 1 KiB are the locality counters, and a
    _Locality must only ever see the misses of ITS shape's calls (one buffer per stream, which the library would
    be content with, let the counts of one test's uniformly random locations decide the kernels of the next
    shape's first calls)."""
    buf = _STATE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        buf._boxattn_fresh = True        # (its first call says so: BOXATTN_HINT_FRESH_STATE)
        # (a tensor first created while the stream captures a graph lives in the graph's private pool: it serves
        # this call -- the graph keeps it alive -- but must not outlive the graph in this table)
        if not torch.cuda.is_current_stream_capturing():
            _bounded(_STATE)[key] = buf
    return buf


class _Locality:
    """Data-driven choice between the window-staged and the row-gather kernels of the encoder case (VERDICT
    round 3, item 7).  The staged forward adds {points that missed their window, points sampled} to counters
    in the state buffer; after a training forward the counters are copied to pinned host memory WITHOUT a
    synchronisation, and whichever copy has arrived by the next call decides its `hints`: above
    MISS_THRESHOLD the calls run the gather kernels (BOXATTN_HINT_NOT_LOCAL); every PROBE_EVERY-th call runs
    the staged kernels again to see whether the sampling locations have become local.  One instance per
    (device, stream, dimensions); results never depend on the choice."""

    MISS_THRESHOLD = 0.5
    PROBE_EVERY = 64
    enabled = True

    def __init__(self):
        self.not_local = False
        self.calls_since_probe = 0
        self.pending = None          # a read of the counters is in flight
        self.host = self.event = None    # pinned buffer / event of the reads
        self.last = None             # counters at the previous read
        self.ratio = None            # the miss ratio the current choice is based on
        self.since_read = 0
        self.reads = 0
        self.last_hints = 0          # of the last training forward (a backward that gets no plan follows them)

    READ_EVERY = 16                  # staged calls between two reads of the counters

    def hints(self):
        # (nothing here touches an event while the stream is capturing a graph: querying one from the capturing
        # thread is not a capture-safe call; the last decision stands)
        if self.pending is not None and not torch.cuda.is_current_stream_capturing() and self.event.query():
            now = self.host.view(-1, 2).sum(0)
            self.pending = None
            # the FIRST read is the baseline (a state buffer that served an earlier _Locality of this shape carries
            # that one's counts); decisions from the second read on
            if self.last is not None:
                d_miss, d_all = int(now[0] - self.last[0]), int(now[1] - self.last[1])
                if d_all > 0:
                    self.ratio = d_miss / d_all
                    self.not_local = self.ratio > self.MISS_THRESHOLD
            self.last = now
        if not self.enabled:
            return 0
        if self.not_local:
            self.calls_since_probe += 1
            if self.calls_since_probe >= self.PROBE_EVERY:
                self.calls_since_probe = 0
                self.since_read = self.READ_EVERY       # (a probe is always read)
                return 0                                   # probe: a staged call refreshes the counters
            return _lib.HINT_NOT_LOCAL
        return 0

    def after_call(self, state, hints):
        """Queue an asynchronous read of the counters: after the first staged calls, then every READ_EVERY-th
        one (one read in flight at a time; nothing here ever waits for the GPU)."""
        if hints or self.pending is not None:
            return
        self.since_read += 1
        if self.since_read < self.READ_EVERY and self.reads >= 2:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        if self.host is None:
            self.host = torch.empty(128, dtype=torch.int64, pin_memory=True)
            self.event = torch.cuda.Event()
        self.host.copy_(state[:1024].view(torch.int64), non_blocking=True)   # the counters: the buffer's first 1 KiB
        self.event.record()
        self.pending = True
        self.since_read = 0
        self.reads += 1


_LOCALITY = {}       # (device index, stream handle, dims) -> _Locality


def _shape_key(value, stream, dims, sh, ls):
    # one state buffer per (device, stream, dimensions, storage type, level shapes): ABI 8 keeps the record ranges of the
    # backward's one-pass fill in it, which belong to ONE geometry
    return (value.device.index, stream, tuple(dims), value.dtype, sh.tobytes(), ls.tobytes())


def _fresh_hint(state):
    return _lib.HINT_FRESH_STATE if getattr(state, "_boxattn_fresh", False) else 0


def _state_for(lib, value, dims, sh, ls, stream):
    key = _shape_key(value, stream, dims, sh, ls)
    skey = ("state",) + key[2:]
    nbytes = _SIZES.get(skey)
    if nbytes is None:
        nbytes = _bounded(_SIZES)[skey] = int(lib.boxattn_state_bytes(*dims, sh.ctypes.data, ls.ctypes.data))
    return key, _state_buffer(key, value.device, nbytes)


def _locality(key):
    # (only the window-staged kernels of bf16 box attention with one query per pixel collect the counters;
    # everything else simply never sees a non-zero miss count)
    loc = _LOCALITY.get(key)
    if loc is None:
        loc = _bounded(_LOCALITY)[key] = _Locality()
    return loc


def _forward_train(name, value, shapes, lsi, loc, weights, dims, args):
    """*_fwd_train_*: forward + (when the binned backward applies) the backward's plan."""
    import ctypes
    lib = _lib.load()
    buf, sh, ls = _sized_buffer(lib.boxattn_plan_bytes, value, shapes, lsi, dims, minimum=0)
    built = ctypes.c_int(0)
    fn = getattr(lib, "%s_%s" % (name, _SUFFIX[value.dtype]))
    with _device_guard(value.device):
        stream = torch.cuda.current_stream(value.device).cuda_stream
        key, state = _state_for(lib, value, dims, sh, ls, stream)
        adapt = _locality(key)
        hints = adapt.hints()
        rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args],
                sh.ctypes.data, ls.ctypes.data, buf.data_ptr() if buf is not None else 0,
                buf.numel() if buf is not None else 0, state.data_ptr(), state.numel(), hints | _fresh_hint(state),
                ctypes.addressof(built), stream)
        if rc == 0:
            state._boxattn_fresh = False
            adapt.after_call(state, hints)
            adapt.last_hints = hints
    if rc != 0:
        _STATE.pop(key, None)                                # (its tickets may not be zero any more)
        _LOCALITY.pop(key, None)
        raise RuntimeError("%s_%s failed with hipError %d" % (name, _SUFFIX[value.dtype], rc))
    if built.value:
        return BackwardPlan(buf, _plan_key(dims, loc, weights, value.dtype), hints)
    # no plan to hand over: the backward of this shape fills its bins in one pass from the ranges in the state buffer
    # (ABI 8), or plans for itself.  The object still carries the hints the forward ran with.
    return BackwardPlan(None, _plan_key(dims, loc, weights, value.dtype), hints)


def workspace_bytes(value, shapes, lsi, dims):
    """(plan bytes, backward workspace bytes) the library asks for at these dimensions."""
    lib = _lib.load()
    sh, ls = _host_table(shapes), _host_table(lsi)
    is_bf16 = int(value.dtype == torch.bfloat16)
    return (int(lib.boxattn_plan_bytes(is_bf16, *dims, sh.ctypes.data, ls.ctypes.data)),
            int(lib.boxattn_bwd_workspace_bytes(is_bf16, *dims, sh.ctypes.data, ls.ctypes.data)))


_WORKSPACE = {}      # (device index, stream handle) -> the backward's scratch tensor (grown as needed)


def _workspace(query, value, shapes, lsi, dims, stream):
    """The backward's scratch (bin records, partial tiles: contents only live inside one call).  One tensor per
    (device, stream), reused: calls on a stream never overlap, and a 0.3 GB torch.empty per backward was a tenth
    of the host time of a training step."""
    sh, ls = _host_table(shapes), _host_table(lsi)
    is_bf16 = int(value.dtype == torch.bfloat16)
    key = (query.__name__, is_bf16, dims, sh.tobytes(), ls.tobytes(), _lib.options_epoch())
    nbytes = _SIZES.get(key)
    if nbytes is None:
        nbytes = _bounded(_SIZES)[key] = int(query(is_bf16, *dims, sh.ctypes.data, ls.ctypes.data))
    nbytes = max(nbytes, 256)
    wkey = (value.device.index, stream)
    ws = _WORKSPACE.get(wkey)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
        # cached per (device, stream) unless the caller opted out (set_workspace_caching(False): the scratch is then
        # transient inside each backward, as in rounds 1-3) or the stream is capturing a graph (private pool)
        if _CACHE_WORKSPACE and not torch.cuda.is_current_stream_capturing():
            if len(_WORKSPACE) >= _WORKSPACE_CAP:          # streams come and go: start over rather than grow
                _WORKSPACE.clear()
            _WORKSPACE[wkey] = ws
    return ws, sh, ls


_CACHE_WORKSPACE = os.environ.get("BOXATTN_CACHE_WORKSPACE", "1") != "0"
_WORKSPACE_CAP = 64


def set_workspace_caching(on):
    """Keep the backward's scratch (0.1-0.3 GB at BoxeR-R50 shapes) per (device, stream) for the life of the process
    (default: saves a torch.empty per backward, ~10 % of a step's host time) or allocate it inside every backward
    (off: it is then not resident during the forward, i.e. not part of the step's peak activation memory).
    Environment: BOXATTN_CACHE_WORKSPACE=0."""
    global _CACHE_WORKSPACE
    _CACHE_WORKSPACE = bool(on)
    if not on:
        _WORKSPACE.clear()


def release_workspaces():
    """Drop the cached scratch / state tensors and parked plans (they are re-created on demand)."""
    _WORKSPACE.clear()
    _STATE.clear()
    _LOCALITY.clear()
    _PARKED.clear()


# ---------------------------------------------------------------------------------------
# The reference's own four-function API with the training step's fast path.
#
# BoxeR's Functions call box_attn_forward, then -- in another call, with nothing but the saved tensors --
# box_attn_backward (box_attention_func.py:10-64).  When an input of the forward requires a gradient (the
# tensors keep that flag inside a Function's forward) the forward runs the *_fwd_train_* entry and PARKS the
# plan under (stream, dimensions, location / weight tensors' addresses and versions); the matching backward
# finds it there.  A backward without a parked plan plans for itself, as ever.  Bounded: a forward whose
# backward never comes (evaluation under enable_grad, a crashed step) is evicted, oldest first.
# ---------------------------------------------------------------------------------------
_PARKED = {}         # (device, stream handle, plan key) -> BackwardPlan, insertion-ordered
_PARK_CAP = 32       # plans alive between forward and backward: layers of a model x micro-batches
_PARK_PLANS = os.environ.get("BOXATTN_PARK_PLANS", "1") != "0"


def set_plan_parking(on):
    """box_attn_forward / instance_attn_forward called with inputs that require a gradient build the backward's plan
    and park it for the matching *_backward call (default) -- or do not (off: every forward is an inference forward,
    every backward plans for itself).  Environment: BOXATTN_PARK_PLANS=0."""
    global _PARK_PLANS
    _PARK_PLANS = bool(on)
    if not on:
        _PARKED.clear()


def _wants_plan(*tensors):
    return _PARK_PLANS and any(t.requires_grad for t in tensors)


def _park(value, plan, keep):
    if plan is None or plan.buf is None:
        return
    # The key names the location / weight tensors by address and version.  A parked plan HOLDS them: as long as it is
    # parked their memory cannot be freed and handed to another tensor, so a backward that presents the same
    # addresses and versions presents the same data.  (A plan handed over explicitly needs no such hold: the
    # Function's context keeps the tensors.)
    plan.keep = keep
    stream = torch.cuda.current_stream(value.device).cuda_stream
    while len(_PARKED) >= _PARK_CAP:
        _PARKED.pop(next(iter(_PARKED)))
    _PARKED[(value.device.index, stream) + plan.key] = plan


def _parked(value, dims, loc, weights):
    if not _PARKED:
        return None
    stream = torch.cuda.current_stream(value.device).cuda_stream
    plan = _PARKED.pop((value.device.index, stream) + _plan_key(dims, loc, weights, value.dtype), None)
    if plan is not None:
        plan.keep = None
    return plan


def _backward_with_workspace(name, value, shapes, lsi, loc, weights, dims, args, plan=None):
    """Run the *_bwd_ws_* entry point (float32 / bfloat16): host level tables + scratch (+ plan)."""
    lib = _lib.load()
    ready = plan is not None and plan.buf is not None and plan.key == _plan_key(dims, loc, weights, value.dtype)
    fn = getattr(lib, "%s_%s" % (name, _SUFFIX[value.dtype]))
    with _device_guard(value.device):
        stream = torch.cuda.current_stream(value.device).cuda_stream
        ws, sh, ls = _workspace(lib.boxattn_bwd_workspace_bytes, value, shapes, lsi, dims, stream)
        # the state buffer of this (stream, shape): the record ranges of the one-pass fill live in it from call to call
        key, state = _state_for(lib, value, dims, sh, ls, stream)
        hints = plan.hints if plan is not None else (_LOCALITY[key].last_hints if key in _LOCALITY else 0)
        rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args],
                sh.ctypes.data, ls.ctypes.data, ws.data_ptr(), ws.numel(),
                plan.buf.data_ptr() if ready else 0, plan.buf.numel() if ready else 0,
                state.data_ptr(), state.numel(), hints | _fresh_hint(state), stream)
    if rc != 0:
        _STATE.pop(key, None)
        raise RuntimeError("%s_%s failed with hipError %d" % (name, _SUFFIX[value.dtype], rc))
    state._boxattn_fresh = False


def _call(name, value, *args):
    fn = getattr(_lib.load(), "%s_%s" % (name, _SUFFIX[value.dtype]))
    with torch.cuda.device(value.device):
        stream = torch.cuda.current_stream(value.device).cuda_stream
        rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], stream)
    if rc != 0:
        raise RuntimeError("%s_%s failed with hipError %d" % (name, _SUFFIX[value.dtype], rc))


def box_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                     im2col_step):
    """-> output (B, Lq, H*C).  Reference: box_attn_cuda_forward (box_attn.cu:15-71)."""
    dims, loc, (attn,), _ = _prepare(value, spatial_shapes, level_start_index, sampling_loc,
                                     [attn_weight])
    B, S, H, C, L, Lq, P = dims
    _chunk_assert(B, im2col_step)
    out = torch.empty((B, Lq, H * C), dtype=value.dtype, device=value.device)
    if value.dtype == torch.float64:
        _call("boxattn_fwd", value, value, spatial_shapes, level_start_index, loc, attn, *dims, out)
    elif _wants_plan(value, sampling_loc, attn_weight):      # a backward will follow: build and park its plan
        _park(value, _forward_train("boxattn_fwd_train", value, spatial_shapes, level_start_index, loc,
                                    (attn,), dims,
                                    [value, spatial_shapes, level_start_index, loc, attn, *dims, out]), (loc, attn))
    else:       # host copies of the level tables let the library recognise the encoder case
        sh, ls = _host_table(spatial_shapes), _host_table(level_start_index)
        _call("boxattn_fwd_hl", value, value, spatial_shapes, level_start_index, loc, attn, *dims,
              out, sh.ctypes.data, ls.ctypes.data)
    return out


def box_attn_forward_train(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    """Forward for training: -> (output, plan).  ``plan`` (or None) goes to
    ``box_attn_backward(..., plan=plan)``: the forward's launch already counted and scanned the
    destination bins of the sample points for the backward.  Not part of the reference API."""
    dims, loc, (attn,), _ = _prepare(value, spatial_shapes, level_start_index, sampling_loc,
                                     [attn_weight])
    B, S, H, C, L, Lq, P = dims
    _chunk_assert(B, im2col_step)
    out = torch.empty((B, Lq, H * C), dtype=value.dtype, device=value.device)
    if value.dtype == torch.float64:
        _call("boxattn_fwd", value, value, spatial_shapes, level_start_index, loc, attn, *dims, out)
        return out, None
    plan = _forward_train("boxattn_fwd_train", value, spatial_shapes, level_start_index, loc,
                          (attn,), dims,
                          [value, spatial_shapes, level_start_index, loc, attn, *dims, out])
    return out, plan


def box_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                      grad_output, im2col_step, plan=None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (box_attn.cu:74-135).
    For bfloat16 ``value`` the location / weight gradients are float32."""
    dims, loc, (attn,), cdt = _prepare(value, spatial_shapes, level_start_index, sampling_loc,
                                       [attn_weight], [("grad_output", grad_output)])
    B, S, H, C, L, Lq, P = dims
    _chunk_assert(B, im2col_step)
    if grad_output.numel() != B * Lq * H * C:
        raise RuntimeError("grad_output must have B*Lq*H*C elements")
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty(sampling_loc.shape, dtype=cdt, device=value.device)
    grad_attn = torch.empty(attn_weight.shape, dtype=cdt, device=value.device)
    args = [value, spatial_shapes, level_start_index, loc, attn, grad_output, *dims, grad_value,
            grad_loc, grad_attn]
    if value.dtype == torch.float64:
        _call("boxattn_bwd", value, *args)
    else:
        if plan is None:
            plan = _parked(value, dims, loc, (attn,))
        _backward_with_workspace("boxattn_bwd_ws", value, spatial_shapes, level_start_index, loc,
                                 (attn,), dims, args, plan)
    return [grad_value, grad_loc, grad_attn]


def instance_attn_forward(value, spatial_shapes, level_start_index, sampling_loc,
                          spatial_attn_weight, level_attn_weight, im2col_step):
    """-> [output (B,Lq,H*C), mask_output (B,Lq,P,H*C)] (instance_attn.cu:15-82)."""
    dims, loc, (sw, lw), _ = _prepare(value, spatial_shapes, level_start_index, sampling_loc,
                                      [spatial_attn_weight, level_attn_weight])
    B, S, H, C, L, Lq, P = dims
    _chunk_assert(B, im2col_step)
    out = torch.empty((B, Lq, H * C), dtype=value.dtype, device=value.device)
    mask = torch.empty((B, Lq, P, H * C), dtype=value.dtype, device=value.device)
    if value.dtype != torch.float64 and _wants_plan(value, sampling_loc, spatial_attn_weight, level_attn_weight):
        _park(value, _forward_train("instattn_fwd_train", value, spatial_shapes, level_start_index, loc,
                                    (sw, lw), dims,
                                    [value, spatial_shapes, level_start_index, loc, sw, lw, *dims, out, mask]),
              (loc, sw, lw))
    else:
        _call("instattn_fwd", value, value, spatial_shapes, level_start_index, loc, sw, lw, *dims,
              out, mask)
    return [out, mask]


def instance_attn_forward_train(value, spatial_shapes, level_start_index, sampling_loc,
                                spatial_attn_weight, level_attn_weight, im2col_step):
    """-> ([output, mask_output], plan); see box_attn_forward_train."""
    dims, loc, (sw, lw), _ = _prepare(value, spatial_shapes, level_start_index, sampling_loc,
                                      [spatial_attn_weight, level_attn_weight])
    B, S, H, C, L, Lq, P = dims
    _chunk_assert(B, im2col_step)
    out = torch.empty((B, Lq, H * C), dtype=value.dtype, device=value.device)
    mask = torch.empty((B, Lq, P, H * C), dtype=value.dtype, device=value.device)
    args = [value, spatial_shapes, level_start_index, loc, sw, lw, *dims, out, mask]
    if value.dtype == torch.float64:
        _call("instattn_fwd", value, *args)
        return [out, mask], None
    plan = _forward_train("instattn_fwd_train", value, spatial_shapes, level_start_index, loc,
                          (sw, lw), dims, args)
    return [out, mask], plan


def instance_attn_backward(value, spatial_shapes, level_start_index, sampling_loc,
                           spatial_attn_weight, level_attn_weight, grad_output,
                           grad_mask_output, im2col_step, plan=None):
    """-> [grad_value, grad_sampling_loc, grad_spatial_attn_weight, grad_level_attn_weight]
    (instance_attn.cu:85-157)."""
    dims, loc, (sw, lw), cdt = _prepare(
        value, spatial_shapes, level_start_index, sampling_loc,
        [spatial_attn_weight, level_attn_weight],
        [("grad_output", grad_output), ("grad_mask_output", grad_mask_output)])
    B, S, H, C, L, Lq, P = dims
    _chunk_assert(B, im2col_step)
    if grad_output.numel() != B * Lq * H * C or grad_mask_output.numel() != B * Lq * P * H * C:
        raise RuntimeError("grad_output / grad_mask_output have the wrong number of elements")
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty(sampling_loc.shape, dtype=cdt, device=value.device)
    grad_sw = torch.empty(spatial_attn_weight.shape, dtype=cdt, device=value.device)
    grad_lw = torch.empty(level_attn_weight.shape, dtype=cdt, device=value.device)
    args = [value, spatial_shapes, level_start_index, loc, sw, lw, grad_output, grad_mask_output,
            *dims, grad_value, grad_loc, grad_sw, grad_lw]
    if value.dtype == torch.float64:
        _call("instattn_bwd", value, *args)
    else:
        if plan is None:
            plan = _parked(value, dims, loc, (sw, lw))
        _backward_with_workspace("instattn_bwd_ws", value, spatial_shapes, level_start_index, loc,
                                 (sw, lw), dims, args, plan)
    return [grad_value, grad_loc, grad_sw, grad_lw]


# ---------------------------------------------------------------------------------------
# reference windows + box offsets -> sampling grid (opt-in, beyond the reference's native
# module)
# ---------------------------------------------------------------------------------------
def _grid_args(ref_windows, offsets, kernel_indices, valid_ratios, angle_mode):
    _check(ref_windows, "ref_windows")
    _check(offsets, "offsets")
    _check(kernel_indices, "kernel_indices")
    for t in (ref_windows, offsets, kernel_indices):
        if t.dtype != torch.float32:
            raise RuntimeError("box_grid: float32 tensors expected")
    if offsets.dim() != 5 or ref_windows.dim() not in (3, 4):
        raise RuntimeError("expected offsets (B,Lq,H,L,V) and ref_windows (B,Lq,D) or (B,Lq,H,D)")
    if kernel_indices.dim() != 2 or kernel_indices.size(1) != 2:
        raise RuntimeError("expected kernel_indices (P,2)")
    B, Lq, H, L, V = offsets.shape
    per_head = ref_windows.dim() == 4
    D = ref_windows.size(-1)
    if tuple(ref_windows.shape[:2]) != (B, Lq) or (per_head and ref_windows.size(2) != H):
        raise RuntimeError("ref_windows do not match offsets")
    if angle_mode not in (0, 1, 2) or V != (5 if angle_mode == 1 else 4) or \
            D < (5 if angle_mode else 4):
        raise RuntimeError("box_grid: offsets / ref_windows do not fit angle_mode %r" % angle_mode)
    if valid_ratios is not None:
        _check(valid_ratios, "valid_ratios")
        if valid_ratios.dtype != torch.float32 or valid_ratios.numel() != B * L * 2:
            raise RuntimeError("expected valid_ratios with B*L*2 float32 elements")
    return (B, Lq, H, L, kernel_indices.size(0)), (D, int(per_head), V, angle_mode)


def _grid_call(name, anchor, *args):
    fn = getattr(_lib.load(), name)
    with torch.cuda.device(anchor.device):
        stream = torch.cuda.current_stream(anchor.device).cuda_stream
        rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], stream)
    if rc != 0:
        raise RuntimeError("%s failed with hipError %d" % (name, rc))


def box_grid_forward(ref_windows, offsets, kernel_indices, valid_ratios=None, angle_mode=0):
    """Everything of the modules' ``_where_to_attend`` after the offset projection
    (box_attention.py:63-81, 304-338) in one kernel:
    ``box = ref[:4] + offsets[:4]/8 * (w,h,w,h)_ref``; rotation ``theta`` none (angle_mode 0),
    ``(ref[4] + offsets[4]/16) * 2 pi`` (1) or ``ref[4]`` (2);
    ``grid = (c + R(theta)(kernel_indices * relu(size))) * valid_ratios`` -> (B,Lq,H,L,P,2)."""
    dims, (D, per_head, V, mode) = _grid_args(ref_windows, offsets, kernel_indices, valid_ratios,
                                              angle_mode)
    B, Lq, H, L, P = dims
    grid = torch.empty((B, Lq, H, L, P, 2), dtype=torch.float32, device=offsets.device)
    _grid_call("boxattn_grid_fwd_f32", offsets, ref_windows, D, per_head, offsets, V, mode,
               kernel_indices, valid_ratios if valid_ratios is not None else 0, *dims, grid)
    return grid


def box_grid_backward(ref_windows, offsets, kernel_indices, valid_ratios, angle_mode, grad_grid,
                      need_ref_grad=False):
    """-> (grad_offsets (B,Lq,H,L,V), grad_ref_rows (B,Lq,H,L,5) or None)."""
    dims, (D, per_head, V, mode) = _grid_args(ref_windows, offsets, kernel_indices, valid_ratios,
                                              angle_mode)
    B, Lq, H, L, P = dims
    _check(grad_grid, "grad_grid")
    if grad_grid.dtype != torch.float32 or grad_grid.numel() != B * Lq * H * L * P * 2:
        raise RuntimeError("expected grad_grid (B,Lq,H,L,P,2) float32")
    grad_offsets = torch.empty_like(offsets)
    grad_rows = torch.empty((B, Lq, H, L, 5), dtype=torch.float32, device=offsets.device) \
        if need_ref_grad else None
    _grid_call("boxattn_grid_bwd_f32", offsets, ref_windows, D, per_head, offsets, V, mode,
               kernel_indices, valid_ratios if valid_ratios is not None else 0, grad_grid, *dims,
               grad_offsets, grad_rows if grad_rows is not None else 0)
    return grad_offsets, grad_rows


# ---------------------------------------------------------------------------------------
# pointwise work around the operator (opt-in)
# ---------------------------------------------------------------------------------------
def _pw_suffix(t, what):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.bfloat16:
        return "bf16"
    raise RuntimeError("%s: float32 or bfloat16 expected, got %s" % (what, t.dtype))


def softmax_forward(logits):
    """softmax over the last axis (the L*P logits of a (query, head); at most 64) of float32 /
    bfloat16 ``logits`` -> float32 weights, one pass (box_attention.py:227-229)."""
    _check(logits, "logits")
    n = logits.size(-1)
    rows = logits.numel() // max(n, 1)
    attn = torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
    _grid_call("boxattn_softmax_fwd_" + _pw_suffix(logits, "logits"), logits, logits, rows, n, attn)
    return attn


def softmax_backward(attn, grad_attn, dtype):
    """-> grad_logits = attn * (grad_attn - sum(attn * grad_attn)) in ``dtype``."""
    _check(attn, "attn")
    _check(grad_attn, "grad_attn")
    if attn.dtype != torch.float32 or grad_attn.dtype != torch.float32 or attn.shape != grad_attn.shape:
        raise RuntimeError("softmax_backward: float32 attn / grad_attn of one shape expected")
    out = torch.empty(attn.shape, dtype=dtype, device=attn.device)
    n = attn.size(-1)
    _grid_call("boxattn_softmax_bwd_" + _pw_suffix(out, "grad_logits"), attn, attn, grad_attn,
               attn.numel() // max(n, 1), n, out)
    return out


def value_mask_cast(value, v_mask):
    """(B, S, d) float32 / bfloat16 -> bfloat16 with the rows of padded pixels (``v_mask`` (B, S)
    bool, or None) zeroed, one pass (box_attention.py:223-225 + the op's bf16 conversion)."""
    _check(value, "value")
    d = value.size(-1)
    rows = value.numel() // max(d, 1)
    mask = 0
    if v_mask is not None:
        _check(v_mask, "v_mask")
        if v_mask.dtype != torch.bool or v_mask.numel() != rows:
            raise RuntimeError("v_mask must be a bool tensor with one entry per value row")
        mask = v_mask
    out = torch.empty(value.shape, dtype=torch.bfloat16, device=value.device)
    _grid_call("boxattn_value_prep_" + _pw_suffix(value, "value"), value, value, mask, rows, d, out)
    return out
