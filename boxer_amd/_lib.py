"""Loader (and in-tree builder) of the C-ABI HIP library ``libboxattn_hip.so``.

The library is the product: there is no CPU or PyTorch fallback behind it.  If it cannot be
found or loaded, every op raises -- nothing here ever imports ``oracle/``.
"""
import ctypes
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_NAME = "libboxattn_hip.so"
LIB_PATH = os.path.join(_PKG, LIB_NAME)
# the C-ABI header: the source checkout's include/ when there is one, else the copy a non-editable
# install ships inside the package (setup.py: package_data "include/*.h")
INCLUDE_DIR = os.path.join(os.path.dirname(_PKG), "include")
if not os.path.exists(os.path.join(INCLUDE_DIR, "boxattn.h")):
    INCLUDE_DIR = os.path.join(_PKG, "include")
# translation units -> extra flags.  boxattn_dense.hip (the matrix-core encoder kernels) is built
# with -fno-slp-vectorize: packed float32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32, which the
# SLP vectoriser forms from adjacent scalar operations) issued while an MFMA of the same wave is
# finishing returned wrong results in lanes 48-63 on MI355X with this compiler
SOURCES = {"boxattn_capi.hip": [], "boxattn_extras.hip": [], "boxattn_dense.hip": ["-fno-slp-vectorize"]}
HEADERS = sorted(f for f in os.listdir(_CSRC) if f.endswith(".h"))     # every kernel header
# -amdgpu-kernarg-preload-count: the first 16 dwords of a kernel's arguments (its pointers) arrive in
# SGPRs with the wave instead of through scalar loads at its top (gfx94x / gfx950); every wave of
# these kernels is short, and their prologues are a measurable part of them
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-pass-failed",
               "-mllvm", "-amdgpu-kernarg-preload-count=16"]
BUILD_DIR = os.path.join(_PKG, "_build")

_lib = None

_vp, _i = ctypes.c_void_p, ctypes.c_int
_DIMS = [_i] * 7                      # B, S, H, C, L, Lq, P
_SIGNATURES = {
    # name: argtypes  (see include/boxattn.h)
    "boxattn_fwd": [_vp] * 5 + _DIMS + [_vp, _vp],
    "boxattn_bwd": [_vp] * 6 + _DIMS + [_vp] * 3 + [_vp],
    "instattn_fwd": [_vp] * 6 + _DIMS + [_vp] * 2 + [_vp],
    "instattn_bwd": [_vp] * 8 + _DIMS + [_vp] * 4 + [_vp],
}
_WS_SIGNATURES = {
    # plain backward args + shapes_host, lsi_host, workspace, workspace_bytes, plan, plan_bytes, state, state_bytes,
    # hints, stream
    "boxattn_bwd_ws": [_vp] * 6 + _DIMS + [_vp] * 3 + [_vp, _vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t,
                                                       _vp, ctypes.c_size_t, _i, _vp],
    "instattn_bwd_ws": [_vp] * 8 + _DIMS + [_vp] * 4 + [_vp, _vp, _vp, ctypes.c_size_t, _vp,
                                                        ctypes.c_size_t, _vp, ctypes.c_size_t, _i, _vp],
    # forward args + shapes_host, lsi_host, plan, plan_bytes, state, state_bytes, hints, int *plan_built, stream
    "boxattn_fwd_train": [_vp] * 5 + _DIMS + [_vp] + [_vp, _vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t,
                                                      _i, _vp, _vp],
    "instattn_fwd_train": [_vp] * 6 + _DIMS + [_vp] * 2 + [_vp, _vp, _vp, ctypes.c_size_t, _vp,
                                                          ctypes.c_size_t, _i, _vp, _vp],
}
_HL_SIGNATURES = {
    # forward args + shapes_host, lsi_host, stream
    "boxattn_fwd_hl": [_vp] * 5 + _DIMS + [_vp] + [_vp, _vp, _vp],
}
HINT_NOT_LOCAL = 1          # BOXATTN_HINT_NOT_LOCAL
HINT_FRESH_STATE = 2        # BOXATTN_HINT_FRESH_STATE
_ll = ctypes.c_longlong
_POINTWISE_SIGNATURES = {
    "boxattn_softmax_fwd_f32": [_vp, _ll, _i, _vp, _vp],
    "boxattn_softmax_fwd_bf16": [_vp, _ll, _i, _vp, _vp],
    "boxattn_softmax_bwd_f32": [_vp, _vp, _ll, _i, _vp, _vp],
    "boxattn_softmax_bwd_bf16": [_vp, _vp, _ll, _i, _vp, _vp],
    "boxattn_value_prep_f32": [_vp, _vp, _ll, _i, _vp, _vp],
    "boxattn_value_prep_bf16": [_vp, _vp, _ll, _i, _vp, _vp],
}
_GRID_SIGNATURES = {
    # ref, ref_dim, ref_per_head, offsets, V, angle_mode, kernel_idx, valid_ratios, [grad_grid,]
    # B, Lq, H, L, P, outputs..., stream
    "boxattn_grid_fwd_f32": [_vp, _i, _i, _vp, _i, _i, _vp, _vp] + [_i] * 5 + [_vp, _vp],
    "boxattn_grid_bwd_f32": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp] + [_i] * 5 + [_vp, _vp, _vp],
}
EXPORTS = ["boxattn_abi_version", "boxattn_build_info", "boxattn_set_variant", "boxattn_set_option",
           "boxattn_options_epoch",
           "boxattn_set_debug_buffer",
           "boxattn_fwd_hl_f32", "boxattn_fwd_hl_bf16", *sorted(_POINTWISE_SIGNATURES),
           "boxattn_profile_begin", "boxattn_profile_end", "boxattn_bwd_workspace_bytes",
           "boxattn_plan_bytes", "boxattn_state_bytes",
           "boxattn_grid_fwd_f32", "boxattn_grid_bwd_f32"] + [
    "%s_%s" % (stem, suf) for stem in _SIGNATURES for suf in ("f32", "f64", "bf16")] + [
    "%s_%s" % (stem, suf) for stem in _WS_SIGNATURES for suf in ("f32", "bf16")]
ABI_VERSION = 8


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build %s" % LIB_NAME)


def _deps():
    deps = [os.path.join(_CSRC, f) for f in list(SOURCES) + HEADERS]
    deps.append(os.path.join(INCLUDE_DIR, "boxattn.h"))
    return [d for d in deps if os.path.exists(d)]


def _flags_stamp(extra_flags=()):
    """The command-line flags of every translation unit: a change of flags rebuilds like a change of
    sources (the -fno-slp-vectorize rule of boxattn_dense.hip is a correctness matter)."""
    return repr((HIPCC_FLAGS, sorted(SOURCES.items()), list(extra_flags)))


def _stamp_path(tag):
    return os.path.join(BUILD_DIR, tag + ".flags")


def _flags_changed(tag, extra_flags=()):
    try:
        with open(_stamp_path(tag)) as fh:
            return fh.read() != _flags_stamp(extra_flags)
    except OSError:
        return True


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    # (a shipped library without a build directory is taken as it is)
    if os.path.isdir(BUILD_DIR) and _flags_changed(os.path.basename(LIB_PATH)):
        return True
    return any(os.path.getmtime(s) > built for s in _deps())


def build(force=False, verbose=False, extra_flags=(), out_path=None):
    """Cross-compile the HIP library for gfx950 in-tree (works without a GPU): one object per
    translation unit (compiled in parallel), then one shared library.  ``extra_flags`` /
    ``out_path``: tuning builds next to the product (tools/build_variants.sh)."""
    variant = out_path is not None
    out_path = out_path or LIB_PATH
    if not (force or variant or needs_build()):
        return out_path
    tag = os.path.basename(out_path)
    os.makedirs(BUILD_DIR, exist_ok=True)
    newest = max(os.path.getmtime(d) for d in _deps())
    stale_flags = _flags_changed(tag, extra_flags)
    procs, objs = [], []
    for src, flags in SOURCES.items():
        obj = os.path.join(BUILD_DIR, "%s.%s.o" % (tag, src))
        objs.append(obj)
        if not (force or variant or stale_flags) and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            continue
        cmd = [hipcc_path()] + HIPCC_FLAGS + list(flags) + list(extra_flags) + [
            "-c", os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_path + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out_path + ".tmp", out_path)
    with open(_stamp_path(tag), "w") as fh:
        fh.write(_flags_stamp(extra_flags))
    if not variant:
        global _lib
        _lib = None
    return out_path


def load():
    """Return the ctypes handle; raises RuntimeError if the HIP library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # BOXATTN_HIP_LIB: load another build of the same library (kernel tuning experiments,
    # tools/build_variants.sh); it must export the same ABI.
    path = os.environ.get("BOXATTN_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing. Build it with `python setup.py build_ext --inplace` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). boxer_amd has no CPU "
            "fallback by design." % path)
    lib = ctypes.CDLL(path)
    lib.boxattn_abi_version.restype = _i
    lib.boxattn_build_info.restype = ctypes.c_char_p
    lib.boxattn_set_variant.argtypes = [_i]
    lib.boxattn_set_variant.restype = _i
    for stem, args in _SIGNATURES.items():
        for suf in ("f32", "f64", "bf16"):
            fn = getattr(lib, "%s_%s" % (stem, suf))
            extra = [_vp] if (suf == "bf16" and stem.endswith("bwd")) else []
            fn.argtypes = args[:-1] + extra + args[-1:]        # ..., [grad_value_ws], stream
            fn.restype = _i
    for stem, args in _WS_SIGNATURES.items():
        for suf in ("f32", "bf16"):
            fn = getattr(lib, "%s_%s" % (stem, suf))
            fn.argtypes = args
            fn.restype = _i
    for stem, args in _HL_SIGNATURES.items():
        for suf in ("f32", "bf16"):
            fn = getattr(lib, "%s_%s" % (stem, suf))
            fn.argtypes = args
            fn.restype = _i
    lib.boxattn_set_option.argtypes = [_i, _i]
    lib.boxattn_set_option.restype = _i
    lib.boxattn_options_epoch.restype = _i
    for name, args in list(_GRID_SIGNATURES.items()) + list(_POINTWISE_SIGNATURES.items()):
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = _i
    lib.boxattn_state_bytes.argtypes = [_i] * 7 + [_vp, _vp]
    lib.boxattn_state_bytes.restype = ctypes.c_size_t
    for name in ("boxattn_bwd_workspace_bytes", "boxattn_plan_bytes"):
        getattr(lib, name).argtypes = [_i] * 8 + [_vp, _vp]
        getattr(lib, name).restype = ctypes.c_size_t
    if lib.boxattn_abi_version() != ABI_VERSION:
        raise RuntimeError("ABI version mismatch in %s" % path)
    _lib = lib
    return lib


def build_info():
    return load().boxattn_build_info().decode()


def options_epoch():
    """boxattn_options_epoch(): bumped by every set_variant / set_option call (ops.py caches the size queries
    per epoch: the workspace layouts depend on some switches)."""
    return load().boxattn_options_epoch()


def set_variant(v):
    """0 = auto, 1 = generic kernels only, 2 = fast atomic kernels only, 3 = auto with the binned
    backward required (2 and 3 return an error when the shape is not eligible)."""
    return load().boxattn_set_variant(int(v))


OPTIONS = {"bin_chunk": 10, "dense": 11, "riders": 15, "acc_f32": 19, "ride_shift": 20}


def set_option(name, value):
    """Tuning knobs for A/B runs (include/boxattn.h: boxattn_set_option); returns the old value."""
    old = load().boxattn_set_option(OPTIONS[name], int(value))
    if old < 0:
        raise KeyError(name)
    return old


def profile_begin():
    """Start bracketing the library's main kernels with hipEvents (bench.py)."""
    load().boxattn_profile_begin()


PROFILE_SLOTS = ("fwd", "bwd_points", "bwd_accumulate", "bwd_binning", "bwd_combine", "bwd_prep")


def profile_end():
    """-> {slot: {"ms": average kernel duration, "launches": n}} for the timing slots."""
    ms = (ctypes.c_double * len(PROFILE_SLOTS))()
    n = (ctypes.c_int * len(PROFILE_SLOTS))()
    load().boxattn_profile_end(ms, n)
    return {name: {"ms": (ms[i] / n[i] if n[i] else None), "launches": int(n[i])}
            for i, name in enumerate(PROFILE_SLOTS)}


if __name__ == "__main__":      # python -m boxer_amd._lib [--out PATH] [extra hipcc flags ...]
    import sys
    args = sys.argv[1:]
    out = None
    if args[:1] == ["--out"]:
        out, args = args[1], args[2:]
    print(build(force=True, verbose=True, extra_flags=args, out_path=out))
