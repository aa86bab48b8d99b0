"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what
include/boxattn.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "boxattn.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:boxattn|instattn)_\w+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from boxer_amd import _lib
    _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_declares_the_four_entry_points_per_dtype():
    names = declared_functions()
    for stem in ("boxattn_fwd", "boxattn_bwd", "instattn_fwd", "instattn_bwd"):
        for suf in ("f32", "f64", "bf16"):
            assert "%s_%s" % (stem, suf) in names
    assert "boxattn_abi_version" in names and "boxattn_set_variant" in names


def test_library_exports_every_declared_symbol(lib):
    from boxer_amd import _lib
    for name in declared_functions():
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(_lib.EXPORTS) == declared_functions()


def test_library_metadata(lib):
    from boxer_amd import _lib
    declared = int(re.search(r"#define BOXATTN_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert _lib.load().boxattn_abi_version() == declared == _lib.ABI_VERSION
    info = _lib.build_info()
    assert "gfx950" in info
    assert _lib.set_variant(1) == 0 and _lib.set_variant(0) == 1


def test_library_contains_gfx950_code_object():
    """The fat binary embedded in the .so carries a gfx950 code object (bundle entry id)."""
    from boxer_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_no_torch_types_in_the_abi():
    code = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)      # strip comments
    for banned in ("torch", "at::", "Tensor", "c10", "#include <ATen"):
        assert banned not in code, banned
    includes = re.findall(r"#include\s+[<\"]([^>\"]+)", code)
    assert sorted(includes) == ["stddef.h", "stdint.h"]


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "boxer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "boxattn_oracle" not in src and "torch_fallback" not in src, f


def test_no_packed_float32_next_to_mfma(lib):
    """ISA guard: packed float32 VALU instructions issued while an MFMA of
    the same wave completes returned wrong values on MI355X; no kernel of the shipped code objects that issues
    a v_mfma may contain a v_pk_{mul,add,fma}_f32.  (Guards the -fno-slp-vectorize rule of boxattn_dense.hip
    against a compiler update, a new kernel in the wrong translation unit, or a changed flag.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_guard
    finally:
        sys.path.pop(0)
    from boxer_amd import _lib
    n, n_mfma, offenders = isa_guard.scan(_lib.LIB_PATH)
    assert n > 50 and n_mfma >= 4, (n, n_mfma)         # the disassembly really covered the library
    assert not offenders, offenders


def test_no_kernel_spills_to_scratch(lib):
    """Code-object guard: a spilled value is a scratch store / load
    pair with a full memory wait -- in the round-4 point-gradient kernel four spilled loads were four serial HBM
    round trips (51 -> 65 us).  No kernel of the shipped library may use scratch."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    from boxer_amd import _lib
    rows = kernel_resources.kernels(_lib.LIB_PATH)
    # every translation unit's code object (the library's fat binary holds one bundle per object: the window-staged
    # kernels live in the third)
    assert len(rows) > 150 and any("fwd_dense_kernel" in r[0] for r in rows) and \
        any("softmax" in r[0] for r in rows), len(rows)
    spilling = [(name, scratch) for name, _v, _s, _lds, scratch in rows if int(scratch) != 0]
    assert not spilling, spilling


def test_build_flags_are_part_of_the_staleness_check(tmp_path, monkeypatch):
    """A change of compiler flags must rebuild like a change of sources (the object cache used to be keyed on
    source mtimes only)."""
    from boxer_amd import _lib
    tag = os.path.basename(_lib.LIB_PATH)
    assert not _lib._flags_changed(tag)                 # the library of this checkout was built with today's flags
    monkeypatch.setattr(_lib, "HIPCC_FLAGS", _lib.HIPCC_FLAGS + ["-DSOMETHING_ELSE"])
    assert _lib._flags_changed(tag) and _lib.needs_build()
