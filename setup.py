"""Build script: compiles the C-ABI HIP library in-tree with hipcc for gfx950.

    python setup.py build_ext --inplace     # -> boxer_amd/libboxattn_hip.so
    pip install -e . --no-build-isolation   # optional

Counterpart of the reference's setup.py:29-76 (CUDAExtension "e2edet.ops").  The product is a
plain shared library with a C ABI (include/boxattn.h), not a torch extension module: PyTorch
binds to it through ctypes (boxer_amd/ops.py), so that part of the build needs neither torch
headers nor pybind11 and works on a machine without a GPU (hipcc cross-compiles gfx950).  When
torch is importable the build also compiles boxer_amd/csrc/e2edet_ops.cpp -- the reference's
pybind11 module re-stated on the C ABI (four functions on at::Tensor, host code only).
"""
import importlib.util
import os

from setuptools import Command, find_packages, setup
from setuptools.command.build_ext import build_ext as _build_ext

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib_module():
    spec = importlib.util.spec_from_file_location(
        "_boxattn_lib", os.path.join(HERE, "boxer_amd", "_lib.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class build_hip(_build_ext):
    """`build_ext` that drives hipcc directly (no Extension objects)."""

    def run(self):
        lib = _lib_module()
        # a copy of the C-ABI header inside the package: a non-editable install can then rebuild
        # the compiled e2edet_ops module without the source checkout (boxer_amd/_lib.INCLUDE_DIR)
        import shutil
        os.makedirs(os.path.join(HERE, "boxer_amd", "include"), exist_ok=True)
        shutil.copy2(os.path.join(HERE, "include", "boxattn.h"),
                     os.path.join(HERE, "boxer_amd", "include", "boxattn.h"))
        path = lib.build(force=True, verbose=True)
        print("built", path)
        # the reference's pybind11 module on the C ABI (host compiler + torch headers); optional:
        # the package itself binds the library through ctypes
        try:
            import sys
            sys.path.insert(0, HERE)
            from boxer_amd import _ext
            print("built", _ext.build(force=True, verbose=True))
        except ImportError as e:
            print("skipping the compiled e2edet_ops module (torch not importable):", e)


setup(
    name="boxer_amd",
    version="0.1.0",
    description="MI355X-native box-attention / instance-attention operator (BoxeR drop-in)",
    packages=find_packages(include=["boxer_amd", "boxer_amd.*"]),
    package_data={"boxer_amd": ["libboxattn_hip.so", "e2edet_ops*.so", "csrc/*", "include/*.h"]},
    cmdclass={"build_ext": build_hip},
    python_requires=">=3.8",
)
