"""Build and import the compiled operator module (``csrc/e2edet_ops.cpp``): the reference's
pybind11 module ``e2edet.ops`` (vision.cpp:7-12) re-stated on the C ABI -- four functions on
``at::Tensor`` that marshal pointers into ``libboxattn_hip.so``.

The package itself binds the library through ctypes (``ops.py``); this module exists so that the
reference's Functions (``box_attention_func.py:3`` ``from e2edet import ops as _C``) can import
a drop-in ``_C`` with no Python of ours in between, and so that the stub INTEGRATION.md shows a
maintainer is compiled and tested rather than prose.  Host code only: built with the host C++
compiler against torch's headers, in-tree (``boxer_amd/e2edet_ops*.so``), linked to the HIP
library with an ``$ORIGIN`` rpath.
"""
import importlib.machinery
import importlib.util
import os
import shutil
import subprocess
import sys
import sysconfig

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
MODULE_NAME = "e2edet_ops"
SOURCE = os.path.join(HERE, "csrc", "e2edet_ops.cpp")
EXT_PATH = os.path.join(HERE, MODULE_NAME + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
FUNCTIONS = ("box_attn_forward", "box_attn_backward", "instance_attn_forward",
             "instance_attn_backward")


def _stale():
    if not os.path.exists(EXT_PATH):
        return True
    built = os.path.getmtime(EXT_PATH)
    # the header (ABI changes), the source, and the library the module links against; a dependency
    # that is not there (installed package without the sources) does not make the module stale
    deps = [SOURCE, os.path.join(_lib.INCLUDE_DIR, "boxattn.h"), _lib.LIB_PATH]
    return any(os.path.getmtime(d) > built for d in deps if os.path.exists(d))


def _compile_command():
    import torch
    from torch.utils import cpp_extension

    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("no host C++ compiler found: cannot build %s" % MODULE_NAME)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    includes = cpp_extension.include_paths() + [
        sysconfig.get_paths()["include"], _lib.INCLUDE_DIR, os.path.join(rocm, "include")]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-deprecated-declarations",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + MODULE_NAME,
           "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in includes]
    cmd += [SOURCE, "-o", EXT_PATH + ".tmp", "-L" + torch_lib, "-L" + HERE,
            "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
            "-l:" + _lib.LIB_NAME, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib]
    return cmd


def build(force=False, verbose=False):
    """Compile the module in-tree (needs the HIP library to link against); returns its path."""
    _lib.build(verbose=verbose)
    if not force and not _stale():
        return EXT_PATH
    cmd = _compile_command()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(EXT_PATH + ".tmp", EXT_PATH)
    return EXT_PATH


def load():
    """Import the compiled module (torch first: its libraries must be in the process)."""
    import torch  # noqa: F401

    if MODULE_NAME in sys.modules:
        return sys.modules[MODULE_NAME]
    if not os.path.exists(EXT_PATH):
        raise RuntimeError("%s is not built: run `python setup.py build_ext --inplace`" % EXT_PATH)
    loader = importlib.machinery.ExtensionFileLoader(MODULE_NAME, EXT_PATH)
    spec = importlib.util.spec_from_file_location(MODULE_NAME, EXT_PATH, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    # the header the module was COMPILED against vs the library that is loaded now
    if getattr(mod, "compiled_abi_version", None) != _lib.load().boxattn_abi_version():
        raise RuntimeError("%s was built against another ABI of %s: rebuild (python setup.py "
                           "build_ext --inplace)" % (MODULE_NAME, _lib.LIB_NAME))
    sys.modules[MODULE_NAME] = mod
    return mod
