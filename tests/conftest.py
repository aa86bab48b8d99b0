import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_kernel_choice():
    """boxer_amd.ops picks the window-staged or the gather kernels of the encoder case from what the previous calls
    at that shape saw (ops._Locality).  Every test starts without that history: a test that compares outputs bit
    for bit must not inherit the choice an earlier test's uniformly random locations led to."""
    ops = sys.modules.get("boxer_amd.ops")
    if ops is not None:
        ops._LOCALITY.clear()
        ops._PARKED.clear()          # (plans parked by a forward whose backward a test never ran)
    yield
    # ... and no test inherits a library switch another one left set (a skip or a failed assertion between a test's
    # boxattn_set_option calls): every option and the kernel variant back to the defaults
    lib = sys.modules.get("boxer_amd._lib")
    if lib is not None and lib._lib is not None:
        for key in lib.OPTIONS.values():
            lib._lib.boxattn_set_option(key, 0)
        lib._lib.boxattn_set_variant(0)
