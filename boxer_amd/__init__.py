"""boxer_amd -- MI355X-native (gfx950 / CDNA4) box-attention operator for BoxeR-2D/3D.

Drop-in for the one native component of kienduynguyen/BoxeR:

* ``boxer_amd.ops``        <-> ``e2edet.ops``            (4 native entry points)
* ``boxer_amd.functions``  <-> ``e2edet.module.ops``     (BoxAttnFunction, InstanceAttnFunction)
* ``boxer_amd.modules``    <-> ``e2edet.module.box_attention`` (BoxAttention, InstanceAttention,
                                                               Box3dAttention)

The compute lives in a C-ABI HIP library (``include/boxattn.h``, ``boxer_amd/csrc``); see
DESIGN.md and INTEGRATION.md.  There is no CPU fallback: ops raise if the library is missing.
"""
from . import _lib, ops
from .functions import (BoxAttnBF16Function, BoxAttnFunction, BoxGridFunction,
                        InstanceAttnBF16Function, InstanceAttnFunction, LogitSoftmaxFunction,
                        ValueMaskCastFunction)
from .modules import Box3dAttention, BoxAttention, InstanceAttention

__all__ = [
    "ops", "BoxAttnFunction", "InstanceAttnFunction", "BoxAttnBF16Function",
    "InstanceAttnBF16Function", "BoxGridFunction", "LogitSoftmaxFunction", "ValueMaskCastFunction",
    "BoxAttention", "InstanceAttention", "Box3dAttention",
    "build", "build_info",
]
__version__ = "0.1.0"

build = _lib.build
build_info = _lib.build_info
