"""spconv-compatible facade over the MI355X HIP operators (SURVEY §8b "operator API").

The reference reaches all sparse-conv arithmetic through ``import spconv.pytorch as spconv``
(pcdet/utils/spconv_utils.py:33-36).  ``install()`` registers this package under the module names ``spconv``,
``spconv.pytorch``, ``spconv.pytorch.conv``, ``spconv.utils`` (and ``cumm.tensorview``) so the reference's backbone file
(pcdet/models/backbones_3d/spconv_backbone.py) runs UNMODIFIED on these operators.
"""
from __future__ import annotations

import sys

from . import conv, tensorview, utils  # noqa: F401
from .conv import (SparseConv2d, SparseConv3d, SparseConvolution, SparseInverseConv2d, SparseInverseConv3d, SubMConv2d,
                   SubMConv3d)
from .core import SparseConvTensor
from .modules import SparseModule, SparseSequential

__version__ = "2.1.22+virconv_amd"

__all__ = ["SparseConvTensor", "SparseModule", "SparseSequential", "SparseConvolution", "SubMConv2d", "SubMConv3d",
           "SparseConv2d", "SparseConv3d", "SparseInverseConv2d", "SparseInverseConv3d", "conv", "utils", "tensorview", "install"]


def install(force: bool = False) -> None:
    """Make ``import spconv`` / ``import spconv.pytorch as spconv`` resolve to this facade."""
    from . import pytorch as _pt
    if "spconv" in sys.modules and not force and sys.modules["spconv"] is not sys.modules[__name__]:
        raise RuntimeError("a different `spconv` is already imported; pass force=True to override it")
    me = sys.modules[__name__]
    sys.modules["spconv"] = me
    sys.modules["spconv.pytorch"] = _pt
    sys.modules["spconv.pytorch.conv"] = conv
    sys.modules["spconv.conv"] = conv
    sys.modules["spconv.utils"] = utils
    sys.modules["spconv.pytorch.utils"] = utils
    # cumm.tensorview: the reference's VoxelGeneratorWrapper wraps its input with tv.from_numpy and calls .numpy() on the
    # results (data_processor.py:8-11,53-58)
    import types
    from . import tensorview
    if "cumm" not in sys.modules or force or getattr(sys.modules["cumm"], "__virconv_amd_shim__", False):
        cumm = types.ModuleType("cumm")
        cumm.__virconv_amd_shim__ = True
        cumm.tensorview = tensorview
        sys.modules["cumm"] = cumm
        sys.modules["cumm.tensorview"] = tensorview
    # data_processor binds `tv` at import time inside a try/except: if it was imported before install(), hand it the shim
    dp = sys.modules.get("pcdet.datasets.processor.data_processor")
    if dp is not None and getattr(dp, "tv", None) is None:
        dp.tv = tensorview
