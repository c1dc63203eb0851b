"""spconv-compatible facade over the MI355X HIP operators (SURVEY §8b "operator API").

The reference reaches all sparse-conv arithmetic through ``import spconv.pytorch as spconv``
(pcdet/utils/spconv_utils.py:33-36).  ``install()`` registers this package under the module names ``spconv``,
``spconv.pytorch``, ``spconv.pytorch.conv`` and ``spconv.utils`` so the reference's backbone file
(pcdet/models/backbones_3d/spconv_backbone.py) runs UNMODIFIED on these operators.
"""
from __future__ import annotations

import sys

from . import conv, utils  # noqa: F401
from .conv import (SparseConv2d, SparseConv3d, SparseConvolution, SparseInverseConv2d, SparseInverseConv3d, SubMConv2d,
                   SubMConv3d)
from .core import SparseConvTensor
from .modules import SparseModule, SparseSequential

__version__ = "2.1.22+virconv_amd"

__all__ = ["SparseConvTensor", "SparseModule", "SparseSequential", "SparseConvolution", "SubMConv2d", "SubMConv3d",
           "SparseConv2d", "SparseConv3d", "SparseInverseConv2d", "SparseInverseConv3d", "conv", "utils", "install"]


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
