"""``spconv.pytorch`` namespace of the facade (the reference does ``import spconv.pytorch as spconv``)."""
from .. import conv, utils  # noqa: F401
from ..conv import (SparseConv2d, SparseConv3d, SparseConvolution, SparseInverseConv2d, SparseInverseConv3d,  # noqa: F401
                    SubMConv2d, SubMConv3d)
from ..core import SparseConvTensor  # noqa: F401
from ..modules import SparseModule, SparseSequential  # noqa: F401
