"""SparseConvTensor: the container type of the reference's operator API.

Mirrors what the reference touches on spconv's SparseConvTensor (spconv_backbone.py:217-222,639-644,147;
spconv_utils.py:58-64; height_compression.py:29): ``features, indices, spatial_shape, batch_size, indice_dict,
dense(), replace_feature()``; ``indices`` is assignable.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops


class SparseConvTensor:
    def __init__(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int,
                 grid=None, voxel_num=None, indice_dict: Optional[Dict] = None, benchmark: bool = False):
        assert features.dim() == 2, "features must be (N, C)"
        assert indices.dim() == 2 and indices.shape[0] == features.shape[0], "indices must be (N, ndim+1)"
        assert indices.dtype == torch.int32, "indices must be int32 (the reference passes coords.int())"
        self._features = features
        self.indices = indices
        self.spatial_shape: List[int] = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        self.benchmark_record = {}

    # spconv 2.x: features is a property; assignment is tolerated (spconv 1.x style, spconv_utils.py:63)
    @property
    def features(self) -> torch.Tensor:
        return self._features

    @features.setter
    def features(self, val: torch.Tensor) -> None:
        self._features = val

    def replace_feature(self, feature: torch.Tensor) -> "SparseConvTensor":
        """New tensor object sharing indices and the rulebook cache.  Like spconv 2.x it does NOT
        check that the row count still matches ``indices`` (the reference's layer_voxel_discard relies on that:
        spconv_backbone.py:146-147 replaces the features first and assigns ``.indices`` afterwards)."""
        t = object.__new__(SparseConvTensor)
        t.__dict__.update(self.__dict__)
        t._features = feature
        return t

    @property
    def spatial_size(self) -> int:
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first: bool = True, pad=(0, 0)) -> torch.Tensor:
        """(B, C, *spatial) scatter of the active rows.  `pad` (an extension, default off): zero border of
        (pad_h, pad_w) cells around the last two axes, written by the same pass (the ZeroPad2d of the first BEV conv)."""
        d = ops.to_dense(self.features, self.indices, self.spatial_shape, self.batch_size, pad)
        if channels_first:
            return d
        nd = len(self.spatial_shape)
        return d.permute(0, *range(2, nd + 2), 1).contiguous()

    def __repr__(self) -> str:
        return (f"SparseConvTensor(N={self.features.shape[0]}, C={self.features.shape[1]}, "
                f"shape={self.spatial_shape}, bs={self.batch_size})")
