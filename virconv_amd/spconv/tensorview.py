"""``cumm.tensorview`` shim: the two calls the reference's data pipeline makes on it.

pcdet/datasets/processor/data_processor.py:8-11 does ``import cumm.tensorview as tv`` and :53-58 uses exactly
``tv.from_numpy(points)`` (input of ``Point2VoxelCPU3d.point_to_voxel``) and ``.numpy()`` on the three results
("make copy with numpy(), since numpy_view() will disappear").  ``virconv_amd.spconv.install()`` registers this module as
``cumm.tensorview`` so that the reference's UNMODIFIED ``VoxelGeneratorWrapper.generate`` reaches the GPU voxeliser.

A :class:`Tensor` wraps either a numpy array (host input) or a torch tensor (device result); ``.numpy()`` is the one
device-to-host copy of a result, ``.torch()`` hands the device tensor over without a copy (for callers that stay on the GPU).
"""
from __future__ import annotations

import numpy as np
import torch


class Tensor:
    __slots__ = ("_a",)

    def __init__(self, array):
        assert isinstance(array, (np.ndarray, torch.Tensor)), type(array)
        self._a = array

    @property
    def shape(self):
        return tuple(self._a.shape)

    @property
    def ndim(self):
        return len(self._a.shape)

    @property
    def dim(self):
        return self.shape

    def __len__(self):
        return self._a.shape[0]

    def numpy(self) -> np.ndarray:
        """A fresh host copy (cumm semantics: an owning ndarray, safe after the generator is gone)."""
        if isinstance(self._a, np.ndarray):
            return self._a.copy()
        return self._a.detach().cpu().numpy().copy()

    def numpy_view(self) -> np.ndarray:
        if isinstance(self._a, np.ndarray):
            return self._a
        return self._a.detach().cpu().numpy()

    def torch(self) -> torch.Tensor:
        return self._a if torch.is_tensor(self._a) else torch.from_numpy(self._a)

    def cpu(self) -> "Tensor":
        return Tensor(self.numpy_view())

    def __repr__(self):
        where = "host" if isinstance(self._a, np.ndarray) else str(self._a.device)
        return f"tv.Tensor(shape={self.shape}, {where})"


def from_numpy(array: np.ndarray) -> Tensor:
    return Tensor(np.ascontiguousarray(array))


def zeros(shape, dtype=np.float32, device=-1) -> Tensor:
    return Tensor(np.zeros(tuple(shape), dtype=dtype))
