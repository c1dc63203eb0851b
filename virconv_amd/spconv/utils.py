"""``spconv.utils`` namespace: the voxel generator the reference's data pipeline instantiates
(pcdet/datasets/processor/data_processor.py:14-59: ``Point2VoxelCPU3d(vsize_xyz, coors_range_xyz, num_point_features,
max_num_points_per_voxel, max_num_voxels).point_to_voxel(tv.from_numpy(points))`` followed by ``.numpy()`` on the three
results).

Here the voxeliser runs on the GPU.  ``point_to_voxel`` keeps the reference's exact return protocol -- zero-padded voxels
``(M, max_points, F)``, coordinates ``(M, 3) [z, y, x]`` and the per-voxel point count ``(M,)``, each as a ``tv.Tensor``
(:mod:`virconv_amd.spconv.tensorview`) whose ``.numpy()`` is the device-to-host copy -- so the reference's unmodified
``VoxelGeneratorWrapper.generate`` + ``MeanVFE`` (mean_vfe.py:39-49) work on it.  ``point_to_voxel_mean`` is the fused path
this package's own front-end uses (vc_voxelize_mean: MeanVFE folded into the voxeliser, nothing leaves the GPU).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from . import tensorview as tv


class Point2VoxelGPU3d:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels,
                 vfe_max_last: bool = True, device=None):
        self.vsize = [float(v) for v in vsize_xyz]
        self.range = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_points = int(max_num_points_per_voxel)
        self.max_voxels = int(max_num_voxels)
        self.vfe_max_last = bool(vfe_max_last)
        # the reference's wrapper cannot pass a device (data_processor.py:35-41): default to the GPU.  Without one the tensors
        # stay on the host, where only an injected test backend accepts them (HipBackend raises: there is no CPU path)
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))

    def _points(self, points) -> torch.Tensor:
        if isinstance(points, tv.Tensor):
            points = points.torch()
        if isinstance(points, np.ndarray):
            points = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
        points = points.to(self.device, dtype=torch.float32).contiguous()
        assert points.dim() == 2 and points.shape[1] == self.num_point_features, \
            f"points must be (P, {self.num_point_features}), got {tuple(points.shape)}"
        return points

    def point_to_voxel_mean(self, points):
        """points (P, F) -> (mean features (M, F) f32, coords (M, 3) i32 [z, y, x], num_points (M,) i32), on device."""
        return ops.get_backend().voxelize_mean(self._points(points), self.range, self.vsize, self.max_points,
                                               self.max_voxels, self.vfe_max_last)

    def point_to_voxel_torch(self, points):
        """-> (voxels (M, max_points, F), coords (M, 3), num_points (M,)) as torch tensors on the voxeliser's device."""
        return ops.get_backend().voxelize(self._points(points), self.range, self.vsize, self.max_points, self.max_voxels)

    def point_to_voxel(self, points):
        """spconv-2.x protocol: three tv.Tensor results, ``.numpy()`` copies them to the host (data_processor.py:53-58)."""
        v, c, n = self.point_to_voxel_torch(points)
        return tv.Tensor(v), tv.Tensor(c), tv.Tensor(n)

    __call__ = point_to_voxel


# the name the reference probes for (data_processor.py:16-25; VoxelGeneratorV2 / VoxelGenerator are spconv-1.x names and
# deliberately absent so that the wrapper selects the spconv-2 protocol)
Point2VoxelCPU3d = Point2VoxelGPU3d
