"""``spconv.utils`` namespace: the voxel generator the reference's data pipeline instantiates
(pcdet/datasets/processor/data_processor.py:14-59: ``Point2VoxelCPU3d(vsize_xyz, coors_range_xyz, num_point_features,
max_num_points_per_voxel, max_num_voxels).point_to_voxel(points)``).

Here the voxeliser runs on the GPU (vc_voxelize_mean) with MeanVFE fused: it returns the per-voxel MEAN features.  To stay
a drop-in for the reference's ``VoxelGeneratorWrapper`` + ``MeanVFE`` pair the result is shaped (M, 1, F) with
``num_points == 1`` so that the unmodified MeanVFE (sum / 1, max over one slot) is the identity on it; the true
per-voxel point count is returned by ``point_to_voxel_mean``.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


class Point2VoxelGPU3d:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels,
                 vfe_max_last: bool = True, device="cuda"):
        self.vsize = [float(v) for v in vsize_xyz]
        self.range = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_points = int(max_num_points_per_voxel)
        self.max_voxels = int(max_num_voxels)
        self.vfe_max_last = bool(vfe_max_last)
        self.device = torch.device(device)

    def point_to_voxel_mean(self, points):
        """points (P, F) numpy or tensor -> (features (M, F) f32, coords (M, 3) i32 [z, y, x], num_points (M,) i32), on device."""
        if isinstance(points, np.ndarray):
            points = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
        points = points.to(self.device, dtype=torch.float32).contiguous()
        assert points.shape[1] == self.num_point_features
        return ops.get_backend().voxelize_mean(points, self.range, self.vsize, self.max_points, self.max_voxels,
                                               self.vfe_max_last)

    def point_to_voxel(self, points):
        f, c, n = self.point_to_voxel_mean(points)
        return f.unsqueeze(1), c, torch.ones_like(n)

    __call__ = point_to_voxel


# names the reference probes for, in order (data_processor.py:16-25)
Point2VoxelCPU3d = Point2VoxelGPU3d
