"""Host-side data front-end of the hot path (the part the reference runs in numpy inside its Dataset):

  input_point_discard   StVD input discard, bin-based (pcdet/datasets/dataset.py:120-189) -- numpy, like the reference
  prepare_frame         LATER_FUSION=False fusion: discard virtual points, concat LiDAR-first
                        (dataset.py:270-294, data_processor.py:152-155)
  voxelize_batch        GPU voxeliser + fused MeanVFE per frame (vc_voxelize_mean), collated with the batch index
                        prepended (dataset.py:315-367 collate_batch)
  input_point_discard_device / prepare_frame_device
                        the same discard on a torch tensor that is already on the GPU (raw points in,
                        voxel features out, nothing on the host but one 10-int histogram read) -- device-agnostic torch ops
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import ops


def partition(points: np.ndarray, num: int = 10, max_dis: float = 60, rate: float = 0.2):
    """Distance bins far->near with the running retain test (dataset.py:120-170)."""
    parts: List[np.ndarray] = []
    inter = max_dis / num
    total = points.shape[0]
    acc, position, distant_acc = 0, num - 1, 0
    for j in range(num):
        i = num - j - 1
        lo = points[:, 0] >= inter * i
        mask = lo if i == num - 1 else (lo & (points[:, 0] < inter * (i + 1)))
        cur = points[mask]
        acc += cur.shape[0]
        if (acc + i * cur.shape[0]) / total < rate:
            position, distant_acc = i, acc
        parts.append(cur)
    return parts, max(position, 0), distant_acc


def input_point_discard(points: np.ndarray, bin_num: int = 2, rate: float = 0.8,
                        permutation: Callable[[int], np.ndarray] = np.random.permutation) -> np.ndarray:
    """Bin-balanced random drop of virtual points (dataset.py:172-189); `permutation` injects the RNG."""
    retain = 1 - rate
    parts, pos, distant_acc = partition(points, num=bin_num, rate=retain)
    out_n = int(points.shape[0] * retain)
    per_bin = int((out_n - distant_acc) / (pos + 0.0001))
    for i in range(len(parts) - pos, len(parts)):
        if parts[i].shape[0] > per_bin:
            parts[i] = parts[i][permutation(parts[i].shape[0])[:per_bin]]
    return np.concatenate(parts)


def prepare_frame(points_lidar: np.ndarray, points_virtual: np.ndarray, training: bool, discard_rate: float = 0.8,
                  rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """-> (P, 8) points, LiDAR rows first (they claim voxel ids / slots first: LIDAR_FIRST)."""
    perm = (rng.permutation if rng is not None else np.random.permutation)
    virt = input_point_discard(points_virtual, bin_num=2 if training else 10, rate=discard_rate, permutation=perm)
    return np.concatenate([points_lidar, virt]).astype(np.float32, copy=False)


def input_point_discard_device(points: torch.Tensor, bin_num: int = 2, rate: float = 0.8, max_dis: float = 60.0,
                               permutation: Optional[Callable[[int], torch.Tensor]] = None) -> torch.Tensor:
    """input_point_discard (dataset.py:120-189) on a tensor that lives on any device; one host read (the bin histogram).

    Same output as the numpy version for the same per-bin permutations: bins far -> near, reduced bins keep
    ``permutation(n_i)[:per_bin]`` of their points (in that order), the others keep all points in input order.
    ``permutation(n) -> int64 tensor`` defaults to a device ``torch.randperm``."""
    retain = 1 - rate
    total = points.shape[0]
    inter = max_dis / bin_num
    x = points[:, 0]
    # bin of a point: i with inter*i <= x < inter*(i+1); the last bin is open-ended; x < 0 belongs to no bin (-1).
    # Compared like the reference does: x against the python-float products inter*i, in float64
    edges = torch.tensor([inter * i for i in range(bin_num)], dtype=torch.float64, device=points.device)
    b = (torch.bucketize(x.to(torch.float64), edges, right=True) - 1).to(torch.int64)
    counts = torch.bincount(b[b >= 0], minlength=bin_num).cpu().tolist()                      # the one host read
    acc, position, distant_acc = 0, bin_num - 1, 0
    for j in range(bin_num):                                                                  # far -> near
        i = bin_num - j - 1
        acc += counts[i]
        if (acc + i * counts[i]) / total < retain:
            position, distant_acc = i, acc
    position = max(position, 0)
    per_bin = int((int(total * retain) - distant_acc) / (position + 0.0001))
    # stable sort by (far -> near): rows of bin i become one contiguous, input-ordered run
    key = torch.where(b >= 0, (bin_num - 1) - b, torch.full_like(b, bin_num))
    order = torch.sort(key, stable=True)[1]
    runs, start = [], 0
    for j in range(bin_num):
        i = bin_num - j - 1
        n_i = counts[i]
        rows = order[start:start + n_i]
        start += n_i
        if j >= bin_num - position and n_i > per_bin:                                        # parts[len - pos:], reduced
            perm = permutation(n_i) if permutation is not None else torch.randperm(n_i, device=points.device)
            rows = rows[perm.to(device=rows.device, dtype=torch.int64)[:per_bin]]
        runs.append(rows)
    keep = torch.cat(runs) if runs else order[:0]
    return points.index_select(0, keep)


def prepare_frame_device(points_lidar: torch.Tensor, points_virtual: torch.Tensor, training: bool, discard_rate: float = 0.8,
                         permutation: Optional[Callable[[int], torch.Tensor]] = None) -> torch.Tensor:
    """prepare_frame on device tensors: (P, 8) float32, LiDAR rows first."""
    virt = input_point_discard_device(points_virtual, bin_num=2 if training else 10, rate=discard_rate,
                                      permutation=permutation)
    return torch.cat([points_lidar, virt]).float()


def voxelize_batch(frames: Sequence[np.ndarray], pc_range, voxel_size, max_points: int = 5, max_voxels: int = 40000,
                   vfe_max_last: bool = True, device="cuda"):
    """-> voxel_features (N, F) f32, voxel_coords (N, 4) i32 [b, z, y, x], voxel_num_points (N,) i32 on `device`."""
    be = ops.get_backend()
    feats, coords, nums = [], [], []
    for b, pts in enumerate(frames):
        t = (pts.to(device=device, dtype=torch.float32).contiguous() if torch.is_tensor(pts)
             else torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)).to(device))
        f, c, n = be.voxelize_mean(t, pc_range, voxel_size, max_points, max_voxels, vfe_max_last)
        bcol = torch.full((c.shape[0], 1), b, dtype=torch.int32, device=c.device)
        feats.append(f)
        coords.append(torch.cat([bcol, c], dim=1))
        nums.append(n)
    return torch.cat(feats), torch.cat(coords), torch.cat(nums)
