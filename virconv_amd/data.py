"""Host-side data front-end of the hot path (the part the reference runs in numpy inside its Dataset):

  input_point_discard   StVD input discard, bin-based (pcdet/datasets/dataset.py:120-189) -- numpy, like the reference
  prepare_frame         LATER_FUSION=False fusion: discard virtual points, concat LiDAR-first
                        (dataset.py:270-294, data_processor.py:152-155)
  voxelize_batch        GPU voxeliser + fused MeanVFE per frame (vc_voxelize_mean), collated with the batch index
                        prepended (dataset.py:315-367 collate_batch)
  input_point_discard_device
                        the same discard as a hand-written HIP path (vc_input_discard) on points already on the GPU
  frontend_voxelize / frontend_batch
                        SURVEY §8f rank 2: raw LiDAR + raw (fp16) virtual points in, voxel features out, ONE C-ABI call per
                        frame (vc_frontend_voxelize_mean: discard + LiDAR-first concat + voxeliser + MeanVFE), no host sync
                        inside; load_virtual_points reads the fp16 .npy the offline depth completion wrote
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
                               perms=None, seed: Optional[int] = None) -> torch.Tensor:
    """input_point_discard (dataset.py:120-189) of points that already live on the GPU: the hand-written HIP path
    (vc_input_discard -- ballot histogram, scanned stable ranks, per-bin point-wise permutation), one count read.

    ``points`` (P, F) float32 or float16 (the dtype of the depth-completion ``.npy`` files).  ``perms`` injects the per-bin
    permutations ``{bin: int64 tensor}`` (bin 0 = nearest; what the reference draws with ``np.random.permutation``) --
    with them the result equals the numpy version bit for bit; without, each reduced bin draws a pseudo-random permutation
    from ``seed`` (default: torch's CPU generator, so ``torch.manual_seed`` reproduces it)."""
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    out, _ = ops.get_backend().input_discard(points, bin_num, rate, max_dis, perms=perms, seed=seed, sync=True)
    return out


def load_virtual_points(path: str, device="cuda") -> torch.Tensor:
    """One frame of depth-completed virtual points as the offline stage wrote it: an ``.npy`` of (P, 8) float16
    ``[x, y, z, intensity, r/3, g/3, b/3, flag]`` (tools/PENet/vis_utils.py:148-152; read by kitti_dataset_mm.py:70-73 as
    ``np.load(f).astype(np.float32)``).  The half-precision rows go to the GPU as they are (half the PCIe bytes) through a
    pinned staging buffer; vc_input_discard / vc_frontend_voxelize_mean widen them on load, which is exactly ``astype``."""
    arr = np.load(path, mmap_mode="r")
    assert arr.ndim == 2 and arr.dtype in (np.float16, np.float32), (arr.shape, arr.dtype)
    host = torch.from_numpy(np.ascontiguousarray(arr))
    if torch.device(device).type == "cuda":
        host = host.pin_memory()
    return host.to(device, non_blocking=True)


def frontend_voxelize(points_lidar: torch.Tensor, points_virtual: torch.Tensor, training: bool, pc_range, voxel_size,
                      max_points: int = 5, max_voxels: int = 40000, vfe_max_last: bool = True, discard_rate: float = 0.8,
                      perms=None, seed: Optional[int] = None, intensity_div: float = 0.0, sync: bool = True):
    """The whole per-frame data front-end of the non-LATER_FUSION configs (VirConv-L) in ONE call on the device
    (vc_frontend_voxelize_mean): input discard (2 bins train / 10 bins test, dataset.py:276-278) -> LiDAR-first concat
    (dataset.py:290-292, data_processor.py:152-155) -> voxeliser + MeanVFE.  -> (features, coords [z, y, x], num_points)."""
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    return ops.get_backend().frontend_voxelize_mean(points_lidar, points_virtual, 2 if training else 10, discard_rate,
                                                    pc_range, voxel_size, max_points, max_voxels, vfe_max_last,
                                                    perms=perms, seed=seed, intensity_div=intensity_div, sync=sync)


def frontend_batch(frames, training: bool, pc_range, voxel_size, max_points: int = 5, max_voxels: int = 40000,
                   vfe_max_last: bool = True, discard_rate: float = 0.8, seed: Optional[int] = None):
    """collate_batch (dataset.py:315-367) over frontend_voxelize: frames = [(lidar (Pl, 8), virtual (Pv, 8)), ...] device
    tensors -> voxel_features (N, F), voxel_coords (N, 4) i32 [b, z, y, x].  The per-frame launches are all queued before the
    first count is read, so the host waits once, not once per frame."""
    be = ops.get_backend()
    if seed is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    pend = [be.frontend_voxelize_mean(l, v, 2 if training else 10, discard_rate, pc_range, voxel_size, max_points, max_voxels,
                                      vfe_max_last, seed=seed + 7919 * b, sync=False) for b, (l, v) in enumerate(frames)]
    feats, coords = [], []
    counts = torch.stack([p[3][0] for p in pend]).cpu().tolist() if pend and pend[0][3].is_cuda else [int(p[3][0]) for p in pend]
    for b, ((f, c, _, _), m) in enumerate(zip(pend, counts)):
        feats.append(f[:m])
        coords.append(torch.cat([torch.full((m, 1), b, dtype=torch.int32, device=c.device), c[:m]], dim=1))
    return torch.cat(feats), torch.cat(coords)


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
