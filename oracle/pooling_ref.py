"""CPU restatement (oracle) of the RoI-grid-pooling operators that consume the backbone outputs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED against a run of the reference: its implementation of
this path is CUDA (`pcdet/ops/pointnet2/pointnet2_stack/src/*.cu`), which cannot be built or run in the build container
(no nvcc, no CUDA device; no hipify by rule) and the reference ships no tests or vectors for it.  Unlike the spconv path
the algorithm is fully in the tree, so every function below follows the cited source line by line:

  voxel2pinds()        pcdet/utils/spconv_utils.py:4-21   (scatter_point_inds / generate_voxel2pinds)
  voxel_centers()      pcdet/utils/common_utils.py:65-81  (get_voxel_centers)
  voxel_query()        pointnet2_stack/src/voxel_query_gpu.cu:10-90 (voxel_query_kernel_stack)
                       + voxel_query_utils.py:39-44 (empty-ball post-processing of VoxelQuery.forward)
  group_points()       pointnet2_stack/src/group_points_gpu.cu:70-100 (group_points_kernel_stack)
  group_points_grad()  pointnet2_stack/src/group_points_gpu.cu:15-45  (group_points_grad_kernel_stack)

Float rule: the distance test is evaluated in float32 with ONE rounding per operation, left to right
(((dx*dx) + (dy*dy)) + (dz*dz)) > r*r, which is what the HIP kernel reproduces (__fmul_rn/__fadd_rn).  nvcc would be
free to contract this into FMAs; the two can only differ for a neighbour within one ulp of the sphere surface.
Duplicate coordinates in voxel2pinds (undefined order in the reference's advanced-index assignment) resolve to the highest
row, the same rule as the coordinate hash of the product.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

F32 = np.float32


def voxel2pinds(indices: np.ndarray, batch_size: int, spatial_shape: Sequence[int]) -> np.ndarray:
    """(N, 4) [b, z, y, x] -> dense (B, Z, Y, X) int32 volume holding the row of each active voxel, -1 elsewhere."""
    vol = np.full((batch_size,) + tuple(int(s) for s in spatial_shape), -1, dtype=np.int32)
    idx = np.asarray(indices, dtype=np.int64)
    rows = np.arange(idx.shape[0], dtype=np.int32)
    np.maximum.at(vol, (idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), rows)
    return vol


def voxel_centers(coords_zyx: np.ndarray, downsample_times, voxel_size, point_cloud_range) -> np.ndarray:
    """(N, 3) [z, y, x] int -> (N, 3) [x, y, z] float32 metric centres: (c + 0.5) * (voxel_size * stride) + range_min."""
    c = np.asarray(coords_zyx)[:, [2, 1, 0]].astype(F32)
    vs = np.asarray(voxel_size, dtype=F32) * F32(downsample_times)
    return ((c + F32(0.5)) * vs + np.asarray(point_cloud_range[0:3], dtype=F32)).astype(F32)


def voxel_query(max_range: Sequence[int], radius: float, nsample: int, xyz: np.ndarray, new_xyz: np.ndarray,
                new_coords: np.ndarray, point_indices: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """-> idx (M, nsample) int32 (rows of `xyz`), empty_ball_mask (M,) bool.

    For query m with voxel coordinate new_coords[m] = [b, z, y, x]: scan dz, dy, dx (each ascending over [-range, +range],
    out-of-grid cells skipped), take the voxels whose centre is within `radius` of new_xyz[m] in scan order; the first hit
    fills all nsample slots, later hits overwrite slots 1, 2, ... until nsample are taken.  No hit: mask set, row = 0.
    """
    zr, yr, xr = (int(v) for v in max_range)
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    q = np.ascontiguousarray(new_xyz, dtype=F32)
    nc = np.asarray(new_coords, dtype=np.int64)
    vol = np.asarray(point_indices)
    _, R1, R2, R3 = vol.shape
    M = nc.shape[0]
    idx = np.zeros((M, nsample), dtype=np.int32)
    cnt = np.zeros(M, dtype=np.int64)
    r2 = F32(radius) * F32(radius)
    rows_all = np.arange(M)
    for dz in range(-zr, zr + 1):
        z = nc[:, 1] + dz
        okz = (z >= 0) & (z < R1)
        for dy in range(-yr, yr + 1):
            y = nc[:, 2] + dy
            oky = okz & (y >= 0) & (y < R2)
            if not oky.any():
                continue
            for dx in range(-xr, xr + 1):
                x = nc[:, 3] + dx
                ok = oky & (x >= 0) & (x < R3) & (cnt < nsample)
                rows = rows_all[ok]
                if rows.size == 0:
                    continue
                nb = vol[nc[rows, 0], z[rows], y[rows], x[rows]]
                hit = nb >= 0
                rows, nb = rows[hit], nb[hit]
                if rows.size == 0:
                    continue
                d = xyz[nb] - q[rows]                       # (x_per - new_x, y_per - new_y, z_per - new_z), float32
                d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
                inside = ~(d2 > r2)
                rows, nb = rows[inside], nb[inside]
                first = cnt[rows] == 0
                idx[rows[first], :] = nb[first, None]
                idx[rows, cnt[rows]] = nb
                cnt[rows] += 1
    empty = cnt == 0
    idx[empty] = 0
    return idx, empty


def _batch_offsets(cnt: np.ndarray) -> np.ndarray:
    c = np.asarray(cnt, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(c)[:-1]])


def group_points(features: np.ndarray, features_batch_cnt, idx: np.ndarray, idx_batch_cnt) -> np.ndarray:
    """features (N, C), batch-LOCAL idx (M, nsample) -> (M, C, nsample): out[m, c, s] = features[start(b(m)) + idx[m, s], c]."""
    f = np.asarray(features)
    idx = np.asarray(idx, dtype=np.int64)
    start = _batch_offsets(features_batch_cnt)
    b_of_m = np.repeat(np.arange(len(idx_batch_cnt)), np.asarray(idx_batch_cnt, dtype=np.int64))
    rows = idx + start[b_of_m][:, None]                     # (M, nsample) global rows
    return np.ascontiguousarray(f[rows].transpose(0, 2, 1))


def group_points_grad(grad_out: np.ndarray, idx: np.ndarray, idx_batch_cnt, features_batch_cnt, n: int) -> np.ndarray:
    """grad_out (M, C, nsample) -> grad_features (N, C): scatter-add of the transpose of group_points."""
    g = np.asarray(grad_out)
    idx = np.asarray(idx, dtype=np.int64)
    start = _batch_offsets(features_batch_cnt)
    b_of_m = np.repeat(np.arange(len(idx_batch_cnt)), np.asarray(idx_batch_cnt, dtype=np.int64))
    rows = idx + start[b_of_m][:, None]
    out = np.zeros((n, g.shape[1]), dtype=np.float64)
    np.add.at(out, rows.reshape(-1), g.transpose(0, 2, 1).reshape(-1, g.shape[1]).astype(np.float64))
    return out.astype(g.dtype)
