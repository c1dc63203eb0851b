"""CPU restatement (oracle O4) of the index / geometry side of the VirConv hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no reference tests exist).

  voxelize()            spconv Point2VoxelCPU3d / VoxelGeneratorV2 semantics, SURVEY App-A.9
                        (call site pcdet/datasets/processor/data_processor.py:14-59,137-168)
  mean_vfe()            pcdet/models/backbones_3d/vfe/mean_vfe.py:39-49                     (App-A.13)
  input_point_discard() pcdet/datasets/dataset.py:120-189                                    (App-A.12)
  layer_voxel_discard() pcdet/models/backbones_3d/spconv_backbone.py:134-147 (spconv-1.x in-place behaviour)
  index2uv()            spconv_backbone.py:8-24,54-83 + X_transform.py:139-154 + augmentor_utils.py:26-32
                        + common_utils.py:34-56 + calibration_kitti.py:120-153               (App-A.11)

All float math is float32 with one rounding per operation in a FIXED order (numpy never fuses), which is
what the HIP kernels reproduce with __fmul_rn/__fadd_rn/__fdiv_rn: integer outputs are compared bit-exactly.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------- voxelizer
def voxelize(points: np.ndarray, vsize_xyz, coors_range_xyz, max_points: int, max_voxels: int):
    """First-touch voxel hashing (App-A.9).

    For points in input order: c_j = floor((p_j - min_j)/vsize_j) (float32), drop if outside the grid; the
    first point of a cell creates the next voxel id unless max_voxels voxels already exist (then the point is
    skipped, later points of EXISTING voxels are still appended); a voxel keeps its first max_points points.
    Returns voxels (M, max_points, F) zero padded, coords (M, 3) int32 [z, y, x], num_points (M,) int32.
    """
    pts = np.ascontiguousarray(points, dtype=F32)
    vs = np.asarray(vsize_xyz, dtype=F32)
    rng = np.asarray(coors_range_xyz, dtype=F32)
    grid = np.round((rng[3:6].astype(np.float64) - rng[0:3]) / vs.astype(np.float64)).astype(np.int64)
    c = np.floor((pts[:, 0:3] - rng[0:3]) / vs).astype(np.int64)  # float32 sub, div, floor
    ok = np.all((c >= 0) & (c < grid), axis=1)
    pid = np.nonzero(ok)[0]
    c = c[ok]
    key = (c[:, 2] * grid[1] + c[:, 1]) * grid[0] + c[:, 0]  # (z, y, x) row-major
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    vid_of_uniq = np.empty(uniq.shape[0], dtype=np.int64)
    vid_of_uniq[np.argsort(first, kind="stable")] = np.arange(uniq.shape[0])
    vid = vid_of_uniq[inv]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    start = np.r_[0, np.nonzero(ks[1:] != ks[:-1])[0] + 1] if ks.size else np.zeros(0, np.int64)
    grp = np.cumsum(np.r_[False, ks[1:] != ks[:-1]]) if ks.size else np.zeros(0, np.int64)
    slot = np.empty(key.shape[0], dtype=np.int64)
    slot[order] = np.arange(key.shape[0]) - start[grp] if ks.size else 0
    m = int(min(uniq.shape[0], max_voxels))
    keep = (vid < m) & (slot < max_points)
    voxels = np.zeros((m, max_points, pts.shape[1]), dtype=F32)
    voxels[vid[keep], slot[keep]] = pts[pid[keep]]
    num = np.zeros(m, dtype=np.int32)
    np.add.at(num, vid[keep], 1)
    coords = np.zeros((m, 3), dtype=np.int32)
    creators = first[np.argsort(first, kind="stable")][:m]
    coords[:, 0] = c[creators, 2]
    coords[:, 1] = c[creators, 1]
    coords[:, 2] = c[creators, 0]
    return voxels, coords, num


def voxelize_sequential(points, vsize_xyz, coors_range_xyz, max_points, max_voxels):
    """Literal per-point loop of App-A.9 (small inputs only) -- pins the vectorised voxelize()."""
    pts = np.ascontiguousarray(points, dtype=F32)
    vs = np.asarray(vsize_xyz, dtype=F32)
    rng = np.asarray(coors_range_xyz, dtype=F32)
    grid = np.round((rng[3:6].astype(np.float64) - rng[0:3]) / vs.astype(np.float64)).astype(np.int64)
    table, voxels, coords, num = {}, [], [], []
    for p in pts:
        c = np.floor((p[0:3] - rng[0:3]) / vs).astype(np.int64)
        if np.any(c < 0) or np.any(c >= grid):
            continue
        k = (int(c[2]), int(c[1]), int(c[0]))
        v = table.get(k)
        if v is None:
            if len(voxels) >= max_voxels:
                continue
            v = len(voxels)
            table[k] = v
            voxels.append(np.zeros((max_points, pts.shape[1]), dtype=F32))
            coords.append(k)
            num.append(0)
        if num[v] < max_points:
            voxels[v][num[v]] = p
            num[v] += 1
    m = len(voxels)
    return (np.stack(voxels) if m else np.zeros((0, max_points, pts.shape[1]), F32),
            np.array(coords, dtype=np.int32).reshape(m, 3), np.array(num, dtype=np.int32))


def mean_vfe(voxels: np.ndarray, num_points: np.ndarray, model: Optional[str] = "max") -> np.ndarray:
    """MeanVFE (App-A.13): sum over slots / max(num, 1); 'max' mode overwrites the last channel."""
    s = voxels[:, 0, :].astype(F32).copy()
    for j in range(1, voxels.shape[1]):  # sequential slot order, float32
        s = s + voxels[:, j, :]
    mean = s / np.maximum(num_points.reshape(-1, 1), 1).astype(F32)
    if model == "max":
        mean[:, -1] = voxels[:, :, -1].max(axis=1)
    return mean.astype(F32)


# ----------------------------------------------------------------------------------------------- discards
def partition(points: np.ndarray, num: int = 10, max_dis: float = 60, rate: float = 0.2):
    """dataset.py:120-170, restated."""
    parts = []
    inter = max_dis / num
    n_all = points.shape[0]
    acc, position, distant_acc = 0, num - 1, 0
    for i in range(num - 1, -1, -1):
        if i == num - 1:
            mask = points[:, 0] >= inter * i
        else:
            mask = (points[:, 0] >= inter * i) & (points[:, 0] < inter * (i + 1))
        this = points[mask]
        acc += this.shape[0]
        sampled_sum = acc + i * this.shape[0]
        if sampled_sum / n_all < rate:
            position = i
            distant_acc = acc
        parts.append(this)
    return parts, max(position, 0), distant_acc


def input_point_discard(points: np.ndarray, bin_num: int = 2, rate: float = 0.8,
                        permutation: Callable[[int], np.ndarray] = np.random.permutation) -> np.ndarray:
    """dataset.py:172-189 (StVD input discard); the RNG is injected so tests can replay it."""
    retain = 1 - rate
    parts, pos, distant_acc = partition(points, num=bin_num, rate=retain)
    out_n = int(points.shape[0] * retain)
    per_bin = int((out_n - distant_acc) / (pos + 0.0001))
    for i in range(len(parts) - pos, len(parts)):
        if parts[i].shape[0] > per_bin:
            r = permutation(parts[i].shape[0])
            parts[i] = parts[i][r[:per_bin]]
    return np.concatenate(parts)


def input_point_discard_binned(points: np.ndarray, bin_num: int, rate: float, max_dis: float = 60, perms=None,
                               rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """input_point_discard (dataset.py:172-189) with the permutations injected PER BIN: ``perms[i]`` (i = bin id, 0 = nearest)
    is what ``np.random.permutation(count_i)`` would have returned for bin i; bins without an entry draw from ``rng``.
    Same arithmetic as input_point_discard above (which takes one callable for all bins)."""
    retain = 1 - rate
    parts, pos, distant_acc = partition(points, num=bin_num, max_dis=max_dis, rate=retain)
    out_n = int(points.shape[0] * retain)
    per_bin = int((out_n - distant_acc) / (pos + 0.0001))
    for j in range(len(parts) - pos, len(parts)):
        if parts[j].shape[0] > per_bin:
            i = bin_num - 1 - j
            r = None if perms is None else (perms.get(i) if isinstance(perms, dict) else perms[i])
            if r is None:
                r = (rng or np.random.default_rng()).permutation(parts[j].shape[0])
            r = np.asarray(r.cpu() if hasattr(r, "cpu") else r).astype(np.int64)
            assert r.shape[0] == parts[j].shape[0]
            parts[j] = parts[j][r[:max(per_bin, 0)]]
    return np.concatenate(parts)


def layer_voxel_discard(features: np.ndarray, indices: np.ndarray, rate: float, permutation: np.ndarray):
    """spconv_backbone.py:134-147, spconv-1.x (in-place) behaviour: rows permutation[:int(N*(1-rate))]."""
    if rate == 0:
        return features, indices
    keep = permutation[: int(features.shape[0] * (1 - rate))]
    return features[keep], indices[keep]


# ----------------------------------------------------------------------------------------------- projection
def _sat_int(x: np.ndarray) -> np.ndarray:
    """float32 -> int32 truncation with GPU semantics (v_cvt_i32_f32: saturating, NaN -> 0)."""
    y = np.where(np.isnan(x), F32(0), x).astype(np.float64)
    return np.clip(np.trunc(y), -2147483648.0, 2147483647.0).astype(np.int64).astype(np.int32)


def projection_params(calib: dict, trans_param: Optional[np.ndarray]):
    """Per-sample parameter block (what vc_project_prepare computes on the device).

    M1 = V2C^T(4x3) @ R0^T(3x3) in float32 with k-ascending, separately-rounded mul/add
    (calibration_kitti.py:126-128); P2T = P2^T (:149-150); cos/sin of -rot through float64 (common_utils.py:44-45).
    """
    v2c = np.asarray(calib["Tr_velo2cam"], dtype=F32)
    r0 = np.asarray(calib["R0"], dtype=F32)
    p2 = np.asarray(calib["P2"], dtype=F32)
    m1 = np.zeros((4, 3), dtype=F32)
    for r in range(4):
        for c in range(3):
            acc = F32(v2c[0, r] * r0[c, 0])
            acc = F32(acc + F32(v2c[1, r] * r0[c, 1]))
            acc = F32(acc + F32(v2c[2, r] * r0[c, 2]))
            m1[r, c] = acc
    p2t = np.ascontiguousarray(p2.T)
    if trans_param is None:
        cosa, sina, flip, scale, has = F32(1), F32(0), False, F32(1), False
    else:
        tp = np.asarray(trans_param, dtype=F32)
        a = -np.float64(tp[0])
        cosa, sina = F32(np.cos(a)), F32(np.sin(a))
        flip, scale, has = bool(tp[1] != 0), F32(tp[2]), True
    return m1, p2t, cosa, sina, flip, scale, has


def index2uv(indices: np.ndarray, batch_size: int, calibs, stride: int, trans_param: Optional[np.ndarray]):
    """Voxel index -> image pixel index (App-A.11).  Returns uv (N, 3) int32 [b, u, v], depth (N,) float32."""
    idx = np.asarray(indices)
    vs = np.array([0.05, 0.05, 0.05], dtype=np.float64) * stride  # hard-coded, spconv_backbone.py:8
    rng = [0.0, -40.0, -3.0]
    vsx, vsy, vsz = F32(vs[0]), F32(vs[1]), F32(vs[2])
    minx, miny, minz = F32(rng[0] + vs[0] / 2), F32(rng[1] + vs[1] / 2), F32(rng[2] + vs[2] / 2)
    uv = np.zeros((idx.shape[0], 3), dtype=np.int32)
    depth = np.zeros(idx.shape[0], dtype=F32)
    for b in range(batch_size):
        sel = idx[:, 0] == b
        cur = idx[sel]
        tp = None if trans_param is None else np.asarray(trans_param)[b]
        m1, p2t, cosa, sina, flip, scale, has = projection_params(calibs[b], tp)
        x = cur[:, 3].astype(F32) * vsx + minx
        y = cur[:, 2].astype(F32) * vsy + miny
        z = cur[:, 1].astype(F32) * vsz + minz
        if has:
            x, y, z = x / scale, y / scale, z / scale
            if flip:
                y = -y
            nsina = F32(-sina)
            x, y = x * cosa + y * nsina, x * sina + y * cosa
        rect = [((x * m1[0, c] + y * m1[1, c]) + z * m1[2, c]) + m1[3, c] for c in range(3)]
        hom = [((rect[0] * p2t[0, c] + rect[1] * p2t[1, c]) + rect[2] * p2t[2, c]) + p2t[3, c] for c in range(3)]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            uf = hom[0] / rect[2]
            vf = hom[1] / rect[2]
        uv[sel, 1] = _sat_int(uf.astype(F32))
        uv[sel, 2] = _sat_int(vf.astype(F32))
        depth[sel] = hom[2] - p2t[3, 2]
    uv[:, 0] = idx[:, 0]
    uv[:, 1] = np.clip(uv[:, 1], 0, 1400 - 1) // stride
    uv[:, 2] = np.clip(uv[:, 2], 0, 600 - 1) // stride
    return uv, depth
