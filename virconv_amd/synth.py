"""Seeded synthetic KITTI-shaped scenes (no KITTI data exists in the build or GPU containers).

Produces the *raw point format* the reference's loader hands to ``prepare_data``:
``(P, 8) float32 = [x, y, z, intensity, r/3, g/3, b/3, flag]`` with ``flag == 2`` for LiDAR
returns and ``flag == 1`` for virtual (depth-completed) points
(reference: pcdet/datasets/kitti/kitti_dataset_mm.py:70-73, tools/PENet/dataloaders/my_loader.py:391-418;
virtual points are stored as fp16: tools/PENet/vis_utils.py:148-152).

The geometry is a 2-manifold scene (ground plane + car boxes + two side walls) ray-cast from the sensor,
because sparse-conv cost depends on surface-like occupancy: i.i.d. random voxels would have ~0 active
neighbours and give unrepresentative rulebook sizes.

numpy only; used by bench.py, tests and the oracle alike (it is a data generator, not an algorithm
under test).
"""
from __future__ import annotations

import numpy as np

POINT_CLOUD_RANGE = np.array([0.0, -40.0, -3.0, 70.4, 40.0, 1.0], dtype=np.float32)
VOXEL_SIZE = np.array([0.05, 0.05, 0.05], dtype=np.float32)
GRID_SIZE = np.array([1408, 1600, 80], dtype=np.int64)  # x, y, z (reference: data_processor.py:130-131)

# KITTI-typical calibration (values of a standard KITTI object-detection calib file).
KITTI_P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728],
                     [0.0, 721.5377, 172.854, 0.2163791],
                     [0.0, 0.0, 1.0, 0.002745884]], dtype=np.float32)
# default R0 of the reference: pcdet/utils/calibration_kitti.py:30-32
KITTI_R0 = np.array([[0.99992624, 0.00965411, -0.0072371],
                     [-0.00968531, 0.99994343, -0.00433077],
                     [0.00719491, 0.00440054, 0.99996366]], dtype=np.float32)
KITTI_V2C = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
                      [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                      [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01]], dtype=np.float32)


def default_calib() -> dict:
    return {"P2": KITTI_P2.copy(), "R0": KITTI_R0.copy(), "Tr_velo2cam": KITTI_V2C.copy()}


def _make_boxes(rng: np.random.Generator, k: int) -> np.ndarray:
    """k car boxes [cx, cy, cz, dx, dy, dz, yaw] on the ground (anchor size VirConv-L.yaml:152)."""
    boxes = np.zeros((k, 7), dtype=np.float64)
    boxes[:, 0] = rng.uniform(6.0, 60.0, k)
    boxes[:, 1] = rng.uniform(-18.0, 18.0, k)
    boxes[:, 2] = -1.73 + 0.78
    boxes[:, 3:6] = np.array([3.9, 1.6, 1.56])
    boxes[:, 6] = rng.uniform(-np.pi, np.pi, k)
    return boxes


def _raycast(dirs: np.ndarray, boxes: np.ndarray, wall_y: float = 22.0) -> np.ndarray:
    """First-hit distance t (inf = no hit) of unit rays from the origin against ground/walls/boxes."""
    n = dirs.shape[0]
    t_best = np.full(n, np.inf)
    dz = dirs[:, 2]
    # ground plane z = -1.73
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(dz < -1e-6, -1.73 / dz, np.inf)
    t_best = np.minimum(t_best, t)
    # side walls y = +-wall_y, height up to z = 0.8
    dy = dirs[:, 1]
    for sgn in (-1.0, 1.0):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(sgn * dy > 1e-6, sgn * wall_y / dy, np.inf)
        zhit = t * dz
        ok = (zhit < 0.8) & (zhit > -1.73)
        t_best = np.minimum(t_best, np.where(ok, t, np.inf))
    # oriented boxes, slab test in the box frame
    for b in boxes:
        c, s = np.cos(-b[6]), np.sin(-b[6])
        ox, oy, oz = -b[0], -b[1], -b[2]
        o = np.array([c * ox - s * oy, s * ox + c * oy, oz])
        d = np.stack([c * dirs[:, 0] - s * dirs[:, 1], s * dirs[:, 0] + c * dirs[:, 1], dirs[:, 2]], axis=1)
        half = b[3:6] / 2.0
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            t1 = (-half - o) * inv
            t2 = (half - o) * inv
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tmax >= tmin) & (tmin > 0.5)
        t_best = np.minimum(t_best, np.where(hit, tmin, np.inf))
    return t_best


def _in_range(p: np.ndarray) -> np.ndarray:
    r = POINT_CLOUD_RANGE
    return ((p[:, 0] >= r[0]) & (p[:, 0] < r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] < r[4])
            & (p[:, 2] >= r[2]) & (p[:, 2] < r[5]))


def make_frame(seed: int, n_lidar: int = 20000, n_virtual: int = 60000) -> dict:
    """One synthetic frame: LiDAR points, virtual points, calib, aug_param.

    Returns dict(points_lidar (Pl,8) f32, points_virtual (Pv,8) f32, calib dict, aug_param (3,) f32).
    """
    rng = np.random.default_rng(seed)
    boxes = _make_boxes(rng, int(rng.integers(8, 21)))

    # --- LiDAR: 64 beams (-24.8..+2 deg), 0.09 deg azimuth over +-45 deg FOV
    elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    azim = np.deg2rad(np.arange(-45.0, 45.0, 0.09))
    ee, aa = np.meshgrid(elev, azim, indexing="ij")
    dirs = np.stack([np.cos(ee) * np.cos(aa), np.cos(ee) * np.sin(aa), np.sin(ee)], axis=-1).reshape(-1, 3)
    t = _raycast(dirs, boxes)
    ok = np.isfinite(t)
    pts = dirs[ok] * (t[ok] * (1.0 + 0.002 * rng.standard_normal(ok.sum())))[:, None]
    pts = pts[_in_range(pts)]
    if pts.shape[0] > n_lidar:
        pts = pts[np.sort(rng.choice(pts.shape[0], n_lidar, replace=False))]
    lidar = np.zeros((pts.shape[0], 8), dtype=np.float32)
    lidar[:, 0:3] = pts
    lidar[:, 3] = rng.uniform(0.0, 1.0, pts.shape[0])
    lidar[:, 7] = 2.0

    # --- virtual points: pseudo depth image 1216x352 through the KITTI pinhole, every 2nd pixel
    f, cu, cv = float(KITTI_P2[0, 0]), float(KITTI_P2[0, 2]), float(KITTI_P2[1, 2])
    uu, vv = np.meshgrid(np.arange(0, 1216, 2, dtype=np.float64), np.arange(0, 352, 1, dtype=np.float64))
    xc = (uu.ravel() + rng.uniform(0, 1, uu.size) - cu) / f
    yc = (vv.ravel() + rng.uniform(0, 1, vv.size) - cv) / f
    d = np.stack([np.ones_like(xc), -xc, -yc], axis=1)  # cam (x right, y down, z fwd) -> lidar (x fwd, y left, z up)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = _raycast(d, boxes)
    ok = np.isfinite(t)
    vp = d[ok] * (t[ok] * (1.0 + 0.01 * rng.standard_normal(ok.sum())))[:, None]  # 1% depth noise
    vp = vp[_in_range(vp) & (vp[:, 2] < 1.0)]
    if vp.shape[0] > n_virtual:
        vp = vp[np.sort(rng.choice(vp.shape[0], n_virtual, replace=False))]
    virt = np.zeros((vp.shape[0], 8), dtype=np.float32)
    virt[:, 0:3] = vp.astype(np.float16).astype(np.float32)  # files are fp16
    virt[:, 4:7] = (rng.uniform(0.0, 1.0, (vp.shape[0], 3)) / 3.0).astype(np.float16).astype(np.float32)
    virt[:, 7] = 1.0
    virt = virt[_in_range(virt)]

    aug = np.array([rng.uniform(-np.pi / 4, np.pi / 4), float(rng.integers(0, 2)), rng.uniform(0.95, 1.05)],
                   dtype=np.float32)  # [rot, flip, scale]  (VirConv-L.yaml:70-77)
    return {"points_lidar": lidar, "points_virtual": virt, "calib": default_calib(), "aug_param": aug}


def small_scene_indices(seed: int, n: int, spatial_shape, batch_size: int = 1, surface: bool = True) -> np.ndarray:
    """Unique random voxel indices (N, 4) [b, z, y, x] int32 inside a small grid.

    ``surface=True`` draws them from a few noisy planes so that neighbourhoods are populated
    (used by the small-grid parity tests that also run the dense oracle).
    """
    rng = np.random.default_rng(seed)
    D, H, W = (int(s) for s in spatial_shape)
    out = []
    per = max(1, n // batch_size)
    for b in range(batch_size):
        if surface:
            yy = rng.integers(0, H, per * 2)
            xx = rng.integers(0, W, per * 2)
            a, c = rng.uniform(-0.2, 0.2, 2)
            z0 = rng.uniform(0.2, 0.8) * D
            zz = np.clip(np.round(z0 + a * (yy - H / 2) + c * (xx - W / 2) + rng.integers(-1, 2, per * 2)), 0, D - 1)
        else:
            zz = rng.integers(0, D, per * 2)
            yy = rng.integers(0, H, per * 2)
            xx = rng.integers(0, W, per * 2)
        idx = np.stack([np.full_like(yy, b), zz.astype(np.int64), yy, xx], axis=1)
        idx = np.unique(idx, axis=0)
        idx = idx[rng.permutation(idx.shape[0])[:per]]
        out.append(idx)
    return np.concatenate(out, axis=0).astype(np.int32)
