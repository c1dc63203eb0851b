"""Generate tests/golden/virconv_8x_ref.npz: the reference's UNMODIFIED VirConv8x (VirConv-T/S backbone, MM stream on,
rot_num = 3) run on the oracle operators, train mode (per-rot LiDAR stream) and eval mode (x-concatenated LiDAR stream +
decompose_tensor).  Build container only.  See make_golden.py for what such a fixture pins."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refharness  # noqa: E402
from helpers import GRID, fill_parameters  # noqa: E402
from oracle import geometry  # noqa: E402
from oracle.backend import OracleBackend  # noqa: E402
from virconv_amd import ops, synth  # noqa: E402

CFG_8X = dict(NAME="VirConv8x", NUM_FILTERS=[16, 32, 64, 64], RETURN_NUM_FEATURES_AS_DICT=True, OUT_FEATURES=64,
              LAYER_DISCARD_RATE=0.15, MM=True)
TRANSFORMS = np.array([[0.0, 0.0, 1.0], [0.39269908, 1.0, 0.98], [-0.39269908, 0.0, 1.02]], dtype=np.float32)  # rot, flip, scale


def transform_points(pts, t):
    """forward world transform: rotation about z, flip y, scaling (X_transform.py:125-137 order)."""
    p = pts.copy()
    c, s = np.float32(np.cos(t[0])), np.float32(np.sin(t[0]))
    x, y = p[:, 0] * c - p[:, 1] * s, p[:, 0] * s + p[:, 1] * c
    p[:, 0], p[:, 1] = x, y
    if t[1] != 0:
        p[:, 1] = -p[:, 1]
    p[:, 0:3] *= t[2]
    return p


def make_inputs(seed, n_lidar=100, n_virtual=320, max_voxels=110):
    fr = synth.make_frame(seed, n_lidar=n_lidar, n_virtual=n_virtual)
    virt = geometry.input_point_discard(fr["points_virtual"], 2, 0.8, np.random.default_rng(seed).permutation)
    d = {"batch_size": 1, "calib": [fr["calib"]], "transform_param": TRANSFORMS[None].copy()}
    for i, t in enumerate(TRANSFORMS):
        rid = "" if i == 0 else str(i)
        for name, pts in (("", fr["points_lidar"]), ("_mm", virt)):
            vox, c, num = geometry.voxelize(transform_points(pts, t), synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, max_voxels)
            d["voxel_features" + name + rid] = geometry.mean_vfe(vox, num, None)
            d["voxel_coords" + name + rid] = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    return d


def to_batch(d, calib_factory, device="cpu"):
    b = {}
    for k, v in d.items():
        if k == "calib":
            b[k] = [calib_factory(c) for c in v]
        elif isinstance(v, np.ndarray):
            t = torch.from_numpy(v.astype(np.float32) if "coords" in k else v.copy())
            b[k] = t.to(device)
        else:
            b[k] = v
    return b


def collect(out, rot_num=3):
    res = {}
    for i in range(rot_num):
        rid = "" if i == 0 else str(i)
        t = out["encoded_spconv_tensor" + rid]
        res[f"out{rid}_features"], res[f"out{rid}_indices"] = t.features.numpy(), t.indices.numpy()
        ms = out["multi_scale_3d_features" + rid]
        res[f"x_conv4{rid}_indices"] = ms["x_conv4"].indices.numpy()
        res[f"x_conv3{rid}_indices"] = ms["x_conv3"].indices.numpy()
        mm = out["multi_scale_3d_features_mm" + rid]
        res[f"mm_x_conv4{rid}_features"] = mm["x_conv4"].features.numpy()
        res[f"mm_x_conv2{rid}_indices"] = mm["x_conv2"].indices.numpy()
    return res


def main():
    ref = refharness.import_reference_backbone()
    from easydict import EasyDict
    d = make_inputs(300)
    payload = {k: v for k, v in d.items() if isinstance(v, np.ndarray)}
    payload["calib_P2"], payload["calib_R0"], payload["calib_V2C"] = (np.stack([c[k] for c in d["calib"]]) for k in ("P2", "R0", "Tr_velo2cam"))
    with ops.use_backend(OracleBackend()):
        for mode in ("eval", "train"):
            model = ref.VirConv8x(EasyDict(CFG_8X), input_channels=8, grid_size=GRID)
            fill_parameters(model, 11)
            model.train(mode == "train")
            with torch.no_grad():
                out = model(to_batch(d, refharness.make_reference_calib))
            res = collect(out)
            for k, v in res.items():
                if mode == "eval" or k.endswith("features"):
                    payload[f"{mode}_{k}"] = v
            print(mode, {k: v.shape for k, v in res.items() if "out" in k and "features" in k})
    path = os.path.join(HERE, "virconv_8x_ref.npz")
    np.savez_compressed(path, **payload)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB; state_dict keys {len(model.state_dict())}")


if __name__ == "__main__":
    main()
