"""Shared test helpers (deterministic parameter fill, golden fixture access, batch construction)."""
from __future__ import annotations

import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

MODEL_CFG = dict(NAME="VirConvL8x", NUM_FILTERS=[16, 32, 64, 64], RETURN_NUM_FEATURES_AS_DICT=True, OUT_FEATURES=64,
                 LAYER_DISCARD_RATE=0.1)
GRID = np.array([1408, 1600, 80], dtype=np.int64)


def fill_parameters(model: torch.nn.Module, seed: int) -> None:
    """Deterministic, torch-RNG-independent parameter/buffer fill (sorted state_dict order; numpy PCG64)."""
    rng = np.random.default_rng(seed)
    sd = model.state_dict()
    for key in sorted(sd.keys()):
        t = sd[key]
        if key.endswith("num_batches_tracked"):
            t.zero_()
            continue
        shape = tuple(t.shape)
        if key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("running_mean"):
            v = rng.uniform(-0.2, 0.2, shape)
        elif key.endswith(".weight") and t.dim() == 1:   # BN gamma
            v = rng.uniform(0.8, 1.2, shape)
        elif key.endswith(".bias"):
            v = rng.uniform(-0.1, 0.1, shape)
        else:                                            # conv weight (Cout, *k, Cin)
            fan_in = int(np.prod(shape[1:]))
            b = 1.0 / np.sqrt(fan_in)
            v = rng.uniform(-b, b, shape) * 3.0
        t.copy_(torch.from_numpy(v.astype(np.float32)).to(t.device))


def load_golden(name="virconv_l_ref.npz"):
    return np.load(os.path.join(GOLDEN, name))


def golden_calibs(g):
    return [{"P2": g["calib_P2"][b], "R0": g["calib_R0"][b], "Tr_velo2cam": g["calib_V2C"][b]}
            for b in range(g["calib_P2"].shape[0])]


def golden_batch(g, device="cpu"):
    calibs = golden_calibs(g)
    return {
        "batch_size": len(calibs),
        "voxel_features": torch.from_numpy(g["voxel_features"].copy()).to(device),
        "voxel_coords": torch.from_numpy(g["voxel_coords"].astype(np.float32)).to(device),
        "calib": calibs,
        "aug_param": torch.from_numpy(g["aug_param"].copy()).to(device),
    }


def sparse_out(batch_dict):
    res = {}
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        t = batch_dict["multi_scale_3d_features"][name]
        res[name] = (t.features.detach().cpu().numpy(), t.indices.cpu().numpy())
    t = batch_dict["encoded_spconv_tensor"]
    res["out"] = (t.features.detach().cpu().numpy(), t.indices.cpu().numpy())
    return res
