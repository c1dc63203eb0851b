"""virconv_amd.backbone.VirConv8x (VirConv-T/S: LiDAR stream + MM stream, rot_num = 3, eval-time x-concatenation) vs the
fixture produced by the reference's unmodified VirConv8x (tests/golden/make_golden_8x.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GRID, fill_parameters, load_golden
from virconv_amd.backbone import VirConv8x

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
TOL = 1e-4
CFG_8X = dict(NAME="VirConv8x", NUM_FILTERS=[16, 32, 64, 64], RETURN_NUM_FEATURES_AS_DICT=True, OUT_FEATURES=64,
              LAYER_DISCARD_RATE=0.15, MM=True, LAYER_DISCARD_MODE="spconv2_noop")


def _batch(g, device):
    b = {"batch_size": 1, "transform_param": torch.from_numpy(g["transform_param"].copy()).to(device),
         "calib": [{"P2": g["calib_P2"][0], "R0": g["calib_R0"][0], "Tr_velo2cam": g["calib_V2C"][0]}]}
    for k in g.files:
        if k.startswith("voxel_features"):
            b[k] = torch.from_numpy(g[k].copy()).to(device)
        elif k.startswith("voxel_coords"):
            b[k] = torch.from_numpy(g[k].astype(np.float32)).to(device)
    return b


def _run_and_check(device, mode):
    g = load_golden("virconv_8x_ref.npz")
    model = VirConv8x(CFG_8X, input_channels=8, grid_size=GRID).to(device)
    fill_parameters(model, 11)
    model.train(mode == "train")
    with torch.no_grad():
        out = model(_batch(g, device))
    for i in range(3):
        rid = "" if i == 0 else str(i)
        t = out["encoded_spconv_tensor" + rid]
        mm = out["multi_scale_3d_features_mm" + rid]
        ms = out["multi_scale_3d_features" + rid]
        if mode == "eval":
            np.testing.assert_array_equal(t.indices.cpu().numpy(), g[f"eval_out{rid}_indices"])
            np.testing.assert_array_equal(ms["x_conv4"].indices.cpu().numpy(), g[f"eval_x_conv4{rid}_indices"])
            np.testing.assert_array_equal(ms["x_conv3"].indices.cpu().numpy(), g[f"eval_x_conv3{rid}_indices"])
            np.testing.assert_array_equal(mm["x_conv2"].indices.cpu().numpy(), g[f"eval_mm_x_conv2{rid}_indices"])
            assert ms["x_conv1"] is None and t.spatial_shape == [4, 200, 176]
        for got, key in ((t.features, f"{mode}_out{rid}_features"), (mm["x_conv4"].features, f"{mode}_mm_x_conv4{rid}_features")):
            ref = g[key]
            err = np.abs(got.cpu().numpy() - ref).max()
            assert err <= TOL * max(1.0, np.abs(ref).max()), f"{key}: {err}"
    assert out["encoded_spconv_tensor_stride_mm"] == 8


def test_state_dict_layout():
    m = VirConv8x(CFG_8X, input_channels=8, grid_size=GRID)
    keys = list(m.state_dict().keys())
    assert len(keys) == 186 and "conv_input.0.weight" in keys and "conv2.0.0.weight" in keys and "vir_conv4.d2_conv2.1.bias" in keys


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_8x_oracle_backend_matches_reference(oracle_backend, mode):
    _run_and_check("cpu", mode)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_8x_hip_matches_reference(hip_backend, mode):
    _run_and_check("cuda", mode)
