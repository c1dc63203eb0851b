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


def _run_and_check(device, mode, plan_ahead=True):
    g = load_golden("virconv_8x_ref.npz")
    model = VirConv8x(dict(CFG_8X, PLAN_AHEAD=plan_ahead), input_channels=8, grid_size=GRID).to(device)
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


@pytest.mark.parametrize("plan_ahead", [True, False])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_8x_oracle_backend_matches_reference(oracle_backend, mode, plan_ahead):
    _run_and_check("cpu", mode, plan_ahead)


@pytest.mark.gpu
@pytest.mark.parametrize("plan_ahead", [True, False])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_8x_hip_matches_reference(hip_backend, mode, plan_ahead):
    _run_and_check("cuda", mode, plan_ahead)


@pytest.mark.gpu
@pytest.mark.parametrize("bwd_epilogue", [0, 1])
def test_8x_train_step_with_layer_discard_plan_equals_inline(hip_backend, bwd_epilogue):
    """Train mode with the spconv-1.x layer discard on (injected permutations): the plan-ahead path (native feature pass) and
    the inline path (node by node) give identical outputs, and identical gradients -- bit for bit when the pass forms the
    BatchNorm-backward sums the way the nodes do (pass_bwd_epilogue = 0: same kernels, same order; only where the geometry is
    built and who issues the launches differs), up to the summation order of those sums with the epilogue fusion (default)."""
    g = load_golden("virconv_8x_ref.npz")
    assert hip_backend.lib.vc_debug_set(b"pass_bwd_epilogue", bwd_epilogue) == 0
    res = []
    try:
        for plan_ahead in (True, False):
            cfg = dict(CFG_8X, PLAN_AHEAD=plan_ahead, LAYER_DISCARD_MODE="spconv1_inplace")
            model = VirConv8x(cfg, input_channels=8, grid_size=GRID).cuda()
            fill_parameters(model, 11)
            model.train()
            b = _batch(g, "cuda")
            torch.manual_seed(7)   # the discard permutations are drawn with torch.randperm in both paths, in the same order
            out = model(b)
            loss = sum((out["encoded_spconv_tensor" + r].features.sum() + out["multi_scale_3d_features_mm" + r]["x_conv4"].features.sum())
                       for r in ("", "1", "2"))
            loss.backward()
            res.append((out["multi_scale_3d_features_mm"]["x_conv4"].features.detach().clone(),
                        {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        assert hip_backend.lib.vc_debug_set(b"pass_bwd_epilogue", 1) == 0
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys()
    for k in res[0][1]:
        if bwd_epilogue == 0:
            assert torch.equal(res[0][1][k], res[1][1][k]), k
        else:
            tol = 1e-5 * max(float(res[1][1][k].abs().max()), 1e-30)
            assert float((res[0][1][k] - res[1][1][k]).abs().max()) <= tol, k
