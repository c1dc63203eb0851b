"""virconv_amd.backbone (own composition: fused projection, shared rulebooks) vs the committed fixture produced by the
REFERENCE's composition code (tests/golden/make_golden.py).  CPU leg: oracle operators; GPU leg: HIP operators."""
import numpy as np
import pytest
import torch

from helpers import GRID, MODEL_CFG, fill_parameters, golden_batch, load_golden, sparse_out
from virconv_amd.backbone import VirConvL8x

FEATURE_TOL = 1e-4  # north_star: features within 1e-4 fp32; indices bit-exact


def _run(device, training):
    g = load_golden()
    cfg = dict(MODEL_CFG, LAYER_DISCARD_MODE="spconv2_noop")  # the fixture was produced under spconv-2.x semantics
    model = VirConvL8x(cfg, input_channels=8, grid_size=GRID).to(device)
    fill_parameters(model, int(g["param_seed"]))
    model.train(training)
    with torch.no_grad():
        out = model(golden_batch(g, device))
    return g, sparse_out(out)


def _check(g, res, mode, names):
    for name in names:
        feats, idx = res[name]
        key = "out" if name == "out" else name
        if mode == "eval":
            np.testing.assert_array_equal(idx, g[f"eval_{key}_indices"], err_msg=f"{name} indices")
        ref = g[f"{mode}_{key}_features"]
        assert feats.shape == ref.shape
        err = np.abs(feats - ref).max()
        assert err <= FEATURE_TOL * max(1.0, np.abs(ref).max()), f"{mode} {name}: max abs err {err}"


def test_state_dict_keys_match_reference_layout():
    g = load_golden()
    model = VirConvL8x(MODEL_CFG, input_channels=8, grid_size=GRID)
    keys = set(model.state_dict().keys())
    assert len(keys) == 120
    assert "vir_conv2.down_layer.0.weight" in keys and "vir_conv1.d2_conv2.1.running_var" in keys
    assert tuple(model.vir_conv3.d3_conv1[0].weight.shape) == (32, 3, 3, 3, 64)  # (Cout, kz, ky, kx, Cin)
    assert tuple(model.conv_out[0].weight.shape) == (64, 3, 1, 1, 64)


def test_eval_oracle_backend_matches_reference_composition(oracle_backend):
    g, res = _run("cpu", training=False)
    _check(g, res, "eval", ["x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"])


def test_train_oracle_backend_matches_reference_composition(oracle_backend):
    g, res = _run("cpu", training=True)
    _check(g, res, "train", ["x_conv1", "out"])


@pytest.mark.gpu
def test_eval_hip_matches_reference_composition(hip_backend):
    g, res = _run("cuda", training=False)
    _check(g, res, "eval", ["x_conv1", "x_conv2", "x_conv3", "x_conv4", "out"])


@pytest.mark.gpu
def test_train_hip_matches_reference_composition(hip_backend):
    g, res = _run("cuda", training=True)
    _check(g, res, "train", ["x_conv1", "out"])
