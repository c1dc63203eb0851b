"""virconv_amd.backbone.VirConvL8x at KITTI scale against the fixture produced by the reference's UNMODIFIED composition code
(tests/golden/make_golden_fullsize.py -> virconv_l_fullsize_ref.npz; two full synthetic frames, 66 k / 146 k / 96 k / 40 k rows).

  * CPU (`-m "not gpu"`): the inputs regenerate bit-identically from the seeds; virconv_amd.backbone on the ORACLE backend equals
    the reference composition (eval and train: N, indices, sampled rows, channel sums, running statistics, loss, gradients);
    when /root/reference is present the reference composition itself is re-run and must reproduce the committed fixture.
  * GPU: the HIP path against the same fixture.  Gradients are judged against the fixture's float64 run with the fp32
    reference-composition run as the yardstick (the fp32 oracle is itself up to 1.7e-3 * max|g| from the exact gradient on this
    batch); the rigorous treatment of isolated ReLU-mask flips is tests/test_fullsize_gpu.py::_compare_train (masks compared and
    forced) -- here entries up to 2e-3 * max are tolerated and <= 3 % of the SAMPLED entries of a tensor may go up to 2e-2 * max.
"""
import os

import numpy as np
import pytest
import torch

import fullsize_fixture as fx
import refharness
from helpers import GRID, MODEL_CFG, fill_parameters, load_golden
from virconv_amd.backbone import VirConvL8x

CFG = dict(MODEL_CFG, LAYER_DISCARD_MODE="spconv2_noop")   # what the reference's code does under spconv 2.x


@pytest.fixture(autouse=True)
def _fixture_thread_count():
    """The fp32 CPU runs below are compared with a fixture made at 8 torch threads; torch's CPU reductions chunk by the thread COUNT (not by
    the cores there are), so the count is pinned: OMP_NUM_THREADS=1 or a 32-thread box otherwise shift sums by fp32 re-association noise that
    the tight bounds of these tests (1e-5 class) do not allow."""
    n = torch.get_num_threads()
    torch.set_num_threads(8)
    yield
    torch.set_num_threads(n)


@pytest.fixture(scope="module")
def fixture_inputs():
    g = load_golden("virconv_l_fullsize_ref.npz")
    inputs = fx.make_inputs([int(s) for s in g["seeds"]])
    return g, inputs


def _batch(inputs, device):
    feats, coords, calibs, aug = inputs
    return {"batch_size": len(calibs), "voxel_features": torch.from_numpy(feats.copy()).to(device),
            "voxel_coords": torch.from_numpy(coords.astype(np.float32)).to(device), "calib": calibs,
            "aug_param": torch.from_numpy(aug.copy()).to(device)}


def _model(device, training):
    m = VirConvL8x(CFG, input_channels=8, grid_size=GRID).to(device)
    fill_parameters(m, fx.PARAM_SEED)
    m.train(training)
    return m


def _eval_check(g, inputs, device, report=None):
    with torch.no_grad():
        out = _model(device, False)(_batch(inputs, device))
    fx.check_outputs(fx.outputs_of(out), g, "eval", report=report)


def _train_run(inputs, device):
    m = _model(device, True)
    out = m(_batch(inputs, device))
    outs = fx.outputs_of(out)
    loss = fx.loss_of(outs)
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    stats = {k: v for k, v in m.state_dict().items() if "running_" in k}
    return outs, float(loss.detach()), grads, stats


def test_fixture_inputs_regenerate_bit_identically(fixture_inputs):
    g, inputs = fixture_inputs
    assert inputs[0].shape[0] == int(g["n_voxels"]) > 60000
    assert fx.sha(inputs[1]) == str(g["coords_sha"]) and fx.sha(inputs[0]) == str(g["feats_sha"])


def test_fullsize_oracle_backend_equals_the_reference_composition_eval(oracle_backend, fixture_inputs):
    g, inputs = fixture_inputs
    _eval_check(g, inputs, "cpu")


def test_fullsize_oracle_backend_equals_the_reference_composition_train(oracle_backend, fixture_inputs):
    """Same operators under both compositions: everything agrees to fp32 re-association noise (1e-5 class)."""
    g, inputs = fixture_inputs
    outs, loss, grads, stats = _train_run(inputs, "cpu")
    fx.check_outputs(outs, g, "train")
    assert abs(loss - float(g["train_loss"])) <= 1e-2
    fx.check_named(stats, g, "train_stat", rtol=1e-5)
    fx.check_named(grads, g, "train_grad", rtol=2e-4)


@pytest.mark.skipif(not refharness.available(), reason="reference tree not present (GPU box)")
def test_fixture_regenerates_from_the_reference_composition(fixture_inputs):
    """Re-run the reference's unmodified VirConvL8x (eval) on the oracle operators: it must reproduce the committed fixture."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_fullsize as mk
    from oracle.backend import OracleBackend
    from virconv_amd import ops
    g, inputs = fixture_inputs
    ref = refharness.import_reference_backbone()
    with ops.use_backend(OracleBackend()):
        with torch.no_grad():
            out = mk.reference_model(ref, False)(mk.reference_batch(*inputs))
        outs = fx.outputs_of(out)
        assert mk.projection_matches(ref, outs, inputs[2], inputs[3])
    new = fx.summarize_outputs(outs, "eval")
    for k, v in new.items():
        if v.dtype.kind in "US":
            assert str(v) == str(g[k]), k
        else:
            np.testing.assert_allclose(v, g[k], rtol=1e-6, atol=1e-6, err_msg=k)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_fullsize_hip_equals_the_reference_composition_eval(hip_backend, fixture_inputs):
    g, inputs = fixture_inputs
    report = []
    _eval_check(g, inputs, "cuda", report)
    _write_report("fixture_eval", report)


@pytest.mark.gpu
def test_fullsize_hip_equals_the_reference_composition_train(hip_backend, fixture_inputs):
    g, inputs = fixture_inputs
    report = []
    outs, loss, grads, stats = _train_run(inputs, "cuda")
    fx.check_outputs(outs, g, "train", report=report)
    l64, l32 = float(g["train64_loss"]), float(g["train_loss"])
    assert abs(loss - l64) <= max(1e-2, 3 * abs(l32 - l64)), (loss, l64, l32)
    fx.check_named(stats, g, "train_stat", rtol=1e-5, report=report)
    worst = _check_grads(grads, g, report)
    report.append(f"loss {loss:.6f} (reference composition float64 {l64:.6f}, float32 {l32:.6f}); worst gradient entry {worst:.2e} of max|g|")
    _write_report("fixture_train", report)


def _write_report(tag, rows):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_{tag}.txt"), "w") as f:
        f.write("\n".join(rows) + "\n")


# ================================================================================================ VirConv8x (VirConv-T/S backbone)
# tests/golden/virconv_8x_fullsize_ref.npz (make_golden_fullsize_8x.py): the reference's VirConv8x.forward, NRConvBlock,
# decompose_tensor and x-concatenation on one full frame -- with the reference's torch index2uv replaced by the oracle's
# restatement during generation (3 of 245 k projected rows differ between the two on this frame; see the generator's docstring).
CFG_8X = dict(NAME="VirConv8x", NUM_FILTERS=[16, 32, 64, 64], RETURN_NUM_FEATURES_AS_DICT=True, OUT_FEATURES=64,
              LAYER_DISCARD_RATE=0.15, MM=True, LAYER_DISCARD_MODE="spconv2_noop")


@pytest.fixture(scope="module")
def fixture_8x():
    g = load_golden("virconv_8x_fullsize_ref.npz")
    return g, fx.make_inputs_8x(int(g["seed"]))


def _batch_8x(d, device):
    b = {"batch_size": 1, "calib": d["calib"], "transform_param": torch.from_numpy(d["transform_param"].copy()).to(device)}
    for k, v in d.items():
        if k.startswith("voxel_features"):
            b[k] = torch.from_numpy(v.copy()).to(device)
        elif k.startswith("voxel_coords"):
            b[k] = torch.from_numpy(v.astype(np.float32)).to(device)
    return b


def _model_8x(device, training):
    from virconv_amd.backbone import VirConv8x
    m = VirConv8x(CFG_8X, input_channels=8, grid_size=GRID).to(device)
    fill_parameters(m, fx.PARAM_SEED_8X)
    m.train(training)
    return m


def _train_run_8x(d, device):
    m = _model_8x(device, True)
    out = m(_batch_8x(d, device))
    outs = fx.outputs_of_8x(out)
    loss = fx.loss_of(outs, fx.TENSORS_8X_TRAIN)
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    stats = {k: v for k, v in m.state_dict().items() if "running_" in k}
    return outs, float(loss.detach()), grads, stats


def _check_grads(grads, g, report):
    """Sampled gradient entries against the fixture's float64 run, the fp32 reference-composition run as yardstick; isolated
    ReLU flips allowed on <= 3 % of a tensor's sampled entries (a fixture cannot carry masks: see the module docstring)."""
    worst = 0.0
    for ti, name in enumerate(sorted(grads)):
        v = grads[name].detach().cpu().numpy().astype(np.float64).reshape(-1)
        pos = fx.sample_positions(v.shape[0], fx.K_GRAD, 100 + ti)
        r64 = g[f"train64_grad|{name}|val"].astype(np.float64)
        r32 = g[f"train_grad|{name}|val"].astype(np.float64)
        mx = max(float(g[f"train64_grad|{name}|sum"][2]), 1e-12)
        e32 = float(np.abs(r32 - r64).max()) / mx
        err = np.abs(v[pos] - r64) / mx
        bound = max(1e-4, 3 * e32)
        # a flipped ReLU-mask entry upstream moves EVERY channel of the downstream BatchNorm gradients a little (measured: up to
        # 2.5e-4 * max on 16 of 64 entries of one gamma) and a few entries of a weight gradient a lot: entries up to 2e-3 * max are
        # tolerated, larger ones (<= 2e-2) on at most 3 % of the sampled entries.  This is the coarse net for COMPOSITION errors
        # (which are O(1)); the 1e-4 statement is made by the forced-mask test (tests/test_fullsize_gpu.py).
        n_over = int((err > max(bound, 2e-3)).sum())
        allowed = max(2, int(0.03 * err.size))
        report.append(f"train_grad {name:34s} hip-f64 {err.max():.2e} f32ref-f64 {e32:.2e} bound {bound:.2e} "
                      f"over-bound {int((err > bound).sum())} over-2e-3 {n_over}/{allowed}")
        assert err.max() <= bound or (n_over <= allowed and err.max() <= 2e-2), report[-1]
        worst = max(worst, float(err.max()))
    return worst


def test_8x_fixture_inputs_regenerate_bit_identically(fixture_8x):
    g, d = fixture_8x
    assert d["voxel_features"].shape[0] > 10000 and d["voxel_features_mm"].shape[0] == 16000
    assert fx.sha(np.concatenate([d[k] for k in sorted(d) if k.startswith("voxel_coords")])) == str(g["coords_sha"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_8x_fullsize_oracle_backend_equals_the_reference_composition(oracle_backend, fixture_8x, mode):
    g, d = fixture_8x
    if mode == "eval":
        with torch.no_grad():
            out = _model_8x("cpu", False)(_batch_8x(d, "cpu"))
        fx.check_outputs(fx.outputs_of_8x(out), g, "eval", names=fx.TENSORS_8X_EVAL, k_rows=fx.K_ROWS_8X)
        return
    outs, loss, grads, stats = _train_run_8x(d, "cpu")
    fx.check_outputs(outs, g, "train", names=fx.TENSORS_8X_TRAIN, k_rows=fx.K_ROWS_8X)
    # Round 6: the fp32 CPU run is compared with the fixture's FLOAT64 run, the fixture's own fp32 run as yardstick, exactly as the HIP test
    # below does.  The former comparison with the fp32 run (|loss| 1e-2 absolute, gradients 2e-4 of max) held only at the thread count
    # the fixture was made with: torch's CPU reductions sum in a thread-dependent order, and one flipped ReLU-mask entry moves a weight
    # gradient by 2e-3 of its max (OMP_NUM_THREADS=1 and the GPU box's 32 threads both failed it, at the round-5 tree as well).
    l64, l32 = float(g["train64_loss"]), float(g["train_loss"])
    assert abs(loss - l64) <= max(1e-2, 3 * abs(l32 - l64)), (loss, l64, l32)
    fx.check_named(stats, g, "train_stat", rtol=1e-5)
    _check_grads(grads, g, [])


@pytest.mark.gpu
def test_8x_fullsize_hip_equals_the_reference_composition_eval(hip_backend, fixture_8x):
    g, d = fixture_8x
    report = []
    with torch.no_grad():
        out = _model_8x("cuda", False)(_batch_8x(d, "cuda"))
    fx.check_outputs(fx.outputs_of_8x(out), g, "eval", report=report, names=fx.TENSORS_8X_EVAL, k_rows=fx.K_ROWS_8X)
    _write_report("fixture_8x_eval", report)


@pytest.mark.gpu
def test_8x_fullsize_hip_equals_the_reference_composition_train(hip_backend, fixture_8x):
    g, d = fixture_8x
    report = []
    outs, loss, grads, stats = _train_run_8x(d, "cuda")
    fx.check_outputs(outs, g, "train", report=report, names=fx.TENSORS_8X_TRAIN, k_rows=fx.K_ROWS_8X)
    l64, l32 = float(g["train64_loss"]), float(g["train_loss"])
    assert abs(loss - l64) <= max(1e-2, 3 * abs(l32 - l64)), (loss, l64, l32)
    fx.check_named(stats, g, "train_stat", rtol=1e-5, report=report)
    worst = _check_grads(grads, g, report)
    report.append(f"loss {loss:.6f} (reference composition float64 {l64:.6f}, float32 {l32:.6f}); worst gradient entry {worst:.2e} of max|g|")
    _write_report("fixture_8x_train", report)
