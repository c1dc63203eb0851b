"""Generate tests/golden/virconv_l_fullsize_ref.npz: the reference's UNMODIFIED VirConvL8x run on the oracle operators over
two FULL synthetic KITTI frames (the 160-voxel fixtures of make_golden.py cannot catch a composition error that
only shows with populated neighbourhoods).  Build container only (needs /root/reference):

    python tests/golden/make_golden_fullsize.py

Seeds are searched until the reference's own torch projection (index2uv + X_TRANS + Calibration, spconv_backbone.py:54-83) and
the oracle's restatement give IDENTICAL pixels at every stride for both frames (they differ on ~2e-5 of the rows at fp32
rounding boundaries, i.e. on 1-2 rows of a typical frame).  Then, with PARAM_SEED weights:
  * eval mode                -> per tensor N, sha256(indices), channel sums, sampled rows          (prefix "eval")
  * train mode, float32      -> the same + BatchNorm running statistics + loss + parameter gradients (prefix "train")
  * train mode, float64      -> loss + parameter gradients of the exact arithmetic                   (prefix "train64")
The float64 run calibrates the gradient comparison: the fp32 oracle itself is up to 1e-3 * max|g| away from the exact gradient
(training-mode BatchNorm backward subtracts nearly equal sums over 1e5 rows), so a GPU run is judged against the exact values
with the fp32 oracle's own distance as the yardstick (tests/test_fullsize_fixture.py).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fullsize_fixture as fx  # noqa: E402
import refharness  # noqa: E402
from helpers import GRID, MODEL_CFG, fill_parameters  # noqa: E402
from oracle import geometry  # noqa: E402
from oracle.backend import OracleBackend  # noqa: E402
from virconv_amd import ops  # noqa: E402


def reference_model(ref, training: bool, dtype=torch.float32):
    from easydict import EasyDict
    model = ref.VirConvL8x(EasyDict(MODEL_CFG), input_channels=8, grid_size=GRID)
    fill_parameters(model, seed=fx.PARAM_SEED)
    model = model.to(dtype)
    model.train(training)
    return model


def reference_batch(feats, coords, calibs, aug, dtype=torch.float32):
    return {
        "batch_size": len(calibs),
        "voxel_features": torch.from_numpy(feats.copy()).to(dtype),
        "voxel_coords": torch.from_numpy(coords.astype(np.float32)),   # load_data_to_gpu casts coords to float
        "calib": [refharness.make_reference_calib(c) for c in calibs],
        "aug_param": torch.from_numpy(aug.copy()),
    }


def projection_matches(ref, outs, calibs, aug) -> bool:
    from pcdet.datasets.augmentor.X_transform import X_TRANS
    xt = X_TRANS()
    rc = [refharness.make_reference_calib(c) for c in calibs]
    ok = True
    for name, stride in (("x_conv1", 1), ("x_conv2", 2), ("x_conv3", 4), ("x_conv4", 8)):
        idx = outs[name][1].numpy()
        uv_ref, _ = ref.index2uv(torch.from_numpy(idx), len(calibs), rc, stride, xt, torch.from_numpy(aug.copy()))
        uv_or, _ = geometry.index2uv(idx, len(calibs), calibs, stride, aug)
        bad = int((uv_ref.numpy() != uv_or).any(axis=1).sum())
        if bad:
            print(f"    stride {stride}: {bad}/{idx.shape[0]} rows differ between the reference's torch projection and the oracle")
            ok = False
    return ok


def train_run(ref, inputs, dtype):
    model = reference_model(ref, True, dtype)
    out = model(reference_batch(*inputs, dtype=dtype))
    outs = fx.outputs_of(out)
    loss = fx.loss_of(outs)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    stats = {k: v for k, v in model.state_dict().items() if "running_" in k}
    return outs, float(loss.detach()), grads, stats


def main(first_seed=0, last_seed=400):
    ref = refharness.import_reference_backbone()
    good = []
    with ops.use_backend(OracleBackend()):
        for seed in range(first_seed, last_seed):
            inputs = fx.make_inputs([seed])
            with torch.no_grad():
                out = reference_model(ref, False)(reference_batch(*inputs))
            print(f"seed {seed}: {inputs[0].shape[0]} voxels")
            if projection_matches(ref, fx.outputs_of(out), inputs[2], inputs[3]):
                good.append(seed)
                print(f"  -> projection identical ({len(good)}/2)")
                if len(good) == 2:
                    break
        assert len(good) == 2, "no two seeds with a bit-identical projection in the range"
        inputs = fx.make_inputs(good)
        payload = {"seeds": np.array(good), "param_seed": np.array(fx.PARAM_SEED), "n_voxels": np.array(inputs[0].shape[0]),
                   "coords_sha": np.array(fx.sha(inputs[1])), "feats_sha": np.array(fx.sha(inputs[0]))}
        with torch.no_grad():
            out = reference_model(ref, False)(reference_batch(*inputs))
        outs = fx.outputs_of(out)
        assert projection_matches(ref, outs, inputs[2], inputs[3])
        payload.update(fx.summarize_outputs(outs, "eval"))
        print("eval:", {n: int(payload[f"eval_{n}_n"]) for n in fx.TENSORS})

        outs, loss, grads, stats = train_run(ref, inputs, torch.float32)
        payload.update(fx.summarize_outputs(outs, "train"))
        payload.update(fx.summarize_named(grads, "train_grad"))
        payload.update(fx.summarize_named(stats, "train_stat"))
        payload["train_loss"] = np.array(loss)
        _, loss64, grads64, _ = train_run(ref, inputs, torch.float64)
        payload.update(fx.summarize_named(grads64, "train64_grad"))
        payload["train64_loss"] = np.array(loss64)
        worst = max(float((grads[k].double() - grads64[k]).abs().max() / grads64[k].abs().max()) for k in grads)
        print(f"train: loss {loss:.6f} (float64 {loss64:.6f}); fp32 oracle vs float64 gradient, worst tensor: {worst:.2e} of max|g|")
    path = os.path.join(HERE, "virconv_l_fullsize_ref.npz")
    np.savez_compressed(path, **payload)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB, seeds {good}")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:3]))
