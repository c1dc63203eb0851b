"""Generate tests/golden/virconv_8x_fullsize_ref.npz: the reference's UNMODIFIED VirConv8x (VirConv-T/S backbone: LiDAR stream +
virtual-point MM stream, rot_num = 3) on the oracle operators over one FULL synthetic frame -- 16 000-voxel cap per stream as in
VirConv-T.yaml's training config -- eval mode (x-concatenated LiDAR stream + decompose_tensor) and train mode (float32 and float64,
with a backward pass).  Build container only:   python tests/golden/make_golden_fullsize_8x.py [first_seed last_seed]

Same recipe and same summary format as make_golden_fullsize.py (tests/fullsize_fixture.py): per tensor N, sha256(indices), channel
sums, sampled rows; per gradient / running statistic sums + sampled entries.

One deviation from "unmodified", stated here and in the file: the reference's torch projection (`index2uv`,
spconv_backbone.py:54-83) and the oracle's restatement disagree on ~2e-5 of the rows (fp32 rounding at pixel borders,
tests/test_oracle_cpu.py measures it), and this model projects 12 tensors (3 rotations x 4 strides) of 10-36 k rows: no seed in
[300, 460) had all of them identical (40 minutes of search).  So `index2uv` -- and only it -- is replaced by the oracle's
restatement while the fixture is generated (the number of rows on which the two disagree for this frame is recorded);
VirConv8x.forward, NRConvBlock, decompose_tensor, the x-concatenation and every operator call remain the reference's own code.
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
sys.path.insert(0, HERE)

import fullsize_fixture as fx  # noqa: E402
import make_golden_8x as small  # noqa: E402
import refharness  # noqa: E402
from helpers import GRID, fill_parameters  # noqa: E402
from oracle import geometry  # noqa: E402
from oracle.backend import OracleBackend  # noqa: E402
from virconv_amd import ops  # noqa: E402


def reference_model(ref, training, dtype=torch.float32):
    from easydict import EasyDict
    model = ref.VirConv8x(EasyDict(small.CFG_8X), input_channels=8, grid_size=GRID)
    fill_parameters(model, fx.PARAM_SEED_8X)
    model = model.to(dtype)
    model.train(training)
    return model


def reference_batch(d, dtype=torch.float32):
    b = small.to_batch(d, refharness.make_reference_calib)
    for k in list(b):
        if k.startswith("voxel_features"):
            b[k] = b[k].to(dtype)
    return b


def count_projection_differences(ref, outs, d) -> int:
    from pcdet.datasets.augmentor.X_transform import X_TRANS
    xt = X_TRANS()
    rc = [refharness.make_reference_calib(c) for c in d["calib"]]
    total = 0
    for i, rid in enumerate(fx.RIDS):
        tp = torch.from_numpy(d["transform_param"][:, i].copy())
        for name, stride in (("mm_x_conv1", 1), ("mm_x_conv2", 2), ("mm_x_conv3", 4), ("mm_x_conv4", 8)):
            idx = outs[name + rid][1].numpy()
            uv_ref, _ = ref.index2uv(torch.from_numpy(idx), 1, rc, stride, xt, tp)
            uv_or, _ = geometry.index2uv(idx, 1, d["calib"], stride, d["transform_param"][:, i])
            bad = int((uv_ref.numpy() != uv_or).any(axis=1).sum())
            total += bad
            print(f"    rot {i} stride {stride}: {bad}/{idx.shape[0]} rows differ")
    return total


def projection_matches(ref, outs, d) -> bool:
    from pcdet.datasets.augmentor.X_transform import X_TRANS
    xt = X_TRANS()
    rc = [refharness.make_reference_calib(c) for c in d["calib"]]
    ok = True
    for i, rid in enumerate(fx.RIDS):
        tp = torch.from_numpy(d["transform_param"][:, i].copy())
        for name, stride in (("mm_x_conv1", 1), ("mm_x_conv2", 2), ("mm_x_conv3", 4), ("mm_x_conv4", 8)):
            idx = outs[name + rid][1].numpy()
            uv_ref, _ = ref.index2uv(torch.from_numpy(idx), 1, rc, stride, xt, tp)
            uv_or, _ = geometry.index2uv(idx, 1, d["calib"], stride, d["transform_param"][:, i])
            bad = int((uv_ref.numpy() != uv_or).any(axis=1).sum())
            if bad:
                print(f"    rot {i} stride {stride}: {bad}/{idx.shape[0]} rows differ")
                ok = False
    return ok


def train_run(ref, d, dtype):
    model = reference_model(ref, True, dtype)
    out = model(reference_batch(d, dtype))
    outs = fx.outputs_of_8x(out)
    loss = fx.loss_of(outs, fx.TENSORS_8X_TRAIN)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    stats = {k: v for k, v in model.state_dict().items() if "running_" in k}
    return outs, float(loss.detach()), grads, stats


def main(seed=300):
    ref = refharness.import_reference_backbone()
    with ops.use_backend(OracleBackend()):
        d = fx.make_inputs_8x(seed)
        with torch.no_grad():
            out = reference_model(ref, False)(reference_batch(d))
        print(f"seed {seed}: lidar {d['voxel_features'].shape[0]} / mm {d['voxel_features_mm'].shape[0]} voxels; reference torch projection vs oracle:")
        n_diff = count_projection_differences(ref, fx.outputs_of_8x(out), d)
        orig_index2uv = ref.index2uv

        def index2uv_oracle(indices, batch_size, calib, stride, x_trans_train, trans_param):
            uv, depth = geometry.index2uv(indices.numpy(), batch_size, d["calib"], stride, trans_param.numpy())
            return torch.from_numpy(uv), torch.from_numpy(np.asarray(depth))

        ref.index2uv = index2uv_oracle
        with torch.no_grad():
            out = reference_model(ref, False)(reference_batch(d))
        outs = fx.outputs_of_8x(out)
        payload = {"seed": np.array(seed), "param_seed": np.array(fx.PARAM_SEED_8X),
                   "projection_rows_differing_in_reference_torch_code": np.array(n_diff),
                   "coords_sha": np.array(fx.sha(np.concatenate([d[k] for k in sorted(d) if k.startswith("voxel_coords")])))}
        payload.update(fx.summarize_outputs(outs, "eval", fx.TENSORS_8X_EVAL, fx.K_ROWS_8X))
        print("eval:", {n: int(payload[f"eval_{n}_n"]) for n in fx.TENSORS_8X_EVAL})
        outs, loss, grads, stats = train_run(ref, d, torch.float32)
        payload.update(fx.summarize_outputs(outs, "train", fx.TENSORS_8X_TRAIN, fx.K_ROWS_8X))
        payload.update(fx.summarize_named(grads, "train_grad"))
        payload.update(fx.summarize_named(stats, "train_stat"))
        payload["train_loss"] = np.array(loss)
        _, loss64, grads64, _ = train_run(ref, d, torch.float64)
        payload.update(fx.summarize_named(grads64, "train64_grad"))
        payload["train64_loss"] = np.array(loss64)
        worst = max(float((grads[k].double() - grads64[k]).abs().max() / grads64[k].abs().max()) for k in grads)
        print(f"train: loss {loss:.6f} (float64 {loss64:.6f}); fp32 oracle vs float64 gradient, worst tensor: {worst:.2e} of max|g|")
        ref.index2uv = orig_index2uv
    path = os.path.join(HERE, "virconv_8x_fullsize_ref.npz")
    np.savez_compressed(path, **payload)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB, seed {seed}")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:2]))
