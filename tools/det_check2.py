"""Which gradients differ between a plain backward and one with the next plan running beside it (split products on)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import backbone as bb, ops, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

dev = torch.device("cuda", 0)
be = ops.get_backend()
batch = bench.make_batch([0, 1], dev, training=True)
lw = bench.make_loss_weights(dev)
torch.manual_seed(3)
probe = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
p0 = probe.build_plan(batch["voxel_coords"], 2, batch["calib"], batch["aug_param"], batch)
bb.join_plan(p0)
batch["layer_discard_keep"] = {f"x_conv{bi + 1}": p0["stages"][bi]["keep"].clone() for bi in range(3)}
torch.cuda.synchronize()
assert be.lib.vc_debug_set(b"bw_split", 0) == 0 and be.lib.vc_debug_set(b"f32_split", int(os.environ.get("FS", "1"))) == 0


def run(ahead):
    torch.manual_seed(0)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    bd = dict(batch)
    bd["voxel_features"] = batch["voxel_features"].clone()
    if ahead:
        model.plan_ahead_begin(batch)
    out = model(bd)
    feats = {k: v.features.detach().clone() for k, v in out["multi_scale_3d_features"].items()}
    loss = bench.synthetic_loss(out, lw)
    loss.backward()
    if ahead:
        for a in model._ahead:      # the eager form the product no longer offers
            model._finish_ahead(a)
    torch.cuda.synchronize()
    model._ahead.clear()
    return float(loss), feats, {k: p.grad.detach().clone() for k, p in model.named_parameters()}


l0, f0, g0 = run(False)
for rep in range(4):
    for ahead in (False, True):
        l, f, g = run(ahead)
        bad_f = [k for k in f0 if not torch.equal(f0[k], f[k])]
        bad = [(k, float((g0[k] - g[k]).abs().max() / (g0[k].abs().max() + 1e-30))) for k in g0 if not torch.equal(g0[k], g[k])]
        print(f"rep {rep} ahead {int(ahead)} loss equal {l == l0} features differing {bad_f} gradients differing {len(bad)}: {bad[:12]}")
