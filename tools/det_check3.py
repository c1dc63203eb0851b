"""Bisecting the run-to-run differences of training with the plan built a step ahead and split products on."""
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
lw = bench.make_loss_weights(dev)
assert be.lib.vc_debug_set(b"bw_split", 0) == 0 and be.lib.vc_debug_set(b"f32_split", 1) == 0


def make(frames):
    batch = bench.make_batch(frames, dev, training=True)
    torch.manual_seed(3)
    probe = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    p0 = probe.build_plan(batch["voxel_coords"], len(frames), batch["calib"], batch["aug_param"], batch)
    bb.join_plan(p0)
    batch["layer_discard_keep"] = {f"x_conv{bi + 1}": p0["stages"][bi]["keep"].clone() for bi in range(3)}
    torch.cuda.synchronize()
    return batch


def step(model, opt, batch, mode):
    opt.zero_grad(set_to_none=True)
    bd = dict(batch)
    bd["voxel_features"] = batch["voxel_features"].clone()
    if mode in ("A", "B", "D"):
        model.plan_ahead_begin(batch)
    out = model(bd)
    if mode == "C":
        model.plan_ahead_begin(batch)
    loss = bench.synthetic_loss(out, lw)
    loss.backward()
    if mode in ("A", "C", "D"):
        for a in model._ahead:          # the eager form the product no longer offers: tables under the running backward pass
            model._finish_ahead(a)
    if mode == "D":
        torch.cuda.synchronize()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
    return float(loss.detach())


def run(batch, mode):
    torch.manual_seed(0)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
    l = [step(model, opt, batch, mode) for _ in range(4)]
    torch.cuda.synchronize()
    return l


KEEP = []
if os.environ.get("KEEP_ARENAS") == "1":   # hypothesis: a consumed plan's arenas are reused while the backward still reads them
    orig_take = VirConvL8x._take_ahead

    def take(self, coords, rid):
        plan = orig_take(self, coords, rid)
        if plan is not None:
            KEEP.append(plan["_arenas"])
        return plan

    VirConvL8x._take_ahead = take

for name, frames in (("bs2", [0, 1]),):
    batch = make(frames)
    ref = run(batch, "plain")
    print(name, "plain", ["%.4f" % v for v in ref])
    for mode in ("A",):
        res = [run(batch, mode) == ref for _ in range(6)]
        print(f"{name} mode {mode}: identical to plain in {sum(res)} of {len(res)} runs")
