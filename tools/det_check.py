"""Run-to-run determinism of the train step: 4 optimiser steps from the same seed, repeated; with / without the plan built a step
ahead; with / without the split weight gradient.  Prints the losses and whether every parameter is bit-identical to the first run."""
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


def run(ahead):
    torch.manual_seed(0)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
    losses = []
    for t in range(4):
        losses.append(float(bench.train_step(model, opt, batch, lw, next_batch=batch if ahead else None)))
    torch.cuda.synchronize()
    return losses, [p.detach().clone() for p in model.parameters()]


for bw in (1, 0):
    for fs in (1, 0):
        assert be.lib.vc_debug_set(b"bw_split", bw) == 0 and be.lib.vc_debug_set(b"f32_split", fs) == 0
        ref = None
        for rep in range(4):
            for ahead in (False, True):
                l, p = run(ahead)
                if ref is None:
                    ref = (l, p)
                same = l == ref[0] and all(torch.equal(a, b) for a, b in zip(p, ref[1]))
                print(f"bw_split {bw} f32_split {fs} rep {rep} ahead {int(ahead)}: losses {['%.4f' % v for v in l]} identical to first: {same}")
