"""GPU-side duration of the phases of the bench train step (events on the main stream, no profiler): forward (+ loss),
backward (incl. the join of the weight-gradient stream), optimizer (clip + fused AdamW), and the gap to the next step.

    python tools/step_phases.py [--steps 30] [--dw-main-tail K]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import ops, parallel, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--dw-main-tail", type=int, default=0)
args = ap.parse_args()
parallel.init_distributed()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
be = ops.get_backend()
assert be.lib.vc_debug_set(b"pass_dw_main_tail", args.dw_main_tail) == 0
batch = bench.make_batch([0, 1, 2, 3], dev, True)
model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
from virconv_amd import feature_pass as _fp
_opt_params = _fp.flatten_parameters(model)     # as bench.py: one flat parameter tensor per native pass
import os as _os
from virconv_amd import optim as _vo
_fused_opt = _os.environ.get("VIRCONV_FUSED_OPT", "1") != "0" and _vo.supports(_opt_params)      # as bench.py: clip + AdamW in two launches
opt = (_vo.ClipAdamW(_opt_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, max_norm=10.0) if _fused_opt else
       torch.optim.AdamW(_opt_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True))
lw = bench.make_loss_weights(dev)
torch.cuda.synchronize()
batch["inputs_ready_event"] = torch.cuda.Event()
batch["inputs_ready_event"].record()
prime = [torch.empty((1 << 30,), dtype=torch.uint8, device=dev) for _ in range(8)]
del prime


def step(ev=None):
    def mark(i):
        if ev is not None:
            ev[i].record()
    mark(0)
    opt.zero_grad(set_to_none=True)
    bd = dict(batch)
    bd["voxel_features"] = batch["voxel_features"].clone()
    out = model(bd)
    loss = (out["encoded_spconv_tensor"].dense() * lw["dense"]).sum()
    for name, t in out["multi_scale_3d_features"].items():
        loss = loss + (t.features * lw[name]).sum()
    mark(1)
    loss.backward()
    mark(2)
    if not _fused_opt:
        torch.nn.utils.clip_grad_norm_(_opt_params, 10.0)
    opt.step()
    mark(3)


import gc
gc.collect()
gc.freeze()      # as bench.py: a generation-2 collection of the ~10^6 objects `import torch` leaves is a 30-100 ms pause in the middle of the loop
for _ in range(60):
    step()
torch.cuda.synchronize()
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
for e in evs:
    step(e)
torch.cuda.synchronize()
import numpy as np
ph = np.array([[e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])] for e in evs])
gap = np.array([evs[i][3].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1)])
tot = evs[0][0].elapsed_time(evs[-1][0]) / (len(evs) - 1)
print(f"dw_main_tail {args.dw_main_tail}: step {tot:.3f} ms = forward+loss {ph[:, 0].mean():.3f} + backward {ph[:, 1].mean():.3f} "
      f"+ clip/adam {ph[:, 2].mean():.3f} + inter-step {gap.mean():.3f}   (medians {np.median(ph[:, 0]):.3f} {np.median(ph[:, 1]):.3f} "
      f"{np.median(ph[:, 2]):.3f} {np.median(gap):.3f})")
