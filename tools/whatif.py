"""Upper bounds for the train step: what would the step cost if one part of it were free?  (priorities, not product numbers)

    python tools/whatif.py [--steps 40]

Variants of the bench train step (same model, batch, optimizer as bench.py):
  base        the product step
  plan_cached the geometry plan of the first step is re-used (no rulebook / sort / row-order kernels, no count reads): the layer
              discard keeps then repeat, which a real step must not do -- an upper bound for everything on the plan stream
  no_dw       conv weights frozen (requires_grad = False): no weight-gradient launches
  both        plan_cached + no_dw
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import parallel, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--only", default="")
args = ap.parse_args()
parallel.init_distributed()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
batch = bench.make_batch([0, 1, 2, 3], dev, True)
lw = bench.make_loss_weights(dev)
torch.cuda.synchronize()
batch["inputs_ready_event"] = torch.cuda.Event()
batch["inputs_ready_event"].record()
prime = [torch.empty((1 << 30,), dtype=torch.uint8, device=dev) for _ in range(8)]
del prime
import gc  # noqa: E402
gc.collect()
gc.freeze()


def run(name, plan_cached, no_dw):
    torch.manual_seed(0)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
    if no_dw:
        for n_, p in model.named_parameters():
            if p.dim() > 1:
                p.requires_grad_(False)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01,
                            fused=True)
    if plan_cached:
        orig = model.build_plan
        cache = {}

        def cached(*a, **k):
            if "p" not in cache:
                cache["p"] = orig(*a, **k)
                torch.cuda.synchronize()
            return cache["p"]
        model.build_plan = cached
    torch.manual_seed(100)
    for _ in range(15):
        bench.train_step(model, opt, batch, lw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bench.train_step(model, opt, batch, lw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps * 1e3
    # host enqueue time of the same loop (no final sync inside the timed part)
    t0 = time.perf_counter()
    for _ in range(10):
        bench.train_step(model, opt, batch, lw)
    te = (time.perf_counter() - t0) / 10 * 1e3
    torch.cuda.synchronize()
    print(f"{name:12s} {dt:7.3f} ms/step   (enqueue {te:6.3f} ms/step)", flush=True)


for name, pc, nd in (("base", False, False), ("plan_cached", True, False), ("no_dw", False, True), ("both", True, True),
                     ("base", False, False)):
    if args.only and name not in args.only.split(","):
        continue
    run(name, pc, nd)
