"""Diagnostic (not a bench line): ms/step of the bench train step per one-second window from process start, to see whether the
first process on a fresh box converges to the steady-state step time and after how long.  python tools/coldstart.py [seconds]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import parallel, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

t_start = time.perf_counter()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
parallel.init_distributed()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
batch = bench.make_batch([0, 1, 2, 3], dev, True)
model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
lw = bench.make_loss_weights(dev)
torch.cuda.synchronize()
batch["inputs_ready_event"] = torch.cuda.Event()
batch["inputs_ready_event"].record()
prime = [torch.empty((1 << 30,), dtype=torch.uint8, device=dev) for _ in range(8)]
del prime
print(f"setup done {time.perf_counter() - t_start:.1f} s after process start", flush=True)
t0 = time.perf_counter()
win_t, win_n, out = t0, 0, []
while time.perf_counter() - t0 < secs:
    bench.train_step(model, opt, batch, lw, None)
    win_n += 1
    if win_n % 8 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        if now - win_t >= 1.0:
            out.append(f"{1e3 * (now - win_t) / win_n:.2f}")
            win_t, win_n = now, 0
print("ms/step per ~1 s window:", " ".join(out), flush=True)
