"""Host-side profile of the bench step (cProfile): where the Python/torch time of one train step goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import parallel, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

parallel.init_distributed()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if os.environ.get("MODEL") == "8x":      # MODEL=8x python tools/hostprof.py: the VirConv8x train step (bs 2)
    from virconv_amd.backbone import VirConv8x
    batch = bench.make_batch_8x([0, 1], dev)
    model = VirConv8x(bench.MODEL_CFG_8X, 8, synth.GRID_SIZE).to(dev).train()
else:
    batch = bench.make_batch([0, 1, 2, 3], dev, True)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
raw = model
import os as _os
if _os.environ.get("MODE") == "infer":   # MODE=infer [BS=1]: the forward-only loop of bench.run_infer (plan of the next frame begun ahead)
    bs = int(_os.environ.get("BS", "1"))
    batch = bench.make_batch(list(range(bs)), dev, training=False)
    model.eval()
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))

    def istep():
        bd = dict(batch)
        bd["voxel_features"] = batch["voxel_features"].clone()
        with torch.no_grad():
            model.plan_ahead_begin(batch)
            return model(bd)["encoded_spconv_tensor"].dense()

    for _ in range(30):
        istep()
    torch.cuda.synchronize()
    N = int(_os.environ.get("STEPS", "300"))
    t0 = time.perf_counter()
    for _ in range(N):
        istep()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"enqueue {1e3 * (t1 - t0) / N:.3f} ms/frame, wall {1e3 * (time.perf_counter() - t0) / N:.3f} ms/frame")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        istep()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(70)
    st.sort_stats("cumulative").print_stats(50)
    sys.exit(0)
gs = None
if _os.environ.get('VIRCONV_TORCH_DDP') == '1':
    model = parallel.wrap_ddp(model, dev)
else:
    gs = parallel.FlatGradAllReduce(model)
_ts = bench.train_step
bench.train_step = lambda m, o, b, l: _ts(m, o, b, l, gs)
from virconv_amd import feature_pass as _fp, optim as _vo  # noqa: E402
_params = _fp.flatten_parameters(raw) if _os.environ.get("VIRCONV_FLAT_PARAMS", "1") != "0" else list(raw.parameters())   # as bench.py
opt = (_vo.ClipAdamW(_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, max_norm=10.0)
       if _os.environ.get("VIRCONV_FUSED_OPT", "1") != "0" and _vo.supports(_params) else
       torch.optim.AdamW(_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True))
lw = bench.make_loss_weights(dev)
torch.cuda.synchronize()
batch["inputs_ready_event"] = torch.cuda.Event()
batch["inputs_ready_event"].record()
for _ in range(3):
    bench.train_step(model, opt, batch, lw)
torch.cuda.synchronize()
# pure host cost: time to ENQUEUE a step when the GPU is not the bottleneck is approximated by timing without final sync
t0 = time.perf_counter()
for _ in range(5):
    bench.train_step(model, opt, batch, lw)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 5:.2f} ms/step, drained after +{1e3 * (t2 - t1):.2f} ms")
N = int(_os.environ.get("STEPS", "40"))
pr = cProfile.Profile()
for _ in range(N):          # queue drained between steps: the count read of the plan never waits, what is left is enqueue cost
    pr.enable()
    bench.train_step(model, opt, batch, lw)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
print(f"{N} steps profiled (main thread only: the backward bodies run on autograd's worker thread)")
st.sort_stats("tottime").print_stats(60)
st.sort_stats("cumulative").print_stats(60)
