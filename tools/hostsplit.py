"""Where the host time of one train step goes, WITHOUT a profiler: wall-clock brackets around the step's segments and around every
C-ABI call (vc_*), queue drained between steps (pure enqueue cost).  python tools/hostsplit.py [steps]"""
import os
import sys
import time
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import ops, parallel, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
parallel.init_distributed()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
if os.environ.get("MODEL", "L") == "8x":      # MODEL=8x: VirConv8x at its benchmark shape (2 frames per rank)
    from virconv_amd.backbone import VirConv8x
    batch = bench.make_batch_8x([0, 1], dev)
    model = VirConv8x(bench.MODEL_CFG_8X, 8, synth.GRID_SIZE).to(dev).train()
else:
    batch = bench.make_batch([0, 1, 2, 3], dev, True)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
gs = parallel.FlatGradAllReduce(model)
import os as _os
from virconv_amd import feature_pass as _fp
# as bench.py: one flat parameter tensor per native pass (VIRCONV_FLAT_PARAMS=0: per module, the round-5 form)
_opt_params = _fp.flatten_parameters(model) if _os.environ.get("VIRCONV_FLAT_PARAMS", "1") != "0" else list(model.parameters())
from virconv_amd import optim as _vo
_fused_opt = _os.environ.get("VIRCONV_FUSED_OPT", "1") != "0" and _vo.supports(_opt_params)      # as bench.py: clip + AdamW in two launches
opt = (_vo.ClipAdamW(_opt_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, max_norm=10.0) if _fused_opt else
       torch.optim.AdamW(_opt_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True))
lw = bench.make_loss_weights(dev)
batch["inputs_ready_event"] = torch.cuda.Event()
batch["inputs_ready_event"].record()
be = ops.get_backend()
pc = time.perf_counter

native = defaultdict(lambda: [0, 0.0])
ON = [False]


def wrap_native(lib):
    from virconv_amd import _lib
    for name in _lib.SIGNATURES:
        fn = getattr(lib, name)

        def w(*a, _fn=fn, _n=name):
            if not ON[0]:
                return _fn(*a)
            t = pc()
            r = _fn(*a)
            e = native[_n]
            e[0] += 1
            e[1] += pc() - t
            return r
        setattr(lib, name, w)


wrap_native(be.lib)
seg = defaultdict(float)
fn_t = defaultdict(lambda: [0, 0.0])
import cProfile
import pstats
PROF = cProfile.Profile()   # PROFILE_BODY=PassFunction.backward: cProfile of that body only (it runs on autograd's worker thread)


def wrap_functions():
    """every autograd.Function of the package: host time of its forward and backward (the backward runs on autograd's worker thread)"""
    def subclasses(c):
        for k in c.__subclasses__():
            yield k
            yield from subclasses(k)
    for cls in set(subclasses(torch.autograd.Function)):
        if not cls.__module__.startswith("virconv_amd") and cls.__module__ != "bench":
            continue
        for which in ("forward", "backward"):
            orig = getattr(cls, which)

            def w(*a, _o=orig, _k=f"{cls.__name__}.{which}"):
                prof = PROF if (_k == os.environ.get("PROFILE_BODY") and ON[0]) else None
                t = pc()
                if prof is not None:
                    prof.enable()
                r = _o(*a)
                if prof is not None:
                    prof.disable()
                if ON[0]:
                    e = fn_t[_k]
                    e[0] += 1
                    e[1] += pc() - t
                return r
            setattr(cls, which, staticmethod(w))


wrap_functions()


def timed(name, fn, *a, **k):
    t = pc()
    r = fn(*a, **k)
    seg[name] += pc() - t
    return r


_build_plan = model.build_plan
model.build_plan = lambda *a, **k: timed("  forward: build_plan", _build_plan, *a, **k)
params = _opt_params


def step():
    timed("zero_grad", opt.zero_grad, set_to_none=True)
    t = pc()
    bd = dict(batch)
    bd["voxel_features"] = batch["voxel_features"].clone()
    seg["clone"] += pc() - t
    out = timed("forward (all)", model, bd)
    loss = timed("loss", bench.synthetic_loss, out, lw)
    timed("backward", loss.backward)
    timed("grad_sync", gs)
    if not _fused_opt:
        timed("clip_grad_norm_", torch.nn.utils.clip_grad_norm_, params, 10.0)
    timed("optimizer.step", opt.step)


prime = [torch.empty((1 << 30,), dtype=torch.uint8, device=dev) for _ in range(8)]   # as bench.py: allocator priming, gc.freeze
del prime
import gc
gc.collect()
gc.freeze()
for _ in range(8):
    step()
torch.cuda.synchronize()
seg.clear()
ON[0] = True
tot = 0.0
for _ in range(STEPS):
    t = pc()
    step()
    tot += pc() - t
    if os.environ.get("DRAIN", "1") != "0":
        torch.cuda.synchronize()
ON[0] = False
DRAIN = os.environ.get("DRAIN", "1") != "0"
print(f"host enqueue {1e3 * tot / STEPS:.3f} ms per step (DRAIN={int(DRAIN)}: 1 = queue drained between steps), {STEPS} steps")
for k, v in seg.items():
    print(f"  {k:32s} {1e3 * v / STEPS:7.3f} ms")
print("C-ABI calls (inside the segments above):")
nt = 0.0
for k, (c, v) in sorted(native.items(), key=lambda kv: -kv[1][1]):
    nt += v
    print(f"  {k:36s} {c / STEPS:6.1f} calls  {1e3 * v / STEPS:7.3f} ms")
print("autograd.Function bodies (host time inside; they contain the C-ABI calls):")
for k, (c, v) in sorted(fn_t.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:44s} {c / STEPS:6.1f} calls  {1e3 * v / STEPS:7.3f} ms")
ms = torch.cuda.memory_stats()
print("allocator:", {k: ms[k] for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "num_sync_all_streams") if k in ms})
print(f"  all C-ABI calls {1e3 * nt / STEPS:.3f} ms; Python / torch / autograd around them {1e3 * (tot - nt) / STEPS:.3f} ms")

if os.environ.get("PROFILE_BODY"):
    pstats.Stats(PROF).sort_stats("tottime").print_stats(25)
