"""Parameter-level soak (VERDICT r5 "next" #1a; LOG.md A.17): an UNSYNCHRONISED multi-step loop leaves exactly the state a fully
serialised run of the same steps leaves -- bit for bit.

tests/test_plan_stress_gpu.py compares geometry plans only.  The fence of LOG.md A.17 (backbone.PLAN_GUARD) rests on two assumptions
that nothing tested: only the pixel projection is vulnerable, and only the conv kernels disturb it.  In every backward pass BatchNorm,
group-sum, loss and AdamW kernels (fp32 vector-ALU work, main stream) run BESIDE the bf16-split weight-gradient kernels of the side
stream, and the next step's integer plan kernels run beside both.  Here the benchmark's own loop -- `bench.train_step` back to back,
no host synchronisation, three streams, `inputs_ready_event` set -- runs 64 steps from a seeded model, and every step's loss, every
parameter, every BatchNorm buffer and both AdamW moments at the end are compared with the SAME 64 steps run serialised: geometry plan
and weight gradients on the caller's stream, `torch.cuda.synchronize()` after every step.  The same for 64 pipelined inference frames
(features of x_conv1..4 and the dense output of every frame) and for the `--frontend` loop.
Reference semantics: spconv_backbone.py:54-83 (index2uv), :86-131, :150-229; tools/train_utils/train_utils.py:40-60.

`python tests/test_soak_gpu.py --guard 0` runs the training comparison with the fence off and prints what differs (report only: the
failure it shows is hardware-timing dependent, so it is not an assertion of the suite)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
STEPS = int(os.environ.get("VIRCONV_SOAK_STEPS", "64"))


class _Serial:
    """Everything on the caller's stream: the geometry plan (backbone._plan_stream) and the weight gradients (VIRCONV_PASS_OVERLAP_DW)."""

    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        from virconv_amd import backbone as bb
        self.bb = bb
        self.saved = bb._PLAN_STREAMS.get(self.dev.index)
        bb._PLAN_STREAMS[self.dev.index] = torch.cuda.current_stream()
        self.env = os.environ.get("VIRCONV_PASS_OVERLAP_DW")
        os.environ["VIRCONV_PASS_OVERLAP_DW"] = "0"
        return self

    def __exit__(self, *exc):
        if self.saved is None:
            self.bb._PLAN_STREAMS.pop(self.dev.index, None)
        else:
            self.bb._PLAN_STREAMS[self.dev.index] = self.saved
        if self.env is None:
            os.environ.pop("VIRCONV_PASS_OVERLAP_DW", None)
        else:
            os.environ["VIRCONV_PASS_OVERLAP_DW"] = self.env


def _fresh(dev, model_kind="L"):
    import bench
    from virconv_amd import synth
    from virconv_amd.backbone import VirConv8x, VirConvL8x
    torch.manual_seed(0)
    if model_kind == "8x":
        model = VirConv8x(bench.MODEL_CFG_8X, 8, synth.GRID_SIZE).to(dev).train()
    else:
        model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
    # as bench.main: flat parameters + the two-launch clip + AdamW (VIRCONV_FLAT_PARAMS=0 / VIRCONV_FUSED_OPT=0: the stock forms)
    from virconv_amd import feature_pass, optim
    params = feature_pass.flatten_parameters(model) if os.environ.get("VIRCONV_FLAT_PARAMS", "1") != "0" else list(model.parameters())
    if os.environ.get("VIRCONV_FUSED_OPT", "1") != "0" and optim.supports(params):
        opt = optim.ClipAdamW(params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, max_norm=10.0)
    else:
        opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
    return model, opt


def _state(model, opt, losses):
    out = {"loss": torch.stack([l.detach().reshape(()) for l in losses])}
    for k, v in model.state_dict().items():
        out["model." + k] = v.detach().clone()
    names = {id(p): n for n, p in model.named_parameters()}
    names.update({id(p): f"flat{i}" for i, p in enumerate(opt.param_groups[0]["params"]) if id(p) not in names})
    for p, st in opt.state.items():
        for k in ("exp_avg", "exp_avg_sq"):
            out[f"adam.{names[id(p)]}.{k}"] = st[k].detach().clone()
    return out


def _train_loop(dev, batch, lw, serial, steps, raw=None, model_kind="L", seed0=5000):
    import bench
    model, opt = _fresh(dev, model_kind)
    losses = []
    torch.cuda.synchronize()

    def loop():
        for t in range(steps):
            torch.manual_seed(seed0 + t)       # layer-discard seeds come from torch's CPU generator at the head of every plan
            losses.append(bench.train_step(model, opt, batch, lw, None, raw))
            if serial:
                torch.cuda.synchronize()

    if serial:
        with _Serial(dev):
            loop()
    else:
        loop()
    torch.cuda.synchronize()
    return _state(model, opt, losses)


def _diff(a, b):
    assert a.keys() == b.keys()
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if x.shape != y.shape:
            bad.append((k, "shape"))
        elif x.dtype.is_floating_point:
            # bit comparison: NaN payloads and signed zeros count
            if not torch.equal(x.contiguous().view(torch.int32) if x.dtype == torch.float32 else x, y.contiguous().view(torch.int32) if y.dtype == torch.float32 else y):
                bad.append((k, int((x != y).sum())))
        elif not torch.equal(x, y):
            bad.append((k, int((x != y).sum())))
    return bad


def _setup_batch(dev, frontend=False):
    import bench
    lw = bench.make_loss_weights(dev)
    if frontend:
        raw, batch = bench.make_raw_frames([0, 1, 2, 3], dev)
    else:
        raw, batch = None, bench.make_batch([0, 1, 2, 3], dev, training=True)
    torch.cuda.synchronize()
    batch["inputs_ready_event"] = torch.cuda.Event()
    batch["inputs_ready_event"].record()
    return batch, lw, raw


def _main_stream(dev):
    """The benchmark's loop runs on a high-priority stream (bench.main); so does this."""
    return torch.cuda.stream(torch.cuda.Stream(device=dev, priority=-1))


@pytest.mark.parametrize("frontend", [False, True], ids=["resident_inputs", "frontend"])
def test_64_unsynchronised_train_steps_leave_the_state_of_a_serialised_run(frontend):
    dev = torch.device("cuda", 0)
    with _main_stream(dev):
        batch, lw, raw = _setup_batch(dev, frontend)
        _train_loop(dev, batch, lw, False, 3, raw)             # allocator / clocks
        got = _train_loop(dev, batch, lw, False, STEPS, raw)
        want = _train_loop(dev, batch, lw, True, STEPS, raw)
    bad = _diff(got, want)
    first = None
    if any(k == "loss" for k, _ in bad):
        first = int((got["loss"].view(torch.int32) != want["loss"].view(torch.int32)).nonzero()[0])
    assert not bad, (f"{len(bad)} of {len(got)} state tensors differ between the unsynchronised three-stream loop and the serialised run "
                     f"after {STEPS} steps (first differing loss: step {first}): {bad[:10]}")
    assert torch.isfinite(got["loss"]).all()


def test_64_unsynchronised_train_steps_of_virconv8x_leave_the_state_of_a_serialised_run():
    """VirConv8x runs its LiDAR stream's pass while the virtual-point stream's image-space tables are still being
    built; the projection must not share the chip with THAT pass either."""
    import bench
    dev = torch.device("cuda", 0)
    with _main_stream(dev):
        lw = bench.make_loss_weights(dev)
        batch = bench.make_batch_8x([0, 1], dev)
        torch.cuda.synchronize()
        batch["inputs_ready_event"] = torch.cuda.Event()
        batch["inputs_ready_event"].record()
        _train_loop(dev, batch, lw, False, 3, None, "8x")
        got = _train_loop(dev, batch, lw, False, STEPS, None, "8x")
        want = _train_loop(dev, batch, lw, True, STEPS, None, "8x")
    bad = _diff(got, want)
    assert not bad, f"{len(bad)} of {len(got)} state tensors differ (VirConv8x, {STEPS} steps): {bad[:10]}"


def test_64_pipelined_inference_frames_equal_a_serialised_frame():
    import bench
    from virconv_amd import synth
    from virconv_amd.backbone import VirConvL8x
    dev = torch.device("cuda", 0)
    with _main_stream(dev):
        batch = bench.make_batch([0], dev, training=False)
        torch.manual_seed(0)
        model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).eval()
        torch.cuda.synchronize()
        batch["inputs_ready_event"] = torch.cuda.Event()
        batch["inputs_ready_event"].record()

        def frame():
            bd = dict(batch)
            bd["voxel_features"] = batch["voxel_features"].clone()
            with torch.no_grad():
                out = model(bd)
                res = {k: v.features for k, v in out["multi_scale_3d_features"].items()}
                res["dense"] = out["encoded_spconv_tensor"].dense()
            return res

        for _ in range(5):
            frame()
        got = [frame() for _ in range(STEPS)]
        torch.cuda.synchronize()
        with _Serial(dev):
            want = frame()
            torch.cuda.synchronize()
    bad = []
    for t, g in enumerate(got):
        for k in want:
            if g[k].shape != want[k].shape or not torch.equal(g[k].view(torch.int32), want[k].view(torch.int32)):
                bad.append((t, k))
    assert not bad, f"{len(bad)} outputs of {len({t for t, _ in bad})} of {STEPS} pipelined frames differ from the serialised frame: {bad[:10]}"


if __name__ == "__main__":      # report-only form, fence on or off:  python tests/test_soak_gpu.py --guard 0 [--model 8x]
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--guard", type=int, default=1)
    ap.add_argument("--model", default="L")
    ap.add_argument("--steps", type=int, default=STEPS)
    a = ap.parse_args()
    import bench
    from virconv_amd import backbone as bb
    dev = torch.device("cuda", 0)
    with _main_stream(dev):
        if a.model == "8x":
            lw = bench.make_loss_weights(dev)
            batch = bench.make_batch_8x([0, 1], dev)
            torch.cuda.synchronize()
            batch["inputs_ready_event"] = torch.cuda.Event()
            batch["inputs_ready_event"].record()
            raw = None
        else:
            batch, lw, raw = _setup_batch(dev)
        bb.PLAN_GUARD = a.guard
        _train_loop(dev, batch, lw, False, 3, raw, a.model)
        got = _train_loop(dev, batch, lw, False, a.steps, raw, a.model)
        bb.PLAN_GUARD = 1
        want = _train_loop(dev, batch, lw, True, a.steps, raw, a.model)
    bad = _diff(got, want)
    nl = (got["loss"].view(torch.int32) != want["loss"].view(torch.int32)).nonzero().reshape(-1).tolist()
    print(json.dumps({"exp": "soak", "model": a.model, "guard": a.guard, "steps": a.steps, "state_tensors": len(got), "differing": len(bad),
                      "first_differing_loss_step": nl[0] if nl else None, "differing_loss_steps": len(nl), "sample": bad[:6]}))
