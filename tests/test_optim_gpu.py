"""vc_clip_adamw / virconv_amd.optim.ClipAdamW against torch: clip_grad_norm_ + AdamW (train_utils.py:50-51; the optimizer of
optimization/__init__.py:19-32 stepped as fastai_optim.py:132-149 = AdamW).  The checker is torch itself: the same steps in float64 on the
CPU (the formula, free of fp32 rounding) and torch's own fp32 GPU pair (the thing the class replaces in bench.train_step)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LR, BETAS, EPS, WD = 1e-3, (0.9, 0.99), 1e-8, 0.01


def _params(sizes, dev, seed=0, misalign=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in sizes:
        base = torch.randn((n + 1,), generator=g)
        t = base.to(dev)
        out.append(torch.nn.Parameter(t[1:] if misalign else t[:n].clone()))
    return out


def _grads(sizes, step, scale, seed=0):
    g = torch.Generator().manual_seed(1000 * seed + step)
    return [scale * torch.randn((n,), generator=g) for n in sizes]


def _reference64(p0, grad_steps, max_norm, lrs=None):
    """float64 CPU: clip_grad_norm_ + torch.optim.AdamW, step by step.  -> (params, norms, clipped grads of the last step)"""
    ps = [torch.nn.Parameter(p.detach().double().cpu().clone()) for p in p0]
    opt = torch.optim.AdamW(ps, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD)
    norms = []
    for t, gs in enumerate(grad_steps):
        if lrs is not None:
            opt.param_groups[0]["lr"] = lrs[t]
        for p, g in zip(ps, gs):
            p.grad = g.double().clone()
        norms.append(float(torch.nn.utils.clip_grad_norm_(ps, max_norm)) if max_norm else
                     float(torch.sqrt(sum((p.grad ** 2).sum() for p in ps))))
        opt.step()
    return [p.detach() for p in ps], norms, [p.grad.clone() for p in ps]


@pytest.mark.parametrize("sizes,scale,max_norm,misalign", [
    ((430_003,), 1.0, 10.0, False),        # one flat tensor the size of VirConv-L's parameters (+ a scalar tail); norm ~ 656: clipped
    ((430_000,), 1e-3, 10.0, False),       # norm ~ 0.66: the clip is inactive
    ((250_001, 180_002, 7), 1.0, 10.0, False),   # VirConv8x: two passes + a small rest; ONE norm over all of them
    ((100_003, 65), 1.0, 10.0, True),      # storage offsets that are not 16-byte aligned: the scalar form
    ((50_000,), 1.0, None, False),         # no clipping, the norm is still reported
])
def test_clip_adamw_matches_torch(hip_backend, sizes, scale, max_norm, misalign):
    from virconv_amd.optim import ClipAdamW
    dev = torch.device("cuda", 0)
    steps = 6
    ps = _params(sizes, dev, misalign=misalign)
    if misalign:
        assert all(p.data_ptr() % 16 != 0 for p in ps)
    grad_steps = [_grads(sizes, t, scale) for t in range(steps)]
    want, norms, clipped = _reference64(ps, grad_steps, max_norm)
    # torch's own fp32 pair on the GPU, for scale: how far does IT land from the float64 run?
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    stock = torch.optim.AdamW(qs, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD, fused=True)
    opt = ClipAdamW(ps, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD, max_norm=max_norm, scale_grads=True)
    for t, gs in enumerate(grad_steps):
        for p, q, g in zip(ps, qs, gs):
            p.grad = g.to(dev)
            q.grad = g.to(dev)
        norm = opt.step()
        if max_norm:
            torch.nn.utils.clip_grad_norm_(qs, max_norm)
        stock.step()
        assert abs(float(norm) - norms[t]) <= 2e-6 * norms[t], (t, float(norm), norms[t])
    for k, (p, q, w) in enumerate(zip(ps, qs, want)):
        got, st = p.detach().double().cpu(), q.detach().double().cpu()
        err, err_stock = (got - w).abs().max().item(), (st - w).abs().max().item()
        # six steps of lr 1e-3 move an entry by <= 6e-3; fp32 rounding of p (|p| <= 5) is 3e-7 per step
        assert err <= 3e-6, (k, err, err_stock)
        assert err <= 4 * err_stock + 1e-6, (k, err, err_stock)
        # scale_grads: .grad holds what clip_grad_norm_ leaves there
        gerr = (p.grad.double().cpu() - clipped[k]).abs().max().item()
        assert gerr <= 1e-6 * max(1.0, clipped[k].abs().max().item()), (k, gerr)
    # the moments too (state layout = torch.optim.AdamW's)
    for p, q in zip(ps, qs):
        a, b = opt.state[p], stock.state[q]
        assert float(a["step"]) == float(b["step"]) == steps
        assert torch.allclose(a["exp_avg"], b["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(a["exp_avg_sq"], b["exp_avg_sq"], rtol=1e-5, atol=1e-9)


def test_clip_adamw_bit_stable_and_grad_untouched(hip_backend):
    from virconv_amd.optim import ClipAdamW
    dev = torch.device("cuda", 0)
    sizes = (430_003, 1_029)
    runs = []
    for _ in range(3):
        ps = _params(sizes, dev)
        opt = ClipAdamW(ps, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD, max_norm=10.0)
        for t in range(4):
            gs = _grads(sizes, t, 1.0)
            for p, g in zip(ps, gs):
                p.grad = g.to(dev)
            before = [p.grad.clone() for p in ps]
            opt.step()
            assert all(torch.equal(p.grad, b) for p, b in zip(ps, before))     # default: grad is only read
        runs.append([p.detach().clone() for p in ps] + [opt.state[p]["exp_avg_sq"].clone() for p in ps] + [opt.total_norm.clone()])
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r))


def test_clip_adamw_state_dict_moves_to_torch_adamw_and_back(hip_backend):
    """An optimizer checkpoint written under one loads under the other, and the next step agrees."""
    from virconv_amd.optim import ClipAdamW
    dev = torch.device("cuda", 0)
    sizes = (20_000, 3_000)
    ps, qs = _params(sizes, dev), _params(sizes, dev)
    a = ClipAdamW(ps, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD, max_norm=10.0)
    for t in range(3):
        for p, g in zip(ps, _grads(sizes, t, 1.0)):
            p.grad = g.to(dev)
        a.step()
    b = torch.optim.AdamW(qs, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD, fused=True)
    sd = copy.deepcopy(a.state_dict())
    for k in ("max_norm", "scale_grads"):
        sd["param_groups"][0].pop(k)
    sd["param_groups"][0].update({k: v for k, v in b.state_dict()["param_groups"][0].items() if k not in sd["param_groups"][0]})
    b.load_state_dict(sd)
    with torch.no_grad():
        for p, q in zip(ps, qs):
            q.copy_(p)
    gs = _grads(sizes, 3, 1.0)
    for p, q, g in zip(ps, qs, gs):
        p.grad, q.grad = g.to(dev), g.to(dev)
    a.step()
    torch.nn.utils.clip_grad_norm_(qs, 10.0)
    b.step()
    for p, q in zip(ps, qs):
        assert (p - q).abs().max().item() <= 1e-6
    # and back
    c = ClipAdamW(_params(sizes, dev), lr=LR, betas=BETAS, eps=EPS, weight_decay=WD, max_norm=10.0)
    sd = copy.deepcopy(b.state_dict())
    sd["param_groups"][0].update(max_norm=10.0, scale_grads=False)
    c.load_state_dict(sd)
    assert all(float(c.state[p]["step"]) == 4.0 for p in c.param_groups[0]["params"])
    # (fused AdamW keeps its step counter on the device; ClipAdamW takes it to the host at its first step, once)
    for p, g in zip(c.param_groups[0]["params"], _grads(sizes, 4, 1.0)):
        p.grad = g.to(dev)
    c.step()
    assert all(float(c.state[p]["step"]) == 5.0 and not c.state[p]["step"].is_cuda for p in c.param_groups[0]["params"])


def test_clip_adamw_follows_a_one_cycle_schedule(hip_backend):
    """lr and beta1 are read at every step (adam_onecycle moves both per iteration)."""
    from virconv_amd.optim import ClipAdamW
    dev = torch.device("cuda", 0)
    sizes = (30_001,)
    ps = _params(sizes, dev)
    opt = ClipAdamW(ps, lr=LR, betas=BETAS, eps=EPS, weight_decay=WD, max_norm=10.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-3, total_steps=8, cycle_momentum=False)
    lrs, grad_steps = [], []
    for t in range(8):
        gs = _grads(sizes, t, 1.0)
        grad_steps.append(gs)
        ps[0].grad = gs[0].to(dev)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert len(set(lrs)) == 8
    want, _, _ = _reference64(_params(sizes, dev), grad_steps, 10.0, lrs=lrs)
    assert (ps[0].detach().double().cpu() - want[0]).abs().max().item() <= 3e-6


def test_clip_adamw_refuses_what_it_cannot_take(hip_backend):
    from virconv_amd import optim
    dev = torch.device("cuda", 0)
    many = [torch.nn.Parameter(torch.zeros(4, device=dev)) for _ in range(17)]
    assert not optim.supports(many)
    with pytest.raises(ValueError, match="flatten_parameters"):
        optim.ClipAdamW(many)
    with pytest.raises(ValueError):
        optim.ClipAdamW([torch.nn.Parameter(torch.zeros(4))])              # CPU tensor
    with pytest.raises(ValueError):
        optim.ClipAdamW([torch.nn.Parameter(torch.zeros(4, device=dev, dtype=torch.float64))])
    p = torch.nn.Parameter(torch.zeros(8, device=dev))
    opt = optim.ClipAdamW([p])
    assert float(opt.step()) == 0.0 and not opt.state[p]                    # no gradient anywhere: nothing moves
    p.grad = torch.full((8,), float("nan"), device=dev)
    opt.step()
    assert torch.isnan(opt.total_norm) and torch.isnan(p).all()             # as clip_grad_norm_ + AdamW: NaN propagates, loudly


def test_train_step_with_clip_adamw_tracks_the_stock_pair(hip_backend):
    """bench.train_step on the benchmark model: ClipAdamW against clip_grad_norm_ + fused AdamW from the same start, same injected
    layer-discard seeds.  One step: equal gradients (bit for bit: the backward does not depend on the optimizer), parameters within fp32
    rounding of the update.  Four steps: losses stay together (the trajectories are allowed their ulps)."""
    import bench
    from virconv_amd import feature_pass, optim, synth
    from virconv_amd.backbone import VirConvL8x
    dev = torch.device("cuda", 0)
    batch = bench.make_batch([0, 1], dev, training=True)
    lw = bench.make_loss_weights(dev)
    losses, finals, grads1 = {}, {}, {}
    for kind in ("stock", "fused"):
        torch.manual_seed(7)
        model = VirConvL8x(bench.MODEL_CFG, input_channels=8, grid_size=synth.GRID_SIZE).to(dev)
        model.train()
        params = feature_pass.flatten_parameters(model)
        assert optim.supports(params)
        opt = (optim.ClipAdamW(params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, max_norm=10.0) if kind == "fused" else
               torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True))
        ls = []
        for t in range(4):
            torch.manual_seed(900 + t)
            ls.append(float(bench.train_step(model, opt, batch, lw)))
            if t == 0:
                grads1[kind] = [p.grad.clone() for p in params]
                finals[kind + "1"] = [p.detach().clone() for p in params]
        losses[kind] = ls
        finals[kind] = [p.detach().clone() for p in params]
    assert losses["stock"][0] == losses["fused"][0]
    # the stock route rescales .grad in place, ClipAdamW leaves it: compare after undoing nothing -- the norm decides
    n_f = torch.sqrt(sum((g.double() ** 2).sum() for g in grads1["fused"]))
    coef = min(10.0 / (float(n_f) + 1e-6), 1.0)
    for gs, gf in zip(grads1["stock"], grads1["fused"]):
        assert torch.allclose(gs, gf * coef, rtol=2e-6, atol=0)
    for a, b in zip(finals["stock1"], finals["fused1"]):
        assert (a - b).abs().max().item() <= 2e-6          # one step of lr 1e-3
    for a, b in zip(losses["stock"], losses["fused"]):
        assert abs(a - b) <= 2e-4 * abs(a), (losses,)
