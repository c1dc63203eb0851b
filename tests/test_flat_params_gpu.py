"""feature_pass.flatten_parameters: one flat parameter tensor per native pass (host time of the optimizer / clip / autograd hand-over).

The modules' parameters become views of the flat buffer; the pass takes the flat tensor as its one parameter input and returns one
gradient.  Against the per-parameter form from the same seed: the first step's loss and the whole gradient are bit-identical (same
kernels, same pointers); the updated parameters differ only through the clip coefficient (norm of one tensor vs norm of 60 norms:
last-bit rounding), and state_dict / load_state_dict are unchanged.  Reference: tools/train_utils/train_utils.py:40-60."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flatten_parameters_is_a_no_op_off_the_gpu():
    from virconv_amd import feature_pass
    m = torch.nn.Linear(4, 4)
    assert [id(p) for p in feature_pass.flatten_parameters(m)] == [id(p) for p in m.parameters()]
    assert [id(p) for p in feature_pass.trainable_parameters(m)] == [id(p) for p in m.parameters()]


def _run(kind, flat, steps, dev):
    import bench
    from virconv_amd import feature_pass, synth
    from virconv_amd.backbone import VirConv8x, VirConvL8x
    torch.manual_seed(0)
    if kind == "8x":
        model = VirConv8x(bench.MODEL_CFG_8X, 8, synth.GRID_SIZE).to(dev).train()
        batch = bench.make_batch_8x([0, 1], dev)
    else:
        model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
        batch = bench.make_batch([0, 1], dev, training=True)
    params = feature_pass.flatten_parameters(model) if flat else list(model.parameters())
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
    lw = bench.make_loss_weights(dev)
    losses, grads = [], None
    for t in range(steps):
        torch.manual_seed(900 + t)
        losses.append(bench.train_step(model, opt, batch, lw))
        if t == 0:    # the clipped gradient of the first step, gathered in module order
            if flat:
                P = [p for p in params]
                views = {}
                for fp, prog in ((fp, pr) for fp in P for pr in feature_pass._training_programs(model) if getattr(pr, "_flat", None) is fp):
                    for i, (conv, bn) in enumerate(prog.units):
                        for t_, o in ((conv.weight, prog.grad_offsets[3 * i]), (bn.weight, prog.grad_offsets[3 * i + 1]),
                                      (bn.bias, prog.grad_offsets[3 * i + 2])):
                            views[id(t_)] = fp.grad[o: o + t_.numel()].view(t_.shape).clone()
                grads = [views[id(p)] for p in model.parameters()]
            else:
                grads = [p.grad.clone() for p in model.parameters()]
    torch.cuda.synchronize()
    return model, params, torch.stack([l.detach() for l in losses]), grads


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["L", "8x"])
def test_flat_parameters_train_like_per_module_parameters(kind):
    dev = torch.device("cuda", 0)
    m0, p0, l0, g0 = _run(kind, False, 3, dev)
    m1, p1, l1, g1 = _run(kind, True, 3, dev)
    assert len(p1) <= 2 and len(p0) >= 60
    # the modules still own their parameters, by the same names, and they alias the flat buffers
    assert list(m0.state_dict().keys()) == list(m1.state_dict().keys())
    lo, hi = p1[0].data_ptr(), p1[0].data_ptr() + 4 * p1[0].numel()
    assert any(lo <= p.data_ptr() < hi for p in m1.parameters())
    assert all(p.grad is None for p in m1.parameters())
    # step 1: same forward bit for bit; the clipped gradient differs by the clip coefficient's last bit at most
    assert torch.equal(l0[:1], l1[:1])
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=5e-7, atol=0), float((a - b).abs().max())
    # three steps later the two models still agree to fp32 rounding of the update -- except where Adam amplifies it: an entry whose
    # gradient is at rounding level (|g| < 1e-6 of the tensor's largest; VirConv8x's conv_input has a few dozen: 2.9e-8 against 1.4) moves
    # by lr * g / (|g| + eps), i.e. by up to +-lr per step on the SIGN of noise, and the clip coefficient's last bit is enough to flip
    # some of them (deterministically: a rerun of either model is bit-identical).  So: every entry within 1e-5 or 1e-4 relative, except
    # entries with such a gradient, which stay within what three such steps can move.
    lr, steps = 1e-3, 3
    grad_of = {n: g for (n, _), g in zip(m0.named_parameters(), g0)}
    for (k, a), b in zip(m0.state_dict().items(), m1.state_dict().values()):
        if a.dtype.is_floating_point:
            d = (a - b).abs()
            off = d > torch.maximum(1e-4 * b.abs(), torch.full_like(d, 1e-5))
            if k in grad_of:
                g = grad_of[k].abs()
                off &= ~(g < 1e-6 * g.max())
            assert not bool(off.any()), (k, int(off.sum()), float(d[off].max()))
            assert float(d.max()) <= 2 * lr * steps, (k, float(d.max()))
        else:
            assert torch.equal(a, b), k
    assert torch.allclose(l0, l1, rtol=1e-5, atol=1e-6), (l0, l1)
    # checkpoints: load_state_dict copies into the views, the flat buffer follows
    sd = {k: (v + 1 if v.dtype.is_floating_point else v) for k, v in m0.state_dict().items()}
    m1.load_state_dict(sd)
    from virconv_amd import feature_pass
    assert all(pr.flat() is not None for pr in feature_pass._training_programs(m1))
    w = next(iter(m1.parameters()))
    assert torch.equal(w, sd[next(iter(sd.keys()))])


@pytest.mark.gpu
def test_the_cached_unit_structs_follow_moved_tensors_and_the_cached_verdict_follows_swapped_modules():
    """feature_pass keeps the vc_pass_unit array on the program and the units' plain-shape verdict behind a fingerprint (host time).  A
    parameter or buffer that moves (p.data = ..., a re-registered buffer, load_state_dict(assign=True)) must be seen by the very next forward,
    and a swapped BatchNorm must send the model off the native pass."""
    import copy
    import bench
    from virconv_amd import feature_pass, synth
    from virconv_amd.backbone import VirConvL8x
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).eval()
    batch = bench.make_batch([0], dev, training=False)

    def fwd(m):
        bd = dict(batch)
        bd["voxel_features"] = batch["voxel_features"].clone()
        with torch.no_grad():
            out = m(bd)
        return [out["encoded_spconv_tensor"].features.clone()] + [t.features.clone() for t in out["multi_scale_3d_features"].values()]

    base = fwd(model)
    assert all(torch.equal(a, b) for a, b in zip(base, fwd(model)))          # (second call: everything comes from the caches)
    # move tensors: new storage with new VALUES, so a stale pointer would be visible in the output
    conv, bn = model.vir_conv2.d3_conv1[0], model.vir_conv2.d3_conv1[1]
    conv.weight.data = conv.weight.data.clone() * 1.5
    bn.running_mean = bn.running_mean.clone() + 0.25                          # buffer re-registered: another tensor object
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    k0 = "vir_conv3.d2_conv2.1.weight"
    sd[k0] = sd[k0] * 0.5
    model.load_state_dict(sd, assign=True)                                    # every parameter becomes another object
    moved = fwd(model)
    fresh = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).eval()
    fresh.load_state_dict(copy.deepcopy(model.state_dict()))
    want = fwd(fresh)
    assert not all(torch.equal(a, b) for a, b in zip(base, moved))
    assert all(torch.equal(a, b) for a, b in zip(moved, want))
    # swap a BatchNorm for a subclass the kernels do not serve: the verdict must flip on the next call (node-by-node path, same numbers)
    class OtherBN(torch.nn.BatchNorm1d):
        pass
    feats = batch["voxel_features"]

    def usable():
        with torch.no_grad():     # (eval with gradients enabled is the node-by-node path by design)
            return feature_pass.usable(model, feats, None)

    assert usable()
    old = model.vir_conv4.d2_conv1[1]
    new = OtherBN(old.num_features, eps=old.eps, momentum=old.momentum).to(dev).eval()
    new.load_state_dict(old.state_dict())
    setattr(model.vir_conv4.d2_conv1, "1", new)        # (SparseSequential has no __setitem__, as spconv's)
    assert not usable()
    for a, b in zip(fwd(model), want):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    setattr(model.vir_conv4.d2_conv1, "1", old)
    assert usable()
    old.momentum = None                                                       # a flag flipped in place is part of the fingerprint
    assert not usable()
