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
    # three steps later the two models still agree to fp32 rounding of the update
    for (k, a), b in zip(m0.state_dict().items(), m1.state_dict().values()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), (k, float((a - b).abs().max()))
        else:
            assert torch.equal(a, b), k
    assert torch.allclose(l0, l1, rtol=1e-5, atol=1e-6)
    # checkpoints: load_state_dict copies into the views, the flat buffer follows
    sd = {k: (v + 1 if v.dtype.is_floating_point else v) for k, v in m0.state_dict().items()}
    m1.load_state_dict(sd)
    from virconv_amd import feature_pass
    assert all(pr.flat() is not None for pr in feature_pass._training_programs(m1))
    w = next(iter(m1.parameters()))
    assert torch.equal(w, sd[next(iter(sd.keys()))])
