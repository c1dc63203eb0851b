"""a14: checkpoint weight-layout adaptation (detector3d_template.py:350-381, spconv_utils.py:41-55)."""
import numpy as np
import pytest
import torch

from virconv_amd.backbone import VirConvL8x
from virconv_amd.checkpoint import adapt_spconv_weight, find_all_spconv_keys, load_state_dict_adapted
from tests.helpers import GRID, MODEL_CFG


def _model(seed):
    torch.manual_seed(seed)
    return VirConvL8x(dict(MODEL_CFG), 8, GRID)


def test_find_all_spconv_keys_matches_reference_walk():
    m = _model(0)
    keys = find_all_spconv_keys(m)
    sd = m.state_dict()
    assert keys and all(k in sd for k in keys)
    # every >=4-D parameter of the backbone is a sparse-conv weight and nothing else is
    assert keys == {k for k, v in sd.items() if v.dim() >= 4}
    assert "conv_out.0.weight" in keys


@pytest.mark.parametrize("layout", ["spconv2", "spconv1", "native"])
def test_load_adapts_foreign_layouts(layout):
    src, dst = _model(1), _model(2)
    keys = find_all_spconv_keys(src)
    disk = {}
    for k, v in src.state_dict().items():
        nd = v.dim()
        if k in keys and layout == "spconv1":
            v = v.permute(*range(1, nd), 0).contiguous()           # (Cout,*k,Cin) -> (*k,Cin,Cout)
        if k in keys and layout == "native":
            v = v.permute(*range(1, nd - 1), 0, nd - 1).contiguous()  # (Cout,*k,Cin) -> (*k,Cout,Cin)  spconv-2 native build
        disk[k] = v.clone()
    disk["not.in.model"] = torch.zeros(3)
    # the two foreign layouts have the same shape when Cin == Cout: 'auto' reads such a tensor as spconv 1.x, a native
    # checkpoint names its layout
    _, taken = load_state_dict_adapted(dst, disk, strict=True, source_layout="native" if layout == "native" else "auto")
    assert set(taken) == set(src.state_dict())
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k


def test_non_strict_keeps_unmatched_parameters():
    src, dst = _model(3), _model(4)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    disk = {k: v for k, v in src.state_dict().items() if k.startswith("vir_conv1.")}
    disk["vir_conv2.d3_conv1.0.weight"] = torch.zeros(1, 2, 3)        # wrong shape: skipped
    _, taken = load_state_dict_adapted(dst, disk, strict=False)
    assert set(taken) == {k for k in disk if k.startswith("vir_conv1.")}
    for k, v in dst.state_dict().items():
        ref = src.state_dict()[k] if k.startswith("vir_conv1.") else before[k]
        assert torch.equal(v, ref), k
    with pytest.raises(RuntimeError):
        load_state_dict_adapted(_model(5), disk, strict=True)


def test_adapt_is_identity_for_matching_or_unknown_shapes():
    w = torch.randn(16, 3, 3, 3, 8)
    assert adapt_spconv_weight(w, w.shape) is w
    odd = torch.randn(5, 5)
    assert adapt_spconv_weight(odd, torch.Size([16, 3, 3, 8])) is odd
    w2 = torch.randn(3, 3, 8, 16)                                   # 2-D conv, spconv1 layout
    assert torch.equal(adapt_spconv_weight(w2, torch.Size([16, 3, 3, 8])), w2.permute(3, 0, 1, 2))


def test_native_layout_is_recognised_by_shape_when_unambiguous():
    w = torch.randn(3, 3, 3, 16, 8)                                 # (*k, Cout, Cin), Cout != Cin
    got = adapt_spconv_weight(w, torch.Size([16, 3, 3, 3, 8]))
    assert torch.equal(got, w.permute(3, 0, 1, 2, 4))
    sq = torch.randn(3, 3, 3, 8, 8)                                 # ambiguous: auto = spconv 1.x, 'native' overrides
    assert torch.equal(adapt_spconv_weight(sq, torch.Size([8, 3, 3, 3, 8])), sq.permute(4, 0, 1, 2, 3))
    assert torch.equal(adapt_spconv_weight(sq, torch.Size([8, 3, 3, 3, 8]), "native"), sq.permute(3, 0, 1, 2, 4))


def test_eval_mode_batchnorm_backward_matches_torch(oracle_backend):
    """Frozen-BN fine-tuning / input-gradient probes: backward through BatchNorm1d in eval mode (+ReLU) = nn.BatchNorm1d's."""
    from virconv_amd import ops
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(8).eval()
    bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 1.5)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
    x = torch.randn(50, 8, requires_grad=True)
    y = ops.bn_relu(x, bn, True)
    g = torch.randn_like(y)
    y.backward(g)
    got = (x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
    x.grad = None; bn.weight.grad = None; bn.bias.grad = None
    torch.relu(bn(x)).backward(g)
    for a, b in zip(got, (x.grad, bn.weight.grad, bn.bias.grad)):
        assert torch.allclose(a, b, atol=1e-6)
