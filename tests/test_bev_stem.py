"""HeightCompression emitting the BEV map with the border of the FIRST BEV conv already written
(vc_to_dense_fill_padded + adapt_bev_backbone), against the reference recipe
    HeightCompression.forward (height_compression.py:27-31) -> BaseBEVBackbone block 0 = ZeroPad2d(1) + Conv2d(k3, p0) + BN + ReLU
    (base_bev_backbone.py:31-38).
CPU: on the oracle backend, with the reference's UNMODIFIED HeightCompression / BaseBEVBackbone when /root/reference is present.
GPU: the padded write-once kernel bit-exact against zero-pad of .dense(), its backward gather, and the stem against the CPU oracle."""
import importlib

import numpy as np
import pytest
import torch
from torch import nn

import refharness
from helpers import GRID, MODEL_CFG, fill_parameters, golden_batch, load_golden
from oracle import sparse_ref
from virconv_amd import ops
from virconv_amd.backbone import HeightCompression, VirConvL8x, adapt_bev_backbone

BEV_CFG = dict(LAYER_NUMS=[1, 1], LAYER_STRIDES=[1, 2], NUM_FILTERS=[64, 128], UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[128, 128])


def _first_block(cin=256, cout=64, seed=3):
    """Restatement of BaseBEVBackbone's first layers (base_bev_backbone.py:31-38) for boxes without /root/reference."""
    blk = nn.Sequential(nn.ZeroPad2d(1), nn.Conv2d(cin, cout, 3, stride=1, padding=0, bias=False),
                        nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01), nn.ReLU())
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        blk[1].weight.copy_(torch.randn(blk[1].weight.shape, generator=g) / np.sqrt(9 * cin))
        blk[2].weight.copy_(torch.rand(cout, generator=g) + 0.5)
        blk[2].bias.copy_(torch.randn(cout, generator=g) * 0.1)
    return blk.eval()


def _backbone_out(device):
    g = load_golden()
    cfg = dict(MODEL_CFG, LAYER_DISCARD_MODE="spconv2_noop")
    m = VirConvL8x(cfg, input_channels=8, grid_size=GRID).to(device)
    fill_parameters(m, 7)
    m.eval()
    with torch.no_grad():
        return m(golden_batch(g, device))


def test_padded_dense_on_the_oracle_backend_is_zero_pad_of_dense(oracle_backend):
    bd = _backbone_out("cpu")
    t = bd["encoded_spconv_tensor"]
    a = t.dense(pad=(1, 1))
    b = torch.nn.functional.pad(t.dense(), (1, 1, 1, 1))
    assert a.shape[-2:] == (202, 178) and torch.equal(a, b)


def test_height_compression_with_bev_pad_feeds_the_adapted_first_block(oracle_backend):
    bd = _backbone_out("cpu")
    blk = _first_block()
    ref = blk(HeightCompression({"NUM_BEV_FEATURES": 256})(dict(bd))["spatial_features"])
    fused_blk = nn.Sequential(nn.Identity(), *list(blk)[1:])
    hc = HeightCompression({"NUM_BEV_FEATURES": 256, "BEV_PAD": 1})
    out = hc(dict(bd))
    assert out["spatial_features"].shape == (2, 256, 202, 178) and out["spatial_features_pad"] == 1
    assert torch.equal(fused_blk(out["spatial_features"]), ref)


@pytest.mark.skipif(not refharness.available(), reason="reference tree not present (GPU box)")
def test_adapt_bev_backbone_on_the_reference_modules(oracle_backend):
    """The reference's UNMODIFIED HeightCompression + BaseBEVBackbone vs HeightCompression(BEV_PAD=1) + adapt_bev_backbone of the
    same BaseBEVBackbone (same parameters): identical st_features_2d."""
    refharness.import_reference_backbone()
    from easydict import EasyDict
    ref_hc = importlib.import_module("pcdet.models.backbones_2d.map_to_bev.height_compression")
    ref_bev = importlib.import_module("pcdet.models.backbones_2d.base_bev_backbone")
    bd = _backbone_out("cpu")
    torch.manual_seed(4)
    bev = ref_bev.BaseBEVBackbone(EasyDict(BEV_CFG), input_channels=256).eval()
    with torch.no_grad():
        want = bev(ref_hc.HeightCompression(EasyDict(NUM_BEV_FEATURES=256))(dict(bd)))["st_features_2d"].clone()
        adapt_bev_backbone(bev, pad=1)
        assert isinstance(bev.blocks[0][0], nn.Identity) and "blocks.0.1.weight" in bev.state_dict()
        got = bev(HeightCompression({"NUM_BEV_FEATURES": 256, "BEV_PAD": 1})(dict(bd)))["st_features_2d"]
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("c,shape,bs,pad", [(64, (4, 200, 176), 2, (1, 1)), (16, (5, 33, 70), 3, (2, 1)), (8, (21, 64, 48), 2, (0, 3))])
def test_padded_write_once_dense_bit_exact_and_its_backward(hip_backend, c, shape, bs, pad):
    rng = np.random.default_rng(c)
    n = 5000
    idx = np.stack([rng.integers(0, bs, n)] + [rng.integers(0, s, n) for s in shape], 1).astype(np.int32)
    idx = np.unique(idx, axis=0)
    idx = idx[rng.permutation(idx.shape[0])]
    f = rng.standard_normal((idx.shape[0], c)).astype(np.float32)
    ft = torch.from_numpy(f).cuda().requires_grad_(True)
    it = torch.from_numpy(idx).cuda()
    d = ops.to_dense(ft, it, shape, bs, pad=pad)
    want = torch.nn.functional.pad(sparse_ref.to_dense(torch.from_numpy(f), idx, shape, bs), (pad[1], pad[1], pad[0], pad[0]))
    assert d.shape == want.shape and torch.equal(d.cpu(), want)
    g = torch.from_numpy(rng.standard_normal(tuple(want.shape)).astype(np.float32)).cuda()
    (d * g).sum().backward()
    gi = g.cpu()[(torch.from_numpy(idx[:, 0]).long(), slice(None)) + tuple(
        torch.from_numpy(idx[:, a + 1]).long() + ([0] * (len(shape) - 2) + list(pad))[a] for a in range(len(shape)))]
    assert torch.equal(ft.grad.cpu(), gi)


@pytest.mark.gpu
def test_bev_stem_on_hip_equals_the_oracle_recipe(hip_backend):
    """backbone (HIP) -> HeightCompression(BEV_PAD=1) -> first BEV block without its pad module, against
    backbone (oracle, CPU) -> reference recipe (dense, ZeroPad2d, conv, BN, ReLU) on the CPU: 1e-4."""
    from oracle.backend import OracleBackend
    with ops.use_backend(OracleBackend()), torch.no_grad():
        bd_o = _backbone_out("cpu")
        blk = _first_block()
        want = blk(HeightCompression({"NUM_BEV_FEATURES": 256})(dict(bd_o))["spatial_features"]).numpy()
    bd_h = _backbone_out("cuda")
    blk_h = _first_block().cuda()
    with torch.no_grad():
        plain = blk_h(HeightCompression({"NUM_BEV_FEATURES": 256})(dict(bd_h))["spatial_features"])
        fused = nn.Sequential(nn.Identity(), *list(blk_h)[1:])(
            HeightCompression({"NUM_BEV_FEATURES": 256, "BEV_PAD": 1})(dict(bd_h))["spatial_features"])
    assert torch.equal(fused, plain)          # same conv on the same padded map: the border costs nothing in accuracy
    got = fused.cpu().numpy()
    err = np.abs(got - want)
    assert np.all(err <= 1e-4 * np.abs(want) + 1e-5 * max(1.0, np.abs(want).max())), float(err.max())


# ------------------------------------------------------------------------------------------------ f3 for real: the sparse stem
class _TinyBEV(nn.Module):
    """BaseBEVBackbone's `blocks` (base_bev_backbone.py:29-45) restated: what adapt_bev_backbone touches."""

    def __init__(self):
        super().__init__()
        blk = list(_first_block())
        blk += [nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64, eps=1e-3, momentum=0.01), nn.ReLU()]
        self.blocks = nn.ModuleList([nn.Sequential(*blk)])

    def forward(self, x):
        return self.blocks[0](x)


def test_sparse_stem_adaptation_keeps_parameters_and_results_on_the_cpu(oracle_backend):
    """adapt_bev_backbone(sparse_stem=True): same state_dict keys, the parameters still belong to the BEV backbone, and on a backend
    without the sparse kernels (the CPU oracle) the stem runs the reference recipe -- identical output."""
    bd = _backbone_out("cpu")
    torch.manual_seed(1)
    bev = _TinyBEV().eval()
    keys = list(bev.state_dict().keys())
    with torch.no_grad():
        want = bev(HeightCompression({"NUM_BEV_FEATURES": 256})(dict(bd))["spatial_features"])
    hc = HeightCompression({"NUM_BEV_FEATURES": 256})
    adapt_bev_backbone(bev, pad=1, height_compression=hc, sparse_stem=True)
    assert list(bev.state_dict().keys()) == keys and len(list(hc.parameters())) == 0
    with torch.no_grad():
        out = hc(dict(bd))
        assert out["spatial_features"].shape == (2, 64, 200, 176)
        got = bev(out["spatial_features"])
    assert torch.equal(got, want)


def test_stem_weight_packing_matches_the_height_compression_channel_order():
    """pack_stem_weight: Conv2d input channel c * D + z (the view of height_compression.py:30) -> sparse offset (z, ky, kx)."""
    from virconv_amd.bev_stem import pack_stem_weight
    w = torch.arange(64 * 256 * 9, dtype=torch.float32).view(64, 256, 3, 3)
    passes = pack_stem_weight(w, 4)
    assert [tuple(p.shape) for p in passes] == [(64, 27, 64), (64, 9, 64)]
    full = torch.cat(passes, 1)          # (C, 36, Cout)
    for (c, z, ky, kx, o) in [(0, 0, 0, 0, 0), (5, 3, 2, 1, 7), (63, 1, 1, 2, 63), (17, 2, 0, 2, 40)]:
        assert float(full[c, z * 9 + ky * 3 + kx, o]) == float(w[o, c * 4 + z, ky, kx])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eval", "train_stats_no_grad"])
def test_sparse_bev_stem_on_hip_equals_the_dense_recipe(hip_backend, mode):
    """SparseBEVStem on the HIP backend (vc_bev_pairs + gather-GEMM passes + vc_nhwc_to_nchw) against (a) the dense recipe on the
    same GPU and (b) the CPU reference recipe on the oracle backbone's output: element-wise 1e-4 (north_star tolerance), running
    statistics updated as nn.BatchNorm2d does in train mode."""
    from oracle.backend import OracleBackend
    from virconv_amd.bev_stem import SparseBEVStem
    bd_h = _backbone_out("cuda")
    t = bd_h["encoded_spconv_tensor"]
    blk_d, blk_s = _first_block().cuda(), _first_block().cuda()
    training = mode != "eval"
    blk_d.train(training), blk_s.train(training)
    stem = SparseBEVStem(blk_s)
    with torch.no_grad():
        assert stem.sparse_path_usable(t)
        got = stem(t)
        again = stem(t) if not training else None
        want = blk_d(HeightCompression({"NUM_BEV_FEATURES": 256})(dict(bd_h))["spatial_features"])
    assert got.shape == want.shape == (2, 64, 200, 176)
    err = (got - want).abs()
    assert bool((err <= 1e-4 * want.abs() + 1e-5 * max(1.0, float(want.abs().max()))).all()), float(err.max())
    if again is not None:
        assert torch.equal(got, again), "not bit-stable run to run"
    if training:
        for name in ("running_mean", "running_var"):
            a, b = getattr(blk_s[2], name), getattr(blk_d[2], name)
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), name
        assert int(blk_s[2].num_batches_tracked) == int(blk_d[2].num_batches_tracked) == 1
    else:
        with ops.use_backend(OracleBackend()), torch.no_grad():
            bd_o = _backbone_out("cpu")
            ref = _first_block()(HeightCompression({"NUM_BEV_FEATURES": 256})(dict(bd_o))["spatial_features"]).numpy()
        e = np.abs(got.cpu().numpy() - ref)
        assert np.all(e <= 1e-4 * np.abs(ref) + 1e-5 * max(1.0, np.abs(ref).max())), float(e.max())


@pytest.mark.gpu
def test_training_through_the_sparse_bev_stem_matches_the_dense_recipe_forward_and_backward(hip_backend):
    """Round 5 (VERDICT r4 "missing" #2): with a gradient required the stem no longer falls back to the dense recipe.  Forward output,
    d features, d conv weight, d gamma, d beta and the running statistics of SparseBEVStem (train mode, HIP) against the reference recipe
    -- dense() -> view -> ZeroPad2d(1) -> Conv2d -> BatchNorm2d -> ReLU (height_compression.py:27-31, base_bev_backbone.py:31-38) -- run
    in FLOAT64 on the CPU with torch autograd from the same sparse rows: 1e-4 (of the largest magnitude of each tensor)."""
    from virconv_amd.bev_stem import SparseBEVStem
    bd_h = _backbone_out("cuda")
    t = bd_h["encoded_spconv_tensor"]
    feats = t.features.detach().clone().requires_grad_(True)
    t = t.replace_feature(feats)
    blk_s = _first_block().cuda().train()
    stem = SparseBEVStem(blk_s)
    assert stem.sparse_path_usable(t)
    g = torch.Generator().manual_seed(5)
    G = torch.randn((2, 64, 200, 176), generator=g, dtype=torch.float64)
    out = stem(t)
    assert out.shape == (2, 64, 200, 176)
    (out * G.cuda().float()).sum().backward()
    # float64 reference on the CPU
    blk_r = _first_block().double().train()
    fr = feats.detach().cpu().double().requires_grad_(True)
    idx = t.indices.cpu().long()
    dense = torch.zeros((2, 4, 200, 176, 64), dtype=torch.float64)
    dense = dense.index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), fr)            # (B, D, H, W, C)
    x = dense.permute(0, 4, 1, 2, 3).reshape(2, 256, 200, 176)                           # channel = c * D + z
    ref = blk_r(x)
    (ref * G).sum().backward()

    def close(a, b, what, tol=1e-4):
        a, b = a.detach().cpu().double(), b.detach().double()
        err = float((a - b).abs().max())
        assert err <= tol * max(1.0, float(b.abs().max())), (what, err, float(b.abs().max()))

    close(out, ref, "forward")
    close(feats.grad, fr.grad, "d features")
    close(blk_s[1].weight.grad, blk_r[1].weight.grad, "d conv weight")
    close(blk_s[2].weight.grad, blk_r[2].weight.grad, "d gamma")
    close(blk_s[2].bias.grad, blk_r[2].bias.grad, "d beta")
    close(blk_s[2].running_mean, blk_r[2].running_mean, "running mean", 1e-5)
    close(blk_s[2].running_var, blk_r[2].running_var, "running var", 1e-5)
    assert int(blk_s[2].num_batches_tracked) == 1


@pytest.mark.gpu
def test_bev_pairs_backward_is_the_transpose_of_bev_pairs(hip_backend):
    """vc_bev_pairs_backward[k][i] = c  <=>  vc_bev_pairs[k][c] = i, on random voxels of a (4, 40, 36) grid."""
    from virconv_amd._lib import check, i32arr
    be = hip_backend
    rng = np.random.default_rng(2)
    bs, D, H, W = 3, 4, 40, 36
    idx = np.unique(np.stack([rng.integers(0, bs, 3000), rng.integers(0, D, 3000), rng.integers(0, H, 3000), rng.integers(0, W, 3000)], 1), axis=0).astype(np.int32)
    idx = idx[rng.permutation(idx.shape[0])]
    it = torch.from_numpy(idx).cuda()
    n, cells, kv = idx.shape[0], bs * H * W, D * 9
    fwd = torch.empty((kv, cells), dtype=torch.int32, device="cuda")
    ws = torch.empty((be.lib.vc_bev_pairs_workspace_bytes(bs, i32arr((D, H, W))),), dtype=torch.uint8, device="cuda")
    check(be.lib.vc_bev_pairs(it.data_ptr(), n, bs, i32arr((D, H, W)), 3, 3, fwd.data_ptr(), ws.data_ptr(), ws.numel(), None), "vc_bev_pairs")
    bwd = torch.empty((kv, n), dtype=torch.int32, device="cuda")
    check(be.lib.vc_bev_pairs_backward(it.data_ptr(), n, bs, i32arr((D, H, W)), 3, 3, bwd.data_ptr(), None), "vc_bev_pairs_backward")
    torch.cuda.synchronize()
    fwd, bwd = fwd.cpu().numpy(), bwd.cpu().numpy()
    k_f, c_f = np.nonzero(fwd >= 0)
    k_b, i_b = np.nonzero(bwd >= 0)
    a = set(zip(k_f.tolist(), c_f.tolist(), fwd[k_f, c_f].tolist()))
    b = set(zip(k_b.tolist(), bwd[k_b, i_b].tolist(), i_b.tolist()))
    assert a == b and len(a) > 0
