"""Host logic of the spconv facade, autograd wiring and the backbone, exercised on the CPU oracle operators."""
import numpy as np
import pytest
import torch
from torch import nn

from helpers import GRID, MODEL_CFG, fill_parameters, golden_batch, load_golden
from oracle import dense_ref
from virconv_amd import ops, spconv, synth
from virconv_amd.backbone import HeightCompression, VirConvL8x, layer_voxel_discard

SHAPE = (9, 24, 20)


def _tensor(seed=0, n=400, c=8, bs=2, dtype=torch.float32):
    idx = synth.small_scene_indices(seed, n, SHAPE, bs)
    g = torch.Generator().manual_seed(seed)
    return spconv.SparseConvTensor(torch.randn((idx.shape[0], c), generator=g, dtype=dtype), torch.from_numpy(idx), SHAPE, bs)


def test_spconv_names_and_weight_layout(oracle_backend):
    m = spconv.SubMConv3d(8, 16, 3, bias=False, indice_key="k")
    assert isinstance(m, spconv.conv.SparseConvolution) and isinstance(m, spconv.SparseModule)
    assert tuple(m.weight.shape) == (16, 3, 3, 3, 8)  # (Cout, kz, ky, kx, Cin)
    assert tuple(spconv.SparseConv3d(8, 8, (3, 1, 1), stride=(2, 1, 1)).weight.shape) == (8, 3, 1, 1, 8)
    assert tuple(spconv.SubMConv2d(4, 4, 3).weight.shape) == (4, 3, 3, 4)
    import virconv_amd.spconv.pytorch as sp2
    assert sp2.SparseConvTensor is spconv.SparseConvTensor


def test_indice_key_reuse_and_replace_feature(oracle_backend):
    x = _tensor()
    a = spconv.SubMConv3d(8, 8, 3, bias=False, indice_key="subm1")
    b = spconv.SubMConv3d(8, 8, 3, bias=False, indice_key="subm1")
    ya = a(x)
    rb = x.indice_dict["subm1"]
    yb = b(ya)
    assert yb.indice_dict["subm1"] is rb and ya.indices is x.indices
    z = ya.replace_feature(ya.features * 2)
    assert z is not ya and z.indices is ya.indices and z.indice_dict is ya.indice_dict
    # same key on a tensor with a different row count must not silently reuse
    other = _tensor(seed=5, n=300)
    other.indice_dict = x.indice_dict
    with pytest.raises(AssertionError):
        a(other)


def test_strided_conv_then_inverse_conv_roundtrip_shapes(oracle_backend):
    x = _tensor()
    down = spconv.SparseConv3d(8, 16, 3, stride=2, padding=1, bias=False, indice_key="sp")
    up = spconv.SparseInverseConv3d(16, 8, 3, indice_key="sp", bias=False)
    y = down(x)
    assert y.spatial_shape == [5, 12, 10] and y.features.shape[1] == 16
    z = up(y)
    assert z.spatial_shape == list(SHAPE) and torch.equal(z.indices, x.indices) and z.features.shape == (x.features.shape[0], 8)


def test_sequential_applies_plain_modules_to_features(oracle_backend):
    x = _tensor()
    seq = spconv.SparseSequential(spconv.SubMConv3d(8, 8, 3, bias=False, indice_key="a"), nn.BatchNorm1d(8), nn.ReLU())
    y = seq(x)
    assert isinstance(y, spconv.SparseConvTensor) and (y.features >= 0).all() and len(seq) == 3 and seq[1].num_features == 8


def test_autograd_matches_dense_autograd(oracle_backend):
    """d/dx, d/dW of SubM -> strided conv through SparseConvFunction == autograd through the dense oracle (O3)."""
    x = _tensor(dtype=torch.float64)
    f = x.features.clone().requires_grad_(True)
    c1 = spconv.SubMConv3d(8, 8, 3, bias=False, indice_key="a").double()
    c2 = spconv.SparseConv3d(8, 4, 3, stride=2, padding=1, bias=False, indice_key="b").double()
    y = c2(c1(x.replace_feature(f)))
    g = torch.randn_like(y.features)
    y.features.backward(g)
    fd = x.features.clone().requires_grad_(True)
    w1 = c1.weight.detach().clone().requires_grad_(True)
    w2 = c2.weight.detach().clone().requires_grad_(True)
    idx = x.indices.numpy()
    h = dense_ref.subm_conv(fd, idx, SHAPE, 2, w1)
    yd, oid, _ = dense_ref.sparse_conv(h, idx, SHAPE, 2, w2, 2, 1)
    np.testing.assert_array_equal(y.indices.numpy(), oid)
    yd.backward(g)
    assert torch.allclose(y.features, yd, atol=1e-10)
    assert torch.allclose(f.grad, fd.grad, atol=1e-10)
    assert torch.allclose(c1.weight.grad, w1.grad, atol=1e-9) and torch.allclose(c2.weight.grad, w2.grad, atol=1e-9)


def test_subm2d_duplicates_gradient_is_exact_transpose(oracle_backend):
    rng = np.random.default_rng(0)
    idx = np.stack([rng.integers(0, 2, 300), rng.integers(0, 10, 300), rng.integers(0, 8, 300)], 1).astype(np.int32)
    f = torch.randn(300, 4, dtype=torch.float64, requires_grad=True)
    conv = spconv.SubMConv2d(4, 4, 3, bias=False, indice_key="d").double()
    y = conv(spconv.SparseConvTensor(f, torch.from_numpy(idx), (40, 30), 2))
    assert y.indice_dict["d"].rep is not None
    assert torch.autograd.gradcheck(lambda t: conv(spconv.SparseConvTensor(t, torch.from_numpy(idx), (40, 30), 2)).features, (f,),
                                    eps=1e-6, atol=1e-6)


def test_dense_and_height_compression(oracle_backend):
    x = _tensor(c=4)
    d = x.dense()
    assert d.shape == (2, 4) + SHAPE
    assert x.dense(channels_first=False).shape == (2,) + SHAPE + (4,)
    idx = x.indices.long()
    assert torch.equal(d[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]], x.features)
    out = HeightCompression({"NUM_BEV_FEATURES": 36})({"encoded_spconv_tensor": x, "encoded_spconv_tensor_stride": 8})
    assert out["spatial_features"].shape == (2, 4 * 9, 24, 20) and out["spatial_features_stride"] == 8


def test_layer_discard_modes_and_injection(oracle_backend):
    x = _tensor(c=8)
    n = x.features.shape[0]
    perm = torch.from_numpy(np.random.default_rng(0).permutation(n))
    keep = perm[: int(n * 0.9)]
    y = layer_voxel_discard(x, 0.1, keep)
    assert torch.equal(y.features, x.features[keep]) and torch.equal(y.indices, x.indices[keep])
    assert layer_voxel_discard(x, 0, None) is x
    g = load_golden()
    for mode, expect_fewer in (("spconv1_inplace", True), ("spconv2_noop", False)):
        model = VirConvL8x(dict(MODEL_CFG, LAYER_DISCARD_MODE=mode), 8, GRID).train()
        fill_parameters(model, 7)
        with torch.no_grad():
            out = model(golden_batch(g))
        n1 = out["multi_scale_3d_features"]["x_conv1"].features.shape[0]
        assert (n1 == int(g["voxel_features"].shape[0] * 0.9)) == expect_fewer


def test_backbone_backward_runs_and_fills_all_grads(oracle_backend):
    g = load_golden()
    model = VirConvL8x(dict(MODEL_CFG), 8, GRID).train()
    fill_parameters(model, 7)
    out = model(golden_batch(g))
    loss = out["encoded_spconv_tensor"].dense().square().mean() + sum(t.features.mean() for t in out["multi_scale_3d_features"].values())
    loss.backward()
    missing = [k for k, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing
    assert out["encoded_spconv_tensor_stride"] == 8 and out["multi_scale_3d_strides"]["x_conv3"] == 4
    assert model.num_point_features == {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 64}


def test_permutation_invariance_of_input_rows(oracle_backend):
    """SubM3d -> strided conv: strided outputs come out in canonical (ascending) order whatever the input row order.
    (The 2-D image-space branch is deliberately NOT invariant: rep(c) = highest row index)"""
    x = _tensor(seed=3, n=500)
    c1 = spconv.SubMConv3d(8, 8, 3, bias=False, indice_key="a")
    c2 = spconv.SparseConv3d(8, 16, 3, stride=2, padding=1, bias=False, indice_key="b")
    perm = torch.from_numpy(np.random.default_rng(1).permutation(x.features.shape[0]))
    xp = spconv.SparseConvTensor(x.features[perm], x.indices[perm], SHAPE, 2)
    with torch.no_grad():
        y0, y1 = c2(c1(x)), c2(c1(xp))
        s0, s1 = c1(x), c1(xp)
    assert torch.equal(y0.indices, y1.indices)
    assert torch.allclose(y0.features, y1.features, atol=1e-5)
    assert torch.allclose(s0.features[perm], s1.features, atol=1e-5)  # SubM keeps the caller's row order


@pytest.mark.parametrize("bins,seed", [(2, 0), (10, 1), (10, 2), (4, 3)])
def test_front_end_host_logic_on_the_oracle_backend(oracle_backend, bins, seed):
    """Host side of the device front-end (virconv_amd.data.input_point_discard_device / frontend_voxelize / frontend_batch)
    on the oracle operators: per-bin injected permutations reproduce the reference-pinned numpy discard, and the fused call
    equals discard -> LiDAR-first concat -> voxeliser + MeanVFE done step by step.  (The HIP kernels behind the same calls
    are checked bit-exactly in tests/test_ops_gpu.py.)"""
    from oracle import geometry
    from virconv_amd import data
    fr = synth.make_frame(seed)
    pts, lidar = fr["points_virtual"], fr["points_lidar"]
    store = {}

    def perm_np(n):
        store.setdefault(n, np.random.default_rng(100 + n).permutation(n))
        return store[n]

    ref = data.input_point_discard(pts, bin_num=bins, rate=0.8, permutation=perm_np)
    parts, _, _ = data.partition(pts, num=bins, rate=1 - 0.8)
    perms = {bins - 1 - j: torch.from_numpy(perm_np(parts[j].shape[0])) for j in range(len(parts))}
    got = data.input_point_discard_device(torch.from_numpy(pts), bin_num=bins, rate=0.8, perms=perms)
    np.testing.assert_array_equal(got.numpy(), ref)
    if bins in (2, 10):
        training = bins == 2
        f, c, n = data.frontend_voxelize(torch.from_numpy(lidar), torch.from_numpy(pts), training, synth.POINT_CLOUD_RANGE,
                                         synth.VOXEL_SIZE, perms=perms)
        fused = np.concatenate([lidar, ref]).astype(np.float32)
        vox, cref, nref = geometry.voxelize(fused, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, 40000)
        np.testing.assert_array_equal(c.numpy(), cref)
        np.testing.assert_array_equal(n.numpy(), nref)
        np.testing.assert_allclose(f.numpy(), geometry.mean_vfe(vox, nref, "max"), rtol=0, atol=1e-6)
        bf, bc = data.frontend_batch([(torch.from_numpy(lidar), torch.from_numpy(pts))] * 2, training, synth.POINT_CLOUD_RANGE,
                                     synth.VOXEL_SIZE, seed=3)
        assert bc.shape[1] == 4 and bf.shape[0] == bc.shape[0] and set(bc[:, 0].tolist()) == {0, 1}


def test_front_end_refuses_host_tensors_on_the_product_backend():
    """No CPU path in the product: the HIP backend raises on host tensors instead of computing on the CPU."""
    from virconv_amd import _lib, backend_hip
    be = backend_hip.HipBackend()
    with pytest.raises(_lib.VirConvError):
        be.input_discard(torch.zeros((10, 8)), 2, 0.8)
    with pytest.raises(_lib.VirConvError):
        be.frontend_voxelize_mean(torch.zeros((10, 8)), torch.zeros((10, 8)), 2, 0.8, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE,
                                  5, 100, True)


def test_geometry_plan_uses_the_one_read_rulebook_chain_when_no_discard_is_active(oracle_backend, monkeypatch):
    """VirConvL8x.build_plan: without layer discard (inference / spconv2_noop) the four strided rulebooks come from ONE
    sparse_rulebook_chain call and the plan equals the level-by-level one; with discard the chain is not used."""
    import numpy as np
    import torch
    from helpers import GRID, MODEL_CFG, golden_batch, load_golden
    from virconv_amd import ops
    from virconv_amd.backbone import VirConvL8x

    g = load_golden()
    calls = []
    orig = type(oracle_backend).sparse_rulebook_chain
    monkeypatch.setattr(type(oracle_backend), "sparse_rulebook_chain",
                        lambda self, *a, **k: (calls.append(len(a[3])), orig(self, *a, **k))[1])

    def plan(mode, chain):
        monkeypatch.setattr(ops, "CHAIN_RULEBOOKS", chain)
        model = VirConvL8x(dict(MODEL_CFG, LAYER_DISCARD_MODE=mode), input_channels=8, grid_size=GRID).train(True)
        bd = golden_batch(g, "cpu")
        calib = ops.calib_tensor(bd["calib"], bd["voxel_features"].device) if not torch.is_tensor(bd["calib"]) else bd["calib"]
        torch.manual_seed(3)
        return model.build_plan(bd["voxel_coords"], bd["batch_size"], calib, bd.get("aug_param"), bd)

    ref = plan("spconv2_noop", False)
    assert calls == []
    got = plan("spconv2_noop", True)
    assert calls == [4]                                   # stage 2, 3, 4 down convs + conv_out in one call
    for a, b in zip(ref["stages"], got["stages"]):
        assert np.array_equal(a["out_indices"].numpy(), b["out_indices"].numpy())
        for k in a["rb3d"]:
            assert np.array_equal(a["rb3d"][k].pair_fwd.numpy(), b["rb3d"][k].pair_fwd.numpy())
    (ka, ra), (kb, rb) = next(iter(ref["conv_out"].items())), next(iter(got["conv_out"].items()))
    assert ka == kb and np.array_equal(ra.pair_fwd.numpy(), rb.pair_fwd.numpy()) and np.array_equal(ra.out_indices.numpy(), rb.out_indices.numpy())
    plan("spconv1_inplace", True)
    assert calls == [4]                                   # layer discard between the stages: per-conv rulebooks


def test_plan_splits_with_one_read_equals_the_per_slab_nonzero_form():
    """VirConv8x test-time path (spconv_backbone.py:314-337, decompose_tensor: strict begin < x < end): backbone.VirConv8x._plan_splits cuts
    every slab of x_conv3 / x_conv4 / out with one host read; same kept rows, same order, same shifted coordinates as _plan_split."""
    import torch
    from virconv_amd.backbone import VirConv8x
    g = torch.Generator().manual_seed(5)
    co = {}
    for k, (w, n) in {"x3": (1408, 5000), "x4": (704, 3000), "out": (704, 2000)}.items():
        lin = torch.randperm(2 * 5 * 40 * w, generator=g)[:n].sort().values          # ascending (b, z, y, x) order, as a strided conv emits
        b, r = lin // (5 * 40 * w), lin % (5 * 40 * w)
        z, r = r // (40 * w), r % (40 * w)
        idx = torch.stack([b, z, r // w, r % w], 1).int()
        idx[::7, 3] = (idx[::7, 3] // (w // 4)) * (w // 4)                            # plenty of x == begin rows (dropped by the strict test)
        co[k] = (idx, [5, 40, w])
    for rids in (["", "1", "2"], [""], ["", "1", "2", "3"]):
        got = VirConv8x._plan_splits(co, rids)
        for i, rid in enumerate(rids):
            for k in ("x3", "x4", "out"):
                keep, idx, shape = VirConv8x._plan_split(co[k][0], co[k][1], i)
                assert torch.equal(got[rid][k][0], keep) and torch.equal(got[rid][k][1], idx) and got[rid][k][2] == shape


def test_clip_adamw_has_no_cpu_path():
    """virconv_amd.optim.ClipAdamW is two HIP launches (vc_clip_adamw): CPU parameters are refused at construction, nothing falls back."""
    import pytest
    import torch
    from virconv_amd import optim
    p = torch.nn.Parameter(torch.zeros(8))
    assert not optim.supports([p])
    assert not optim.supports([])
    with pytest.raises(ValueError, match="CUDA"):
        optim.ClipAdamW([p])
