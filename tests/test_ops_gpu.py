"""GPU parity tests proper: every HIP operator (through the C ABI) against the CPU oracle on the same seeded inputs.

Integer / index outputs are compared bit-exactly; float outputs within the north-star tolerance (1e-4 fp32).
"""
import numpy as np
import pytest
import torch

from oracle import dense_ref, geometry, sparse_ref
from oracle.backend import OracleBackend
from virconv_amd import ops, synth

pytestmark = pytest.mark.gpu

TOL = 1e-4
SHAPE3 = (21, 64, 48)


def _rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """Element-wise normalised error: `_rel_err(a, b) < TOL` means |a_i - b_i| <= TOL * (|b_i| + 0.1 * max(1, max|b|)) for EVERY
    element, i.e. rtol = TOL and atol = TOL / 10 of the tensor's scale (the former max|a - b| / max(1, max|b|) was
    blind to small-magnitude channels)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0
    return float((np.abs(a - b) / (np.abs(b) + 0.1 * max(1.0, np.abs(b).max()))).max())


def _max_err(a: np.ndarray, b: np.ndarray) -> float:
    """max-normalised error, for the reduced-precision operand experiments only: against an oracle run on the SAME rounded
    operands the difference is accumulation order, but with bf16's 8-bit mantissa the products themselves are coarse and an
    element that cancels to ~0 has no meaningful relative error."""
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def _indices3(seed, n, bs=2, shape=SHAPE3):
    return synth.small_scene_indices(seed, n, shape, bs)


def _indices2(seed, n, bs=2, shape=(160, 60), dup=True):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, bs, n)
    u = rng.integers(0, shape[0] if not dup else shape[0] // 4, n)
    v = rng.integers(0, shape[1] if not dup else shape[1] // 4, n)
    idx = np.stack([b, u, v], 1).astype(np.int32)
    if not dup:
        idx = np.unique(idx, axis=0)
        idx = idx[rng.permutation(idx.shape[0])]
    return idx


# ------------------------------------------------------------------------------------------------ rulebooks
@pytest.mark.parametrize("n", [0, 1, 63, 2000])
def test_subm_rulebook_3d_bit_exact(hip_backend, n):
    idx = _indices3(1, n) if n else np.zeros((0, 4), np.int32)
    pair, rep = hip_backend.subm_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=True)
    ref = sparse_ref.subm_rulebook(idx, SHAPE3, (3, 3, 3))
    np.testing.assert_array_equal(pair.cpu().numpy(), ref)
    np.testing.assert_array_equal(rep.cpu().numpy(), np.arange(idx.shape[0]))


def test_subm_rulebook_2d_duplicates_bit_exact(hip_backend):
    shape = (160, 60)
    idx = _indices2(2, 3000, dup=True)
    assert np.unique(idx, axis=0).shape[0] < idx.shape[0]  # duplicates present
    pair, rep = hip_backend.subm_rulebook(torch.from_numpy(idx).cuda(), shape, (3, 3), (1, 1), want_rep=True)
    np.testing.assert_array_equal(pair.cpu().numpy(), sparse_ref.subm_rulebook(idx, shape, (3, 3)))
    lut = sparse_ref.CoordLookup(idx, shape)
    np.testing.assert_array_equal(rep.cpu().numpy(), lut.find(idx[:, 0].astype(np.int64), idx[:, 1:].astype(np.int64)))


@pytest.mark.parametrize("ks,st,pd", [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
                                      ((3, 1, 1), (2, 1, 1), (0, 0, 0)), ((3, 3, 3), (1, 1, 1), (1, 1, 1))])
def test_sparse_rulebook_bit_exact(hip_backend, ks, st, pd):
    idx = _indices3(3, 2500)
    oi, osh, pf, pb = hip_backend.sparse_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, 2, ks, st, pd, (1, 1, 1))
    roi, rosh, rpf, rpb = sparse_ref.sparse_rulebook(idx, SHAPE3, 2, ks, st, pd)
    assert tuple(osh) == tuple(rosh)
    np.testing.assert_array_equal(oi.cpu().numpy(), roi)
    np.testing.assert_array_equal(pf.cpu().numpy(), rpf)
    np.testing.assert_array_equal(pb.cpu().numpy(), rpb)


def test_sparse_rulebook_empty_and_single(hip_backend):
    for idx in (np.zeros((0, 4), np.int32), np.array([[1, 0, 0, 0]], np.int32), np.array([[0, 20, 63, 47]], np.int32)):
        oi, osh, pf, pb = hip_backend.sparse_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, 2, (3, 3, 3), (2, 2, 2),
                                                      (1, 1, 1), (1, 1, 1))
        roi, rosh, rpf, rpb = sparse_ref.sparse_rulebook(idx, SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        np.testing.assert_array_equal(oi.cpu().numpy(), roi)
        np.testing.assert_array_equal(pf.cpu().numpy(), rpf)
        np.testing.assert_array_equal(pb.cpu().numpy(), rpb)


# ------------------------------------------------------------------------------------------------ convolution
CHANNELS = [(8, 8), (32, 16), (16, 16), (64, 32), (32, 32), (16, 32), (32, 64), (64, 64), (4, 16), (8, 16)]


@pytest.mark.parametrize("cin,cout", CHANNELS)
def test_subm_conv_forward_backward_vs_oracle(hip_backend, cin, cout):
    rng = np.random.default_rng(cin * 100 + cout)
    idx = _indices3(4, 3000)
    n = idx.shape[0]
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    g = rng.standard_normal((n, cout)).astype(np.float32)
    pair = sparse_ref.subm_rulebook(idx, SHAPE3, (3, 3, 3))
    xt, wt, gt, pt = (torch.from_numpy(a).cuda() for a in (x, w, g, pair))
    y = hip_backend.conv_forward(xt, wt, pt)
    y_ref = sparse_ref.conv_forward(torch.from_numpy(x).double(), torch.from_numpy(w).double(), pair).numpy()
    assert _rel_err(y.cpu().numpy(), y_ref) < TOL
    dx = hip_backend.conv_backward_input(gt, wt, pt, n, mirror=True)
    dw = hip_backend.conv_backward_weight(xt, gt, pt, w.shape)
    dx_ref, dw_ref = sparse_ref.conv_backward(torch.from_numpy(x).double(), torch.from_numpy(w).double(), pair,
                                              torch.from_numpy(g).double())
    assert _rel_err(dx.cpu().numpy(), dx_ref.numpy()) < TOL
    assert _rel_err(dw.cpu().numpy(), dw_ref.numpy()) < TOL


def test_subm_conv_matches_dense_oracle(hip_backend):
    """HIP rulebook + HIP conv against the INDEPENDENT dense conv3d oracle."""
    rng = np.random.default_rng(11)
    idx = _indices3(5, 1500)
    n = idx.shape[0]
    x = rng.standard_normal((n, 16)).astype(np.float32)
    w = (rng.standard_normal((32, 3, 3, 3, 16)) / 20).astype(np.float32)
    pair, _ = hip_backend.subm_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    y = hip_backend.conv_forward(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), pair)
    yd = dense_ref.subm_conv(torch.from_numpy(x).double(), idx, SHAPE3, 2, torch.from_numpy(w).double()).numpy()
    assert _rel_err(y.cpu().numpy(), yd) < TOL


@pytest.mark.parametrize("ks,st,pd,cin,cout", [((3, 3, 3), (2, 2, 2), (1, 1, 1), 16, 32),
                                               ((3, 3, 3), (2, 2, 2), (0, 1, 1), 64, 64),
                                               ((3, 1, 1), (2, 1, 1), (0, 0, 0), 64, 64)])
def test_strided_conv_forward_backward_vs_dense_oracle(hip_backend, ks, st, pd, cin, cout):
    rng = np.random.default_rng(5)
    idx = _indices3(6, 2500)
    n = idx.shape[0]
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((cout,) + ks + (cin,)) / np.sqrt(np.prod(ks) * cin)).astype(np.float32)
    oi, osh, pf, pb = hip_backend.sparse_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, 2, ks, st, pd, (1, 1, 1))
    xt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    y = hip_backend.conv_forward(xt, wt, pf)
    xd = torch.from_numpy(x).double().requires_grad_(True)
    wd = torch.from_numpy(w).double().requires_grad_(True)
    yd, oid, _ = dense_ref.sparse_conv(xd, idx, SHAPE3, 2, wd, st, pd)
    np.testing.assert_array_equal(oi.cpu().numpy(), oid)
    assert _rel_err(y.cpu().numpy(), yd.detach().numpy()) < TOL
    g = rng.standard_normal(tuple(yd.shape)).astype(np.float32)
    gx, gw = torch.autograd.grad(yd, (xd, wd), torch.from_numpy(g).double())
    gt = torch.from_numpy(g).cuda()
    dx = hip_backend.conv_backward_input(gt, wt, pb, n, mirror=False)
    dw = hip_backend.conv_backward_weight(xt, gt, pf, w.shape)
    assert _rel_err(dx.cpu().numpy(), gx.numpy()) < TOL
    assert _rel_err(dw.cpu().numpy(), gw.numpy()) < TOL


def test_subm2d_duplicates_forward_backward_vs_oracle(hip_backend):
    """Image-space branch: duplicate pixels (SURVEY App-A.5 rule), backward through the group-sum path."""
    rng = np.random.default_rng(9)
    shape = (160, 60)
    idx = _indices2(7, 4000, dup=True)
    n, cin, cout = idx.shape[0], 16, 16
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((cout, 3, 3, cin)) / 12).astype(np.float32)
    g = rng.standard_normal((n, cout)).astype(np.float32)
    it = torch.from_numpy(idx).cuda()
    pair, rep = hip_backend.subm_rulebook(it, shape, (3, 3), (1, 1), want_rep=True)
    xt, wt, gt = (torch.from_numpy(a).cuda() for a in (x, w, g))
    y = hip_backend.conv_forward(xt, wt, pair)
    pref = sparse_ref.subm_rulebook(idx, shape, (3, 3))
    xd, wd, gd = (torch.from_numpy(a).double() for a in (x, w, g))
    assert _rel_err(y.cpu().numpy(), sparse_ref.conv_forward(xd, wd, pref).numpy()) < TOL
    dx = hip_backend.conv_backward_input(gt, wt, pair, n, mirror=True, centre=4, rep=rep)
    dw = hip_backend.conv_backward_weight(xt, gt, pair, w.shape)
    dx_ref, dw_ref = sparse_ref.conv_backward(xd, wd, pref, gd)  # exact transpose of the forward gather
    assert _rel_err(dx.cpu().numpy(), dx_ref.numpy()) < TOL
    assert _rel_err(dw.cpu().numpy(), dw_ref.numpy()) < TOL


def test_conv_forward_is_bitwise_deterministic(hip_backend):
    rng = np.random.default_rng(3)
    idx = _indices3(8, 3000)
    x = torch.from_numpy(rng.standard_normal((idx.shape[0], 32)).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.standard_normal((32, 3, 3, 3, 32)).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((idx.shape[0], 32)).astype(np.float32)).cuda()
    pair, _ = hip_backend.subm_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    a = hip_backend.conv_forward(x, w, pair)
    b = hip_backend.conv_forward(x, w, pair)
    assert torch.equal(a, b)
    assert torch.equal(hip_backend.conv_backward_weight(x, g, pair, w.shape), hip_backend.conv_backward_weight(x, g, pair, w.shape))


def test_unsupported_channels_raise(hip_backend):
    from virconv_amd._lib import VirConvError
    pair = torch.zeros((27, 10), dtype=torch.int32, device="cuda")
    with pytest.raises(VirConvError):
        hip_backend.conv_forward(torch.zeros((10, 7), device="cuda"), torch.zeros((8, 3, 3, 3, 7), device="cuda"), pair)
    with pytest.raises(VirConvError):
        hip_backend.conv_forward(torch.zeros((10, 8)), torch.zeros((8, 3, 3, 3, 8)), pair.cpu())  # CPU tensors: no CPU path


# ------------------------------------------------------------------------------------------------ BN(+ReLU)
@pytest.mark.parametrize("c,relu", [(8, True), (16, True), (32, False), (64, True)])
def test_bn_relu_forward_backward_vs_torch(hip_backend, c, relu):
    torch.manual_seed(c)
    n = 5000
    x = (torch.randn(n, c) * 2 + 0.5).cuda()
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    bn2 = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).cuda()
    bn2.load_state_dict(bn.state_dict())
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = ops.bn_relu(xa, bn, relu)
    yb = bn2(xb)
    if relu:
        yb = torch.relu(yb)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert _rel_err(ya.detach().cpu().numpy(), yb.detach().cpu().numpy()) < TOL
    assert _rel_err(xa.grad.cpu().numpy(), xb.grad.cpu().numpy()) < TOL
    assert _rel_err(bn.weight.grad.cpu().numpy(), bn2.weight.grad.cpu().numpy()) < TOL
    assert _rel_err(bn.bias.grad.cpu().numpy(), bn2.bias.grad.cpu().numpy()) < TOL
    assert _rel_err(bn.running_mean.cpu().numpy(), bn2.running_mean.cpu().numpy()) < 1e-5
    assert _rel_err(bn.running_var.cpu().numpy(), bn2.running_var.cpu().numpy()) < 1e-5
    bn.eval(); bn2.eval()
    with torch.no_grad():
        ye = ops.bn_relu(x, bn, relu)
        yr = torch.relu(bn2(x)) if relu else bn2(x)
    assert _rel_err(ye.cpu().numpy(), yr.cpu().numpy()) < TOL


# ------------------------------------------------------------------------------------------------ projection etc.
@pytest.mark.parametrize("stride", [1, 2, 4, 8])
@pytest.mark.parametrize("with_trans", [True, False])
def test_project_uv_bit_exact(hip_backend, stride, with_trans):
    rng = np.random.default_rng(stride)
    bs, n = 3, 20000
    shape = np.array([81, 1600, 1408]) // np.array([stride if stride < 8 else 8] * 3)
    idx = np.stack([rng.integers(0, bs, n), rng.integers(0, max(shape[0], 1), n), rng.integers(0, shape[1], n),
                    rng.integers(0, shape[2], n)], 1).astype(np.int32)
    calibs = []
    for b in range(bs):
        c = synth.default_calib()
        c["P2"] = c["P2"] + rng.uniform(-1, 1, (3, 4)).astype(np.float32) * np.float32(0.01)
        calibs.append(c)
    trans = np.stack([[rng.uniform(-0.78, 0.78), float(b % 2), rng.uniform(0.95, 1.05)] for b in range(bs)]).astype(np.float32)
    tp = trans if with_trans else None
    uv_ref, depth_ref = geometry.index2uv(idx, bs, calibs, stride, tp)
    uv, depth = hip_backend.project_uv(torch.from_numpy(idx).cuda(), ops.calib_tensor(calibs, "cuda"),
                                       None if tp is None else torch.from_numpy(tp).cuda(), bs, stride, want_depth=True)
    np.testing.assert_array_equal(uv.cpu().numpy(), uv_ref)
    np.testing.assert_allclose(depth.cpu().numpy(), depth_ref, rtol=0, atol=0)


def test_gather_scatter_rows_and_dense(hip_backend):
    rng = np.random.default_rng(0)
    idx = _indices3(9, 1000)
    n = idx.shape[0]
    f = rng.standard_normal((n, 32)).astype(np.float32)
    keep = rng.permutation(n)[: int(n * 0.9)]
    fo, io = hip_backend.gather_rows(torch.from_numpy(f).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(keep).cuda())
    np.testing.assert_array_equal(fo.cpu().numpy(), f[keep])
    np.testing.assert_array_equal(io.cpu().numpy(), idx[keep])
    gi = hip_backend.scatter_rows(fo, torch.from_numpy(keep).cuda(), n).cpu().numpy()
    exp = np.zeros_like(f)
    exp[keep] = f[keep]
    np.testing.assert_array_equal(gi, exp)
    d = hip_backend.to_dense(torch.from_numpy(f).cuda(), torch.from_numpy(idx).cuda(), SHAPE3, 2)
    dref = sparse_ref.to_dense(torch.from_numpy(f), idx, SHAPE3, 2).numpy()
    np.testing.assert_array_equal(d.cpu().numpy(), dref)
    back = hip_backend.from_dense(d, torch.from_numpy(idx).cuda(), SHAPE3, 2)
    np.testing.assert_array_equal(back.cpu().numpy(), f)


@pytest.mark.parametrize("max_voxels", [40000, 3000])
def test_voxelize_mean_vs_oracle(hip_backend, max_voxels):
    fr = synth.make_frame(21, n_lidar=6000, n_virtual=9000)
    pts = np.concatenate([fr["points_lidar"], fr["points_virtual"]])
    out_of_range = np.array([[-1.0, 0, 0, 0, 0, 0, 0, 2], [10, 45.0, 0, 0, 0, 0, 0, 2], [10, 0, 1.5, 0, 0, 0, 0, 1]], np.float32)
    pts = np.concatenate([pts[:100], out_of_range, pts[100:]])
    f, c, num = hip_backend.voxelize_mean(torch.from_numpy(pts).cuda(), synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, 5,
                                          max_voxels, True)
    vox, cref, nref = geometry.voxelize(pts, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, max_voxels)
    fref = geometry.mean_vfe(vox, nref, "max")
    np.testing.assert_array_equal(c.cpu().numpy(), cref)
    np.testing.assert_array_equal(num.cpu().numpy(), nref)
    np.testing.assert_allclose(f.cpu().numpy(), fref, rtol=0, atol=1e-6)


def test_duplicate_pixel_backward_is_bitwise_deterministic(hip_backend):
    """The group sum of the duplicate-coordinate SubM backward is carried in 64-bit fixed point: re-runs are bit-equal
    even with thousands of rows clamped onto one border pixel."""
    rng = np.random.default_rng(21)
    shape = (160, 60)
    idx = _indices2(11, 20000, dup=True)
    idx[:6000, 1:] = 0  # a huge group on one pixel (what out-of-frustum voxels do)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    pair, rep = hip_backend.subm_rulebook(it, shape, (3, 3), (1, 1), want_rep=True)
    w = torch.from_numpy((rng.standard_normal((32, 3, 3, 32)) / 17).astype(np.float32)).cuda()
    g = torch.from_numpy((rng.standard_normal((n, 32)) * 1e-3).astype(np.float32)).cuda()
    a = hip_backend.conv_backward_input(g, w, pair, n, mirror=True, centre=4, rep=rep)
    for _ in range(3):
        assert torch.equal(a, hip_backend.conv_backward_input(g, w, pair, n, mirror=True, centre=4, rep=rep))
    pref = sparse_ref.subm_rulebook(idx, shape, (3, 3))
    dx_ref, _ = sparse_ref.conv_backward(torch.zeros((n, 32), dtype=torch.float64), w.cpu().double(), pref, g.cpu().double())
    assert _rel_err(a.cpu().numpy(), dx_ref.numpy()) < TOL
    z = hip_backend.conv_backward_input(torch.zeros_like(g), w, pair, n, mirror=True, centre=4, rep=rep)
    assert float(z.abs().max()) == 0.0  # all-zero gradient: scale 0 path


# ------------------------------------------------------------------------------------------------ row order (scheduling hint)
def _masks(pair: np.ndarray, rep=None, centre=-1) -> np.ndarray:
    m = np.zeros(pair.shape[1], np.int64)
    for k in range(pair.shape[0]):
        m |= (pair[k] >= 0).astype(np.int64) << k
    if rep is not None:
        m = np.where(rep != np.arange(rep.shape[0]), np.int64(1) << centre, m)
    return m


@pytest.mark.parametrize("win", [1024, 2048, 4096])
@pytest.mark.parametrize("n", [1, 100, 4096, 4097, 9000])
def test_row_order_is_windowed_stable_mask_sort(hip_backend, n, win):
    idx = synth.small_scene_indices(21, n, (21, 128, 96), 2)
    pair = sparse_ref.subm_rulebook(idx, (21, 128, 96), (3, 3, 3))
    order = hip_backend.row_order(torch.from_numpy(pair).cuda(), window=win).cpu().numpy()
    m = _masks(pair)
    want = np.concatenate([s + np.argsort(m[s:s + win], kind="stable") for s in range(0, pair.shape[1], win)])
    np.testing.assert_array_equal(order, want.astype(np.int32))


def test_row_order_with_duplicate_pixels_uses_centre_only_rows(hip_backend):
    idx = _indices2(3, 6000)
    pair, rep = sparse_ref.subm_rulebook(idx, (160, 60), (3, 3)), None
    pt = torch.from_numpy(pair).cuda()
    _, rep_t = hip_backend.subm_rulebook(torch.from_numpy(idx).cuda(), (160, 60), (3, 3), (1, 1), want_rep=True)
    rep = rep_t.cpu().numpy()
    order = hip_backend.row_order(pt, rep_t, 4, window=4096).cpu().numpy()
    m = _masks(pair, rep, 4)
    want = np.concatenate([s + np.argsort(m[s:s + 4096], kind="stable") for s in range(0, pair.shape[1], 4096)])
    np.testing.assert_array_equal(order, want.astype(np.int32))


@pytest.mark.parametrize("cin,cout", [(8, 8), (32, 16), (64, 32), (64, 64)])
def test_row_order_never_changes_conv_results(hip_backend, cin, cout):
    """The permutation is a scheduling hint: forward / backward-input are BIT-identical with and without it, for the
    sorted order and for an arbitrary permutation (subm, strided, duplicate-pixel 2-D)."""
    rng = np.random.default_rng(cin + cout)
    idx = _indices3(8, 5000)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 10).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
    pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    rand = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    for order in (hip_backend.row_order(pair), rand):
        assert torch.equal(hip_backend.conv_forward(x, w, pair), hip_backend.conv_forward(x, w, pair, order=order))
        assert torch.equal(hip_backend.conv_backward_input(g, w, pair, n, mirror=True),
                           hip_backend.conv_backward_input(g, w, pair, n, mirror=True, order=order))
    oi, _, pf, pb = hip_backend.sparse_rulebook(it, SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    go = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).cuda()
    assert torch.equal(hip_backend.conv_forward(x, w, pf), hip_backend.conv_forward(x, w, pf, order=hip_backend.row_order(pf)))
    assert torch.equal(hip_backend.conv_backward_input(go, w, pb, n, mirror=False),
                       hip_backend.conv_backward_input(go, w, pb, n, mirror=False, order=hip_backend.row_order(pb)))
    if cin == cout:
        idx2 = torch.from_numpy(_indices2(5, 4000)).cuda()
        n2 = idx2.shape[0]
        p2, rep = hip_backend.subm_rulebook(idx2, (160, 60), (3, 3), (1, 1), want_rep=True)
        w2 = torch.from_numpy((rng.standard_normal((cout, 3, 3, cin)) / 5).astype(np.float32)).cuda()
        g2 = torch.from_numpy(rng.standard_normal((n2, cout)).astype(np.float32)).cuda()
        a = hip_backend.conv_backward_input(g2, w2, p2, n2, mirror=True, centre=4, rep=rep)
        b = hip_backend.conv_backward_input(g2, w2, p2, n2, mirror=True, centre=4, rep=rep,
                                            order=hip_backend.row_order(p2, rep, 4))
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ 16-bit MFMA operands
# BASELINE configs[4] ("fp16 MFMA contraction"): the reference has no reduced-precision path, so the bound is re-declared:
#   * against the oracle with the SAME operand rounding (exact products, wide accumulation): the fp32 tolerance 1e-4
#   * against the full-precision oracle: 2e-2 relative
TOL_REDUCED = 2e-2


@pytest.mark.parametrize("operand", ["f16", "bf16"])
@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 16), (64, 32), (32, 64), (64, 64)])
def test_reduced_operand_subm_conv_vs_oracle(hip_backend, operand, cin, cout):
    rng = np.random.default_rng(cin * 7 + cout)
    idx = _indices3(14, 3000)
    n = idx.shape[0]
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    g = rng.standard_normal((n, cout)).astype(np.float32)
    pair = sparse_ref.subm_rulebook(idx, SHAPE3, (3, 3, 3))
    xt, wt, gt, pt = (torch.from_numpy(a).cuda() for a in (x, w, g, pair))
    ob = OracleBackend()
    xd, wd, gd = (torch.from_numpy(a).double() for a in (x, w, g))
    pc = torch.from_numpy(pair)
    y = hip_backend.conv_forward(xt, wt, pt, operand=operand).cpu().numpy()
    dx = hip_backend.conv_backward_input(gt, wt, pt, n, mirror=True, operand=operand).cpu().numpy()
    dw = hip_backend.conv_backward_weight(xt, gt, pt, w.shape, operand=operand).cpu().numpy()
    # same rounding, wide accumulation
    assert _max_err(y, ob.conv_forward(xd, wd, pc, operand=operand).numpy()) < TOL
    assert _max_err(dx, ob.conv_backward_input(gd, wd, pc, n, True, operand=operand).numpy()) < TOL
    assert _max_err(dw, ob.conv_backward_weight(xd, gd, pc, w.shape, operand=operand).numpy()) < TOL
    # full precision oracle
    assert _max_err(y, ob.conv_forward(xd, wd, pc).numpy()) < TOL_REDUCED
    assert _max_err(dx, ob.conv_backward_input(gd, wd, pc, n, True).numpy()) < TOL_REDUCED
    assert _max_err(dw, ob.conv_backward_weight(xd, gd, pc, w.shape).numpy()) < TOL_REDUCED
    # and the mode really is reduced precision (not silently fp32)
    y32 = hip_backend.conv_forward(xt, wt, pt).cpu().numpy()
    assert np.abs(y - y32).max() > 0


@pytest.mark.parametrize("operand", ["f16", "bf16"])
def test_reduced_operand_strided_conv_and_row_order(hip_backend, operand):
    rng = np.random.default_rng(3)
    idx = _indices3(15, 2500)
    n, cin, cout = idx.shape[0], 32, 64
    it = torch.from_numpy(idx).cuda()
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 20).astype(np.float32)).cuda()
    oi, _, pf, pb = hip_backend.sparse_rulebook(it, SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    g = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).cuda()
    ob = OracleBackend()
    y = hip_backend.conv_forward(x, w, pf, operand=operand)
    dx = hip_backend.conv_backward_input(g, w, pb, n, mirror=False, operand=operand)
    dw = hip_backend.conv_backward_weight(x, g, pf, tuple(w.shape), operand=operand)
    xd, wd, gd = x.cpu().double(), w.cpu().double(), g.cpu().double()
    assert _rel_err(y.cpu().numpy(), ob.conv_forward(xd, wd, pf.cpu(), operand=operand).numpy()) < TOL
    assert _rel_err(dx.cpu().numpy(), ob.conv_backward_input(gd, wd, pb.cpu(), n, False, operand=operand).numpy()) < TOL
    assert _rel_err(dw.cpu().numpy(), ob.conv_backward_weight(xd, gd, pf.cpu(), tuple(w.shape), operand=operand).numpy()) < TOL
    # the scheduling hint stays a pure hint in this mode too
    assert torch.equal(dx, hip_backend.conv_backward_input(g, w, pb, n, mirror=False, operand=operand,
                                                           order=hip_backend.row_order(pb)))


def test_reduced_operand_is_ignored_below_16_channels(hip_backend):
    rng = np.random.default_rng(9)
    idx = _indices3(16, 2000)
    n = idx.shape[0]
    pair = torch.from_numpy(sparse_ref.subm_rulebook(idx, SHAPE3, (3, 3, 3))).cuda()
    for cin, cout in ((8, 8), (8, 16), (16, 8)):
        x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
        w = torch.from_numpy(rng.standard_normal((cout, 3, 3, 3, cin)).astype(np.float32)).cuda()
        g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
        assert torch.equal(hip_backend.conv_forward(x, w, pair), hip_backend.conv_forward(x, w, pair, operand="f16"))
        assert torch.equal(hip_backend.conv_backward_input(g, w, pair, n, mirror=True),
                           hip_backend.conv_backward_input(g, w, pair, n, mirror=True, operand="f16"))
        assert torch.equal(hip_backend.conv_backward_weight(x, g, pair, tuple(w.shape)),
                           hip_backend.conv_backward_weight(x, g, pair, tuple(w.shape), operand="bf16"))


# ------------------------------------------------------------------------------------------------ conv epilogues (BN fusion)
@pytest.mark.parametrize("cin,cout", [(8, 8), (16, 32), (64, 32), (64, 64)])
def test_conv_stats_epilogue_gives_the_same_conv_and_batchnorm_statistics(hip_backend, cin, cout):
    rng = np.random.default_rng(cin + 3 * cout)
    idx = _indices3(31, 5000)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    assert hip_backend.conv_epilogue_supported(n, cin, cout, 27)
    y_ref = hip_backend.conv_forward(x, w, pair)
    for order in (None, hip_backend.row_order(pair)):
        y, partial = hip_backend.conv_forward_stats(x, w, pair, order=order)
        assert torch.equal(y, y_ref)                                       # the epilogue never changes the conv output
        p = partial.view(-1, 2, cout).double()
        yd = y_ref.double()
        np.testing.assert_allclose(p[:, 0].sum(0).cpu().numpy(), yd.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(p[:, 1].sum(0).cpu().numpy(), (yd * yd).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    g, b = torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda()
    rm1, rv1, rm2, rv2 = torch.zeros(cout).cuda(), torch.ones(cout).cuda(), torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    n1, n2 = torch.zeros((), dtype=torch.int64).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    y, partial = hip_backend.conv_forward_stats(x, w, pair)
    o1, m1, v1 = hip_backend.bn_forward(y, g, b, rm1, rv1, True, 0.01, 1e-3, True, num_batches_tracked=n1, partial=partial)
    o2, m2, v2 = hip_backend.bn_forward(y, g, b, rm2, rv2, True, 0.01, 1e-3, True, num_batches_tracked=n2)
    assert int(n1) == 1 and int(n2) == 1
    for a, c in ((m1, m2), (v1, v2), (rm1, rm2), (rv1, rv2), (o1, o2)):
        assert float((a - c).abs().max()) <= 1e-5 * max(1.0, float(c.abs().max()))
    # bit-stable: the per-block sums and the finalize order are fixed
    y2, partial2 = hip_backend.conv_forward_stats(x, w, pair)
    assert torch.equal(partial, partial2)


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("cin,cout", [(8, 8), (32, 16), (64, 64)])
def test_conv_affine_epilogue_equals_conv_then_eval_batchnorm(hip_backend, cin, cout, relu):
    rng = np.random.default_rng(cin * 5 + cout)
    idx = _indices3(32, 4000)
    n = idx.shape[0]
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    pair, _ = hip_backend.subm_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    g, b = torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda()
    mean, var = torch.randn(cout).cuda() * 0.2, torch.rand(cout).cuda() + 0.3
    fused = hip_backend.conv_forward_affine(x, w, pair, None, mean, var, g, b, 1e-3, relu)
    y = hip_backend.conv_forward(x, w, pair)
    ref, _, _ = hip_backend.bn_forward(y, g, b, mean, var, False, 0.01, 1e-3, relu)
    assert float((fused - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    yd = y.double()
    exact = (yd - mean.double()) / torch.sqrt(var.double() + 1e-3) * g.double() + b.double()
    if relu:
        exact = exact.clamp_min(0)
    assert float((fused.double() - exact).abs().max()) <= TOL * max(1.0, float(exact.abs().max()))


# ------------------------------------------------------------------------------------------------ random keep (layer discard)
@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 65536, 65537, 310351])
def test_random_keep_is_a_permutation_prefix(hip_backend, n):
    full = hip_backend.random_keep(n, n, 1234, "cuda").cpu().numpy()
    np.testing.assert_array_equal(np.sort(full), np.arange(n))                       # a bijection of range(n)
    n_keep = int(n * 0.9)
    part = hip_backend.random_keep(n, n_keep, 1234, "cuda").cpu().numpy()
    np.testing.assert_array_equal(part, full[:n_keep])                                # any prefix of the SAME permutation
    again = hip_backend.random_keep(n, n_keep, 1234, "cuda").cpu().numpy()
    np.testing.assert_array_equal(part, again)                                        # a function of the seed only
    if n > 100:
        other = hip_backend.random_keep(n, n, 99, "cuda").cpu().numpy()
        assert (other != full).mean() > 0.9
        # rough uniformity: the kept rows of a 50 % prefix are spread over the index range, not clustered
        half = full[: n // 2]
        quart = np.histogram(half, bins=4, range=(0, n))[0] / (n // 2)
        assert np.all(np.abs(quart - 0.25) < 0.05), quart
        assert abs(np.corrcoef(np.arange(n), full)[0, 1]) < 0.05                      # no trend between position and value


def test_layer_discard_draws_through_the_fast_generator_and_follows_the_torch_seed(hip_backend):
    from virconv_amd import backbone
    torch.manual_seed(5)
    a = backbone.draw_random_keep(5000, 4500, "cuda")
    torch.manual_seed(5)
    b = backbone.draw_random_keep(5000, 4500, "cuda")
    c = backbone.draw_random_keep(5000, 4500, "cuda")
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.dtype == torch.int64 and a.unique().numel() == 4500 and int(a.max()) < 5000 and int(a.min()) >= 0


# ------------------------------------------------------------------------------------------------ fallback kernel
@pytest.mark.parametrize("cin,cout", [(8, 8), (16, 32), (64, 32), (64, 64)])
def test_fallback_gather_gemm_kernel_matches_the_default_one(hip_backend, cin, cout):
    """gather_gemm_kernel (v1: per-wave loads, no LDS staging) is what runs when a source exceeds 2 GiB (32-bit buffer
    offsets) or KV > 32.  Forced here through vc_debug_set: same k order and MFMA sequence => bit-identical results."""
    rng = np.random.default_rng(cin + cout)
    idx = _indices3(41, 4000)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 10).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
    pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    oi, _, pf, pb = hip_backend.sparse_rulebook(it, SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    go = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).cuda()

    def run():
        return (hip_backend.conv_forward(x, w, pair), hip_backend.conv_backward_input(g, w, pair, n, mirror=True),
                hip_backend.conv_forward(x, w, pf), hip_backend.conv_backward_input(go, w, pb, n, mirror=False))

    ref = run()
    assert hip_backend.lib.vc_debug_set(b"conv_variant", 1) == 0
    try:
        assert not hip_backend.conv_epilogue_supported(n, cin, cout, 27)      # epilogues exist for the default kernel only
        alt = run()
    finally:
        assert hip_backend.lib.vc_debug_set(b"conv_variant", 2) == 0
    for a, b in zip(ref, alt):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("win", [1024, 2048, 4096])
@pytest.mark.parametrize("distinct", [1, 5, 100, 128, 129, 400])
def test_row_order_counting_and_bitonic_paths_agree_with_a_stable_sort(hip_backend, win, distinct):
    """Windows with <= 128 distinct masks take the ballot/prefix counting sort, the others the LDS bitonic network; both must
    equal numpy's stable argsort of the masks inside each window (synthetic tables with a controlled number of masks)."""
    rng = np.random.default_rng(distinct)
    n, kv = 3 * win + 77, 27
    palette = rng.choice(1 << 20, size=distinct, replace=False).astype(np.int64)
    masks = palette[rng.integers(0, distinct, n)]
    pair = np.where((masks[None, :] >> np.arange(kv)[:, None]) & 1, 7, -1).astype(np.int32)
    order = hip_backend.row_order(torch.from_numpy(pair).cuda(), window=win).cpu().numpy()
    want = np.concatenate([s + np.argsort(masks[s:s + win], kind="stable") for s in range(0, n, win)])
    np.testing.assert_array_equal(order, want.astype(np.int32))


def test_row_order_of_a_strided_backward_table_is_a_stable_mask_sort(hip_backend):
    idx = _indices3(51, 6000)
    _, _, _, pb = hip_backend.sparse_rulebook(torch.from_numpy(idx).cuda(), SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    m = _masks(pb.cpu().numpy())
    assert len(np.unique(m)) < 128                       # parity classes (+ boundaries): the counting path
    order = hip_backend.row_order(pb, window=2048).cpu().numpy()
    want = np.concatenate([s + np.argsort(m[s:s + 2048], kind="stable") for s in range(0, m.shape[0], 2048)])
    np.testing.assert_array_equal(order, want.astype(np.int32))


@pytest.mark.parametrize("max_voxels", [40000, 3000])
def test_point_to_voxel_reference_wrapper_protocol_on_hip(hip_backend, max_voxels):
    """the voxeliser side of the boundary.  The body of the reference's VoxelGeneratorWrapper.generate
    (data_processor.py:53-58, restated here because /root/reference is absent on the GPU box; the build-container test
    tests/test_reference_composition.py runs the unmodified wrapper itself):
        voxel_output = gen.point_to_voxel(tv.from_numpy(points)); voxels = tv_voxels.numpy() ...
    through the facade's Point2VoxelCPU3d + cumm.tensorview shim on HIP: zero-padded (M, 5, F) voxels, coords, counts
    bit-exact vs the oracle (incl. out-of-range points, cells with > 5 points, the max_voxels cap)."""
    from virconv_amd import spconv as facade
    facade.install(force=True)
    import cumm.tensorview as tv
    from spconv.utils import Point2VoxelCPU3d
    fr = synth.make_frame(23, n_lidar=6000, n_virtual=9000)
    pts = np.concatenate([fr["points_lidar"], fr["points_virtual"]])
    crowd = np.tile(pts[500:501], (9, 1))  # 9 more points in one existing cell: only the first 5 (input order) are kept
    crowd[:, 3] = np.arange(9, dtype=np.float32)
    out_of_range = np.array([[-1.0, 0, 0, 0, 0, 0, 0, 2], [10, 45.0, 0, 0, 0, 0, 0, 2], [10, 0, 1.5, 0, 0, 0, 0, 1]], np.float32)
    pts = np.concatenate([pts[:100], out_of_range, pts[100:], crowd]).astype(np.float32)
    gen = Point2VoxelCPU3d(vsize_xyz=list(synth.VOXEL_SIZE), coors_range_xyz=synth.POINT_CLOUD_RANGE, num_point_features=8,
                           max_num_points_per_voxel=5, max_num_voxels=max_voxels)
    tv_voxels, tv_coordinates, tv_num_points = gen.point_to_voxel(tv.from_numpy(pts))
    voxels, coordinates, num_points = tv_voxels.numpy(), tv_coordinates.numpy(), tv_num_points.numpy()
    vref, cref, nref = geometry.voxelize(pts, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, max_voxels)
    assert voxels.shape == vref.shape and voxels.shape[1:] == (5, 8)
    np.testing.assert_array_equal(coordinates, cref)
    np.testing.assert_array_equal(num_points, nref)
    np.testing.assert_array_equal(voxels, vref)
    assert nref.max() == 5 and (max_voxels != 3000 or cref.shape[0] == 3000)
    # MeanVFE('max') (mean_vfe.py:39-49) on that output == the fused voxeliser
    f, c, n = gen.point_to_voxel_mean(pts)
    np.testing.assert_array_equal(c.cpu().numpy(), cref)
    np.testing.assert_allclose(f.cpu().numpy(), geometry.mean_vfe(vref, nref, "max"), rtol=0, atol=1e-6)
    # empty input
    ev, ec, en = gen.point_to_voxel(tv.from_numpy(np.zeros((0, 8), np.float32)))
    assert ev.numpy().shape == (0, 5, 8) and ec.numpy().shape == (0, 3) and en.numpy().shape == (0,)


# ------------------------------------------------------------------------------------------------ a2 / f2: data front-end
def _bin_perms(pts, bins, rate=0.8, max_dis=60.0):
    """The permutations np.random.permutation would hand to the reference, keyed by bin id (0 = nearest)."""
    from virconv_amd import data
    parts, _, _ = data.partition(pts, num=bins, max_dis=max_dis, rate=1 - rate)
    return {bins - 1 - j: np.random.default_rng(1000 + 17 * j + parts[j].shape[0]).permutation(parts[j].shape[0])
            for j in range(len(parts))}


@pytest.mark.parametrize("bins,seed,half", [(2, 0, False), (10, 1, False), (10, 2, True), (4, 3, False), (16, 4, True), (1, 5, False)])
def test_input_point_discard_kernel_bit_exact_under_injected_permutations(hip_backend, bins, seed, half):
    """vc_input_discard vs oracle/geometry.input_point_discard_binned (pinned to the reference's own dataset.py:120-189 in
    tests/test_oracle_cpu.py) with the per-bin permutations injected: the same rows, in the same order, bit for bit --
    float32 points and the float16 rows of the depth-completion .npy files (widened exactly like .astype(float32))."""
    fr = synth.make_frame(seed)
    pts = fr["points_virtual"]
    if half:
        pts = pts.astype(np.float16).astype(np.float32)
    perms = _bin_perms(pts, bins)
    ref = geometry.input_point_discard_binned(pts, bins, 0.8, 60.0, perms)
    dev_pts = torch.from_numpy(pts.astype(np.float16) if half else pts).cuda()
    got, n_out = hip_backend.input_discard(dev_pts, bins, 0.8, 60.0, perms={b: torch.from_numpy(p) for b, p in perms.items()})
    assert int(n_out.item()) == ref.shape[0]
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_input_point_discard_kernel_edge_cases(hip_backend):
    """x < 0 belongs to no bin, points on a bin edge, an empty input, fewer points than bins, a rate of 0 (nothing dropped)."""
    rng = np.random.default_rng(3)
    pts = rng.uniform(-5, 75, size=(5000, 8)).astype(np.float32)
    pts[:40, 0] = np.repeat(np.array([0.0, 6.0, 30.0, 59.999996, 60.0, 54.0, -0.0, -1e-7], np.float32), 5)
    for bins in (2, 10):
        perms = _bin_perms(pts, bins)
        ref = geometry.input_point_discard_binned(pts, bins, 0.8, 60.0, perms)
        got, _ = hip_backend.input_discard(torch.from_numpy(pts).cuda(), bins, 0.8, 60.0,
                                           perms={b: torch.from_numpy(p) for b, p in perms.items()})
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
    e, n = hip_backend.input_discard(torch.zeros((0, 8)).cuda(), 2, 0.8)
    assert e.shape == (0, 8) and int(n.item()) == 0
    few = pts[100:103]
    ref = geometry.input_point_discard_binned(few, 10, 0.8, 60.0, _bin_perms(few, 10))
    got, _ = hip_backend.input_discard(torch.from_numpy(few).cuda(), 10, 0.8, perms={b: torch.from_numpy(p) for b, p in _bin_perms(few, 10).items()})
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    ref0 = geometry.input_point_discard_binned(pts, 2, 0.0, 60.0, _bin_perms(pts, 2, rate=0.0))
    got0, _ = hip_backend.input_discard(torch.from_numpy(pts).cuda(), 2, 0.0)
    np.testing.assert_array_equal(got0.cpu().numpy(), ref0)


def test_input_point_discard_kernel_own_generator(hip_backend):
    """Without injected permutations each reduced bin keeps per_bin DISTINCT points of that bin (a permutation prefix), the
    kept-whole bins are untouched, the total equals the reference's count, and the draw follows the seed."""
    fr = synth.make_frame(7)
    pts = fr["points_virtual"]
    for bins in (2, 10):
        ref = geometry.input_point_discard_binned(pts, bins, 0.8, 60.0, _bin_perms(pts, bins))
        a, _ = hip_backend.input_discard(torch.from_numpy(pts).cuda(), bins, 0.8, seed=11)
        b, _ = hip_backend.input_discard(torch.from_numpy(pts).cuda(), bins, 0.8, seed=11)
        c, _ = hip_backend.input_discard(torch.from_numpy(pts).cuda(), bins, 0.8, seed=12)
        assert torch.equal(a, b) and not torch.equal(a, c) and a.shape[0] == ref.shape[0] == c.shape[0]
        a = a.cpu().numpy()
        inter = 60.0 / bins
        bin_of = lambda x: np.minimum(np.floor(x / np.float32(inter)).astype(np.int64), bins - 1)
        np.testing.assert_array_equal(bin_of(a[:, 0]), bin_of(ref[:, 0]))        # same bin layout, far -> near
        src = {r.tobytes() for r in pts}
        rows = [r.tobytes() for r in a]
        assert len(set(rows)) == len(rows) and set(rows) <= src                   # distinct rows of the input
        keep_whole = np.array([r.tobytes() for r in ref]) == np.array(rows)
        assert keep_whole.mean() > 0.05                                           # the far bins are identical (kept whole)


@pytest.mark.parametrize("training,half", [(True, False), (False, True)])
def test_fused_front_end_equals_the_step_by_step_reference_pipeline(hip_backend, training, half):
    """vc_frontend_voxelize_mean (ONE call: discard + LiDAR-first concat + intensity /= 10 + voxeliser + MeanVFE) against the
    oracle pipeline dataset.py:270-294 -> data_processor.py:128-187 -> mean_vfe.py:39-49 on the same injected permutations."""
    from virconv_amd import data
    fr = synth.make_frame(9)
    lidar, virt = fr["points_lidar"], fr["points_virtual"]
    if half:
        virt = virt.astype(np.float16).astype(np.float32)
    bins = 2 if training else 10
    perms = _bin_perms(virt, bins)
    kept = geometry.input_point_discard_binned(virt, bins, 0.8, 60.0, perms)
    fused = np.concatenate([lidar, kept]).astype(np.float32)
    fused[:, 3] /= np.float32(10)
    vox, cref, nref = geometry.voxelize(fused, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, 40000)
    fref = geometry.mean_vfe(vox, nref, "max")
    dv = torch.from_numpy(virt.astype(np.float16) if half else virt).cuda()
    f, c, n = data.frontend_voxelize(torch.from_numpy(lidar).cuda(), dv, training, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE,
                                     perms={b: torch.from_numpy(p) for b, p in perms.items()}, intensity_div=10.0)
    np.testing.assert_array_equal(c.cpu().numpy(), cref)
    np.testing.assert_array_equal(n.cpu().numpy(), nref)
    np.testing.assert_allclose(f.cpu().numpy(), fref, rtol=0, atol=1e-6)
    # the batched form: all launches queued, one count read
    bf, bc = data.frontend_batch([(torch.from_numpy(lidar).cuda(), dv)] * 2, training, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, seed=5)
    assert bc.shape[1] == 4 and bf.shape[0] == bc.shape[0] and sorted(set(bc[:, 0].tolist())) == [0, 1]


def test_load_virtual_points_reads_the_fp16_npy_layout(hip_backend, tmp_path):
    from virconv_amd import data
    fr = synth.make_frame(2)
    arr = fr["points_virtual"].astype(np.float16)           # tools/PENet/vis_utils.py:148-152 writes float16 (P, 8)
    path = str(tmp_path / "000002.npy")
    np.save(path, arr)
    t = data.load_virtual_points(path, "cuda")
    assert t.dtype == torch.float16 and t.is_cuda and t.shape == arr.shape
    np.testing.assert_array_equal(t.cpu().numpy(), arr)
    out = data.input_point_discard_device(t, bin_num=2, rate=0.8, seed=1)
    assert out.dtype == torch.float32 and 0 < out.shape[0] < arr.shape[0]


# ------------------------------------------------------------------------------------------------ LDS row-window gather-GEMM (v3)
def _sorted_scene(seed=41, bs=2):
    """Coordinate-SORTED active set with surface-like occupancy (what the SubM convs of stages 2-4 see): the output of a
    stride-2 conv over voxelised synthetic frames."""
    from virconv_amd import data
    feats, coords = [], []
    for b in range(bs):
        fr = synth.make_frame(seed + b, n_lidar=6000, n_virtual=9000)
        pts = np.concatenate([fr["points_lidar"], fr["points_virtual"]])
        _, c, _ = geometry.voxelize(pts, synth.VOXEL_SIZE, synth.POINT_CLOUD_RANGE, 5, 40000)
        coords.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    idx = np.concatenate(coords)
    oi, osh, _, _ = sparse_ref.sparse_rulebook(idx, [81, 1600, 1408], bs, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    return oi, list(osh)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 16), (16, 32), (64, 32), (32, 64), (32, 32), (64, 64), (16, 8), (64, 4)])
def test_window_gather_gemm_is_bit_identical_to_the_direct_kernel_and_matches_the_oracle(hip_backend, cin, cout):
    """VC_CONV_SORTED_ROWS (LDS row windows, per-wave, with the direct-gather fall-back for runs that do not fit) is a pure
    scheduling choice: forward and backward-input are bit-identical to the direct kernel, and within 1e-4 of the oracle."""
    from conftest import require_experiments
    require_experiments(hip_backend)   # the LDS row-window kernel (v3) lives in csrc/experiments/ since round 4
    rng = np.random.default_rng(100 * cin + cout)
    idx, shape = _sorted_scene()
    n = idx.shape[0]
    assert n > 20000
    it = torch.from_numpy(idx).cuda()
    pair, _ = hip_backend.subm_rulebook(it, shape, (3, 3, 3), (1, 1, 1), want_rep=False)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    y_dir = hip_backend.conv_forward(x, w, pair)
    y_win = hip_backend.conv_forward(x, w, pair, sorted_rows=True)
    assert torch.equal(y_win, y_dir)
    assert torch.equal(y_win, hip_backend.conv_forward(x, w, pair, sorted_rows=True))      # run-to-run bit-stable
    dx_dir = hip_backend.conv_backward_input(g, w, pair, n, mirror=True)
    dx_win = hip_backend.conv_backward_input(g, w, pair, n, mirror=True, sorted_rows=True)
    assert torch.equal(dx_win, dx_dir)
    pref = sparse_ref.subm_rulebook(idx, shape, (3, 3, 3))
    yref = sparse_ref.conv_forward(x.cpu().double(), w.cpu().double(), pref)
    assert _rel_err(y_win.cpu().numpy(), yref.numpy()) < TOL
    np.testing.assert_allclose(y_win.cpu().numpy(), yref.numpy(), rtol=1e-3, atol=1e-4)   # element-wise beside the max-norm
    dxref, _ = sparse_ref.conv_backward(torch.zeros((n, cin), dtype=torch.float64), w.cpu().double(), pref, g.cpu().double())
    assert _rel_err(dx_win.cpu().numpy(), dxref.numpy()) < TOL
    np.testing.assert_allclose(dx_win.cpu().numpy(), dxref.numpy(), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("case", ["permuted", "tiny", "ragged", "kv9", "kv3", "strided_pairs"])
def test_window_gather_gemm_hint_is_safe_on_any_table(hip_backend, case):
    """The hint is only a hint: row orders that give no contiguous runs (every (tile, group) falls back to direct gathers),
    1 / 15 / 65 rows, 2-D (KV = 9) and (3,1,1) (KV = 3) kernels, a strided conv's table -- always the direct kernel's bits."""
    from conftest import require_experiments
    require_experiments(hip_backend)   # the LDS row-window kernel (v3) lives in csrc/experiments/ since round 4
    rng = np.random.default_rng(7)
    cin, cout = 32, 32
    if case in ("permuted", "tiny", "ragged"):
        idx, shape = _sorted_scene(43, 1)
        if case == "permuted":
            idx = idx[rng.permutation(idx.shape[0])]
        elif case == "tiny":
            idx = idx[:1]
        else:
            idx = idx[:65 + 15]
        it = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
        pair, _ = hip_backend.subm_rulebook(it, shape, (3, 3, 3), (1, 1, 1), want_rep=False)
        n_in = n_out = idx.shape[0]
        ksz = (3, 3, 3)
    elif case == "kv9":
        idx = _indices2(5, 6000, dup=False)
        idx = idx[np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))]
        pair, _ = hip_backend.subm_rulebook(torch.from_numpy(np.ascontiguousarray(idx)).cuda(), (160, 60), (3, 3), (1, 1), want_rep=False)
        n_in = n_out = idx.shape[0]
        ksz = (3, 3)
    else:
        idx, shape = _sorted_scene(44, 1)
        it = torch.from_numpy(idx).cuda()
        if case == "kv3":
            oi, osh, pair, _ = hip_backend.sparse_rulebook(it, shape, 1, (3, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1))
            ksz = (3, 1, 1)
        else:
            oi, osh, pair, _ = hip_backend.sparse_rulebook(it, shape, 1, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
            ksz = (3, 3, 3)
        n_in, n_out = idx.shape[0], oi.shape[0]
    x = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout,) + ksz + (cin,)) / 5).astype(np.float32)).cuda()
    y_dir = hip_backend.conv_forward(x, w, pair)
    y_win = hip_backend.conv_forward(x, w, pair, sorted_rows=True)
    assert y_win.shape == (n_out, cout) and torch.equal(y_win, y_dir)
    yref = sparse_ref.conv_forward(x.cpu().double(), w.cpu().double(), pair.cpu().numpy())
    assert _rel_err(y_win.cpu().numpy(), yref.numpy()) < TOL


@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 32), (32, 64)])
def test_window_gather_gemm_epilogues(hip_backend, cin, cout):
    """BatchNorm epilogues of the window kernel: per-WAVE partial statistics (no barrier) feed the same mean / var as the
    pass over y; the folded eval-mode BN(+ReLU) equals conv-then-BN."""
    from conftest import require_experiments
    require_experiments(hip_backend)   # the LDS row-window kernel (v3) lives in csrc/experiments/ since round 4
    rng = np.random.default_rng(cin + 7 * cout)
    idx, shape = _sorted_scene(45, 1)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    pair, _ = hip_backend.subm_rulebook(it, shape, (3, 3, 3), (1, 1, 1), want_rep=False)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    y_ref = hip_backend.conv_forward(x, w, pair)
    y, partial = hip_backend.conv_forward_stats(x, w, pair, sorted_rows=True)
    assert torch.equal(y, y_ref)
    p = partial.view(-1, 2, cout).double()
    assert p.shape[0] in (4 * ((n + 63) // 64), 8 * ((n + 127) // 128))   # one partial row per 16-row wave tile (4- or 8-wave blocks)
    yd = y_ref.double()
    np.testing.assert_allclose(p[:, 0].sum(0).cpu().numpy(), yd.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(p[:, 1].sum(0).cpu().numpy(), (yd * yd).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    g, b = torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda()
    rm1, rv1, rm2, rv2 = torch.zeros(cout).cuda(), torch.ones(cout).cuda(), torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    o1, m1, v1 = hip_backend.bn_forward(y, g, b, rm1, rv1, True, 0.01, 1e-3, True, partial=partial)
    o2, m2, v2 = hip_backend.bn_forward(y, g, b, rm2, rv2, True, 0.01, 1e-3, True)
    for a, c in ((m1, m2), (v1, v2), (rm1, rm2), (rv1, rv2), (o1, o2)):
        assert float((a - c).abs().max()) <= 1e-5 * max(1.0, float(c.abs().max()))
    _, partial2 = hip_backend.conv_forward_stats(x, w, pair, sorted_rows=True)
    assert torch.equal(partial, partial2)
    mean, var = torch.randn(cout).cuda() * 0.2, torch.rand(cout).cuda() + 0.3
    fused = hip_backend.conv_forward_affine(x, w, pair, None, mean, var, g, b, 1e-3, True, sorted_rows=True)
    assert torch.equal(fused, hip_backend.conv_forward_affine(x, w, pair, None, mean, var, g, b, 1e-3, True))


def test_backbone_marks_sorted_tables_and_uses_the_window_kernel(hip_backend, monkeypatch):
    """With VIRCONV_WINDOW_GATHER on, the geometry plan tags the SubM rulebooks of stages 2-4 (rows = output of a strided conv:
    ascending order)."""
    import bench
    from virconv_amd.backbone import VirConvL8x
    monkeypatch.setattr(ops, "WINDOW_GATHER", True)
    batch = bench.make_batch([0], torch.device("cuda", 0), training=False)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().eval()
    plan = model.build_plan(batch["voxel_coords"], 1, batch["calib"], None, batch)
    flags = [[rb.sorted_rows for rb in st["rb3d"].values()] for st in plan["stages"]]
    assert flags[0] == [False]                                   # stage 1: first-touch order
    assert all(f == [False, True] for f in flags[1:])            # strided conv table, then the SubM table on its sorted output
    assert not any(rb.sorted_rows for st in plan["stages"] for rb in st["rb2d"].values())
    for st in plan["stages"][1:]:
        oi = st["out_indices"].long()
        lin = ((oi[:, 0] * st["out_shape"][0] + oi[:, 1]) * st["out_shape"][1] + oi[:, 2]) * st["out_shape"][2] + oi[:, 3]
        assert bool((lin[1:] > lin[:-1]).all())


@pytest.mark.parametrize("wdma,winrows", [(1, 32), (0, 24), (1, 24)])
@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 32), (32, 64), (64, 64)])
def test_window_gather_gemm_pipeline_variants_keep_the_bits(hip_backend, wdma, winrows, cin, cout):
    """The experiment switches of the window kernel (W images through the LDS-DMA engine; 24-row windows) are scheduling
    variants: forward and backward-input stay bit-identical to the direct kernel, on sorted and on permuted tables."""
    from conftest import require_experiments
    require_experiments(hip_backend)   # the LDS row-window kernel (v3) lives in csrc/experiments/ since round 4
    rng = np.random.default_rng(cin * 3 + cout + wdma)
    idx, shape = _sorted_scene(47, 1)
    lib = hip_backend.lib
    try:
        assert lib.vc_debug_set(b"conv_wdma", wdma) == 0 and lib.vc_debug_set(b"conv_winrows", winrows) == 0
        for permute in (False, True):
            ii = idx[rng.permutation(idx.shape[0])] if permute else idx
            n = ii.shape[0]
            it = torch.from_numpy(np.ascontiguousarray(ii)).cuda()
            pair, _ = hip_backend.subm_rulebook(it, shape, (3, 3, 3), (1, 1, 1), want_rep=False)
            x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
            g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
            w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
            for _ in range(2):
                assert torch.equal(hip_backend.conv_forward(x, w, pair, sorted_rows=True), hip_backend.conv_forward(x, w, pair))
                assert torch.equal(hip_backend.conv_backward_input(g, w, pair, n, mirror=True, sorted_rows=True),
                                   hip_backend.conv_backward_input(g, w, pair, n, mirror=True))
    finally:
        lib.vc_debug_set(b"conv_wdma", 0)
        lib.vc_debug_set(b"conv_winrows", 32)


# ------------------------------------------------------------------------------------------------ post_act_block as one call
@pytest.mark.parametrize("kind", ["subm_sorted", "subm_first_touch", "strided", "subm2d_dup"])
def test_post_act_block_calls_equal_the_operator_by_operator_path(hip_backend, kind, monkeypatch):
    """vc_post_act_block_forward / _backward are host-side compositions of the single operators: outputs, BatchNorm
    statistics, running statistics and every gradient are bit-identical to issuing the operators one by one (the duplicate-
    pixel conv uses the persistent, self-clearing group-sum accumulator: also run twice)."""
    from torch import nn
    rng = np.random.default_rng(3)
    cin, cout = 32, 16
    if kind == "subm2d_dup":
        idx = _indices2(9, 20000, dup=True)
        it = torch.from_numpy(idx).cuda()
        rb = ops.build_subm_rulebook(it, (160, 60), (3, 3), 1, allow_duplicates=True)
        wshape = (cout, 3, 3, cin)
    else:
        idx, shape = _sorted_scene(51, 1)
        if kind == "subm_first_touch":
            idx = idx[rng.permutation(idx.shape[0])]
        it = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
        if kind == "subm_sorted":
            it._vc_sorted = True
            monkeypatch.setattr(ops, "WINDOW_GATHER", True)
        if kind == "strided":
            rb = ops.build_sparse_rulebook(it, shape, 1, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1)
        else:
            rb = ops.build_subm_rulebook(it, shape, (3, 3, 3), 1, False)
        assert rb.sorted_rows == (kind == "subm_sorted")
        wshape = (cout, 3, 3, 3, cin)
    x0 = torch.from_numpy(rng.standard_normal((rb.n_in, cin)).astype(np.float32)).cuda()
    w0 = torch.from_numpy((rng.standard_normal(wshape) / 5).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((rb.n_out, cout + 8)).astype(np.float32)).cuda()

    def run(fused):
        monkeypatch.setattr(ops, "FUSED_UNIT_CALLS", fused)
        # the unit call takes its BatchNorm statistics from the conv epilogue's per-wave partial sums: the operator-by-operator
        # reference has to do the same to be bit-comparable (a pass over y_raw rounds differently)
        monkeypatch.setattr(ops, "FUSE_BN_STATS", True)
        bn = nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, cout)); bn.bias.copy_(torch.linspace(-0.2, 0.2, cout))
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        y = ops.conv_bn_relu(x, w, rb, False, bn, True)
        wide = torch.cat([y, torch.zeros((rb.n_out, 8), device="cuda")], 1)   # the consumer is a concat: strided grad slices
        (wide * g).sum().backward()
        return [y.detach(), x.grad, w.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(),
                bn.num_batches_tracked.clone()]

    from virconv_amd import backend_hip
    # the unit call lets its conv launch finish the BatchNorm sums (another fixed summation order than the operator-by-operator
    # path's reduce + finalize kernels): this test is about the host-side composition, so both sides take the kernel route
    assert hip_backend.lib.vc_debug_set(b"conv_bn_finish", 0) == 0
    try:
        ref = run(False)
        for overlap in (False, True, True):   # the weight gradient forked onto a side stream inside the call: same bits
            monkeypatch.setattr(backend_hip, "UNIT_OVERLAP_DW", overlap)
            got = run(True)
            for a, b in zip(ref, got):
                assert torch.equal(a, b)
    finally:
        assert hip_backend.lib.vc_debug_set(b"conv_bn_finish", 1) == 0


# ------------------------------------------------------------------------------------------------ f3: write-once dense / HeightCompression
@pytest.mark.parametrize("c,shape,bs", [(64, (4, 200, 176), 2), (16, (5, 33, 70), 3), (8, (21, 64, 48), 2)])
def test_write_once_dense_equals_zero_fill_plus_scatter(hip_backend, c, shape, bs, monkeypatch):
    """vc_to_dense_fill (no zero-fill of the output, every element written once) against the oracle's dense() and against
    the scatter form; HeightCompression's (B, C*D, H, W) view on top of it (height_compression.py:27-31)."""
    from virconv_amd import backend_hip, spconv
    from virconv_amd.backbone import HeightCompression
    rng = np.random.default_rng(c)
    idx = synth.small_scene_indices(5, 3000, shape, bs)
    f = rng.standard_normal((idx.shape[0], c)).astype(np.float32)
    ft, it = torch.from_numpy(f).cuda(), torch.from_numpy(idx).cuda()
    junk = torch.full((bs, c) + shape, float("nan"), device="cuda")   # park NaNs in the allocator's free blocks
    del junk
    monkeypatch.setattr(backend_hip, "DENSE_WRITE_ONCE", True)
    d1 = hip_backend.to_dense(ft, it, shape, bs)
    monkeypatch.setattr(backend_hip, "DENSE_WRITE_ONCE", False)
    d0 = hip_backend.to_dense(ft, it, shape, bs)
    assert torch.equal(d1, d0)
    np.testing.assert_array_equal(d1.cpu().numpy(), sparse_ref.to_dense(torch.from_numpy(f), idx, shape, bs).numpy())
    monkeypatch.setattr(backend_hip, "DENSE_WRITE_ONCE", True)
    sp = spconv.SparseConvTensor(ft.clone().requires_grad_(True), it, shape, bs)
    bd = HeightCompression()({"encoded_spconv_tensor": sp, "encoded_spconv_tensor_stride": 8})
    bev = bd["spatial_features"]
    assert bev.shape == (bs, c * shape[0], shape[1], shape[2]) and torch.equal(bev.reshape(d1.shape), d1)
    g = torch.randn_like(bev)
    (bev * g).sum().backward()
    gref = g.reshape(d1.shape)[it[:, 0].long(), :, it[:, 1].long(), it[:, 2].long(), it[:, 3].long()]
    assert torch.equal(sp.features.grad, gref)


def test_write_once_dense_duplicate_coordinates_last_row_wins(hip_backend):
    idx = _indices2(3, 5000, dup=True)
    f = np.random.default_rng(0).standard_normal((idx.shape[0], 16)).astype(np.float32)
    d = hip_backend.to_dense(torch.from_numpy(f).cuda(), torch.from_numpy(idx).cuda(), (160, 60), 2)
    np.testing.assert_array_equal(d.cpu().numpy(), sparse_ref.to_dense(torch.from_numpy(f), idx, (160, 60), 2).numpy())


def test_rep_order_is_a_stable_partition_and_never_changes_the_duplicate_pixel_backward(hip_backend):
    """vc_rep_order: representatives (rep[r] == r) first, both parts in ascending row order; the backward-input of the
    duplicate-pixel conv gives the same bits with and without it."""
    rng = np.random.default_rng(12)
    shape = (160, 60)
    idx = _indices2(13, 4000, dup=True)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    pair, rep = hip_backend.subm_rulebook(it, shape, (3, 3), (1, 1), want_rep=True)
    order = hip_backend.rep_order(rep)
    r = rep.cpu().numpy()
    is_rep = r == np.arange(n)
    exp = np.concatenate([np.nonzero(is_rep)[0], np.nonzero(~is_rep)[0]])
    np.testing.assert_array_equal(order.cpu().numpy(), exp)
    assert 0.05 < is_rep.mean() < 0.95
    w = torch.from_numpy((rng.standard_normal((32, 3, 3, 32)) / 17).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, 32)).astype(np.float32)).cuda()
    a = hip_backend.conv_backward_input(g, w, pair, n, mirror=True, centre=4, rep=rep)
    b = hip_backend.conv_backward_input(g, w, pair, n, mirror=True, centre=4, rep=rep, order=order)
    assert torch.equal(a, b)
    assert hip_backend.rep_order(rep[:0]).shape == (0,)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(20, 12), (68, 36), (8, 72), (6, 10)])
def test_facade_handles_channel_counts_the_kernels_are_not_instantiated_for(hip_backend, cin, cout):
    """channel counts outside {4, 8, 16, 32, 64} (e.g. NRConvBlock(conv_depth=True): 16 + 4 input channels,
    spconv_backbone.py:172-173) are tiled into supported pieces with zero padding instead of failing: a SubM conv + BatchNorm1d +
    ReLU unit of the façade, forward and backward, against the oracle."""
    from oracle.backend import OracleBackend
    from virconv_amd import spconv
    rng = np.random.default_rng(5)
    shape = [9, 40, 36]
    n = 3000
    lin = rng.choice(2 * shape[0] * shape[1] * shape[2], size=n, replace=False)
    b, rem = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    idx = torch.from_numpy(np.stack([b, z, y, x], 1).astype(np.int32))
    feats = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32))

    def build():
        torch.manual_seed(3)
        seq = spconv.SparseSequential(spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k"),
                                      torch.nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), torch.nn.ReLU())
        return seq.train()

    def run(seq, dev):
        f = feats.clone().to(dev).detach().requires_grad_(True)
        sp = spconv.SparseConvTensor(f, idx.to(dev), shape, 2)
        out = seq(sp).features
        (out * g.to(dev)).sum().backward()
        return (out.detach().cpu(), f.grad.cpu(), seq[0].weight.grad.cpu(), seq[1].weight.grad.cpu(), seq[1].bias.grad.cpu(),
                seq[1].running_mean.cpu().clone(), seq[1].running_var.cpu().clone())

    with ops.use_backend(OracleBackend()):
        ref = run(build(), "cpu")
    got = run(build().cuda(), "cuda")
    for a, b_, name in zip(got, ref, ("out", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var")):
        tol = 2e-4 * max(1.0, float(b_.abs().max()))
        assert float((a - b_).abs().max()) <= tol, (name, float((a - b_).abs().max()), tol)


# ------------------------------------------------------------------------------------------------ chained strided rulebooks
@pytest.mark.parametrize("n", [0, 1, 700, 9000])
def test_chained_strided_rulebooks_equal_the_level_by_level_ones(hip_backend, n):
    """sparse_rulebook_chain (stage 1 + coordinate emission of every level before ONE host read; capacity buffers, device-side
    counts) returns exactly what sparse_rulebook returns level by level: the backbone's stage 2 -> 3 -> 4 -> conv_out geometry."""
    shape = (41, 160, 128)
    idx = synth.small_scene_indices(61, n, shape, 2) if n else np.zeros((0, 4), np.int32)
    it = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
    geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
             ((3, 3, 3), (2, 2, 2), (0, 1, 1), (1, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1))]
    chain = hip_backend.sparse_rulebook_chain(it, shape, 2, geoms)
    assert len(chain) == 4
    cur, cur_shape = it, shape
    for (ks, st, pd, dl), (oi, osh, pf, pb, src) in zip(geoms, chain):
        r_oi, r_osh, r_pf, r_pb = hip_backend.sparse_rulebook(cur, cur_shape, 2, ks, st, pd, dl)
        assert tuple(osh) == tuple(r_osh) and oi.is_contiguous()
        assert torch.equal(src, cur) and torch.equal(oi, r_oi) and torch.equal(pf, r_pf) and torch.equal(pb, r_pb)
        cur, cur_shape = r_oi, r_osh
    # and against the oracle for the first two levels
    if n:
        o1, s1, p1, _ = sparse_ref.sparse_rulebook(idx, list(shape), 2, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        assert np.array_equal(chain[0][0].cpu().numpy(), o1) and np.array_equal(chain[0][2].cpu().numpy(), p1)
        o2, _, p2, _ = sparse_ref.sparse_rulebook(o1, list(s1), 2, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        assert np.array_equal(chain[1][0].cpu().numpy(), o2) and np.array_equal(chain[1][2].cpu().numpy(), p2)
