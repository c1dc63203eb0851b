"""fp32 products on the bf16 matrix cores (vc_debug_set f32_split, csrc/conv_kernels.hip split3): every operand is cut EXACTLY into
three bf16 values (round to nearest), six of the nine cross products are issued on v_mfma_f32_16x16x32_bf16 and accumulate in fp32; the
three dropped terms together stay below 2^-24 of a product (oracle/split_ref.py, tests/test_split_cpu.py).  These tests hold that claim against float64: the split kernels' error is the exact-fp32
kernels' error (accumulation order), not a reduced-precision error -- on unit-scale data, on data spanning 12 decades, and on the
values where truncation splits are delicate (negative numbers, powers of two, denormal-adjacent magnitudes)."""
import numpy as np
import pytest
import torch

from virconv_amd import synth

pytestmark = pytest.mark.gpu
SHAPE3 = (21, 64, 48)


def _ref64(x, w, pair):
    """y[o] = sum_k x[pair[k, o]] @ w[:, k, :]^T in float64 on the GPU (pair: (KV, N) int32, -1 = no pair)."""
    kv, n = pair.shape
    x64, w64 = x.double(), w.double().reshape(w.shape[0], kv, w.shape[-1])
    y = torch.zeros((n, w.shape[0]), dtype=torch.float64, device=x.device)
    for k in range(kv):
        idx = pair[k].long()
        ok = idx >= 0
        y[ok] += x64[idx[ok]] @ w64[:, k, :].T
    return y


def _ref64_bwd(dy, w, pair, n_in):
    """dx[j] = sum_k dy[pair[kv-1-k... mirrored table of a SubM conv]: the transpose of _ref64 written out: dx[pair[k, o]] += dy[o] @ w[:, k, :]."""
    kv, n = pair.shape
    d64, w64 = dy.double(), w.double().reshape(w.shape[0], kv, w.shape[-1])
    dx = torch.zeros((n_in, w.shape[-1]), dtype=torch.float64, device=dy.device)
    for k in range(kv):
        idx = pair[k].long()
        ok = idx >= 0
        dx.index_add_(0, idx[ok], d64[ok] @ w64[:, k, :])
    return dx


def _scaled(rng, shape, decades):
    a = rng.standard_normal(shape).astype(np.float32)
    if decades:
        a *= (10.0 ** rng.uniform(-decades / 2, decades / 2, size=(shape[0], 1))).astype(np.float32)
    return a


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 64), (64, 32), (64, 64)])
@pytest.mark.parametrize("decades", [0, 12])
def test_split_products_are_as_accurate_as_fp32_mfma(hip_backend, cin, cout, decades):
    lib = hip_backend.lib
    rng = np.random.default_rng(cin * 131 + cout + decades)
    idx = torch.from_numpy(synth.small_scene_indices(5, 6000, SHAPE3, 2)).cuda()
    n = idx.shape[0]
    pair, _ = hip_backend.subm_rulebook(idx, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    x = torch.from_numpy(_scaled(rng, (n, cin), decades)).cuda()
    g = torch.from_numpy(_scaled(rng, (n, cout), decades)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)).cuda()
    y64, dx64 = _ref64(x, w, pair), _ref64_bwd(g, w, pair, n)
    err = {}
    try:
        for split in (0, 1):
            assert lib.vc_debug_set(b"f32_split", split) == 0
            y = hip_backend.conv_forward(x, w, pair)
            dx = hip_backend.conv_backward_input(g, w, pair, n, mirror=True)
            # row-wise: every output row against ITS OWN scale (the 12-decade case has rows of very different magnitude)
            sy = y64.abs().amax(1, keepdim=True).clamp_min(1e-30)
            sd = dx64.abs().amax(1, keepdim=True).clamp_min(1e-30)
            err[split] = (float(((y.double() - y64).abs() / sy).max()), float(((dx.double() - dx64).abs() / sd).max()))
    finally:
        assert lib.vc_debug_set(b"f32_split", 1) == 0
    # fp32 accumulation of <= 27 * 64 terms: a few 1e-6 of the row's scale either way; the split must not be worse than 2x the
    # exact-product kernel's own error (+ 1 ulp)
    for e0, e1 in zip(err[0], err[1]):
        assert e1 <= 2.0 * e0 + 1.2e-7, (err, cin, cout, decades)
        assert e1 < 1e-5, err


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 16), (32, 32), (64, 32), (64, 64)])
@pytest.mark.parametrize("decades", [0, 8])
def test_split_weight_gradient_is_as_accurate_as_fp32_mfma(hip_backend, cin, cout, decades):
    """dW_k = sum over pairs of x[in]^T dy[out] (vc_debug_set bw_split: 32 pairs per v_mfma_f32_16x16x32_bf16 step) against float64."""
    lib = hip_backend.lib
    rng = np.random.default_rng(cin * 17 + cout + decades)
    idx = torch.from_numpy(synth.small_scene_indices(6, 6000, SHAPE3, 2)).cuda()
    n = idx.shape[0]
    pair, _ = hip_backend.subm_rulebook(idx, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    x = torch.from_numpy(_scaled(rng, (n, cin), decades)).cuda()
    g = torch.from_numpy(_scaled(rng, (n, cout), decades)).cuda()
    kv = pair.shape[0]
    ref = torch.zeros((cout, kv, cin), dtype=torch.float64, device="cuda")
    for k in range(kv):
        ii = pair[k].long()
        ok = ii >= 0
        ref[:, k, :] = g.double()[ok].T @ x.double()[ii[ok]]
    # every entry against the size of the sum it is (sum of |terms|): a cancelling sum has no meaningful error relative to itself
    mag = torch.zeros_like(ref)
    for k in range(kv):
        ii = pair[k].long()
        ok = ii >= 0
        mag[:, k, :] = g.double()[ok].abs().T @ x.double()[ii[ok]].abs()
    err = {}
    try:
        for split in (0, 1):
            assert lib.vc_debug_set(b"bw_split", split) == 0
            dw = hip_backend.conv_backward_weight(x, g, pair, (cout, 3, 3, 3, cin)).reshape(cout, kv, cin)
            err[split] = float(((dw.double() - ref).abs() / mag.clamp_min(1e-300)).max())
    finally:
        assert lib.vc_debug_set(b"bw_split", 1) == 0
    assert err[1] <= 2.0 * err[0] + 1.2e-7, (err, cin, cout, decades)
    assert err[1] < 1e-5, err


def test_split_is_exact_on_delicate_values(hip_backend):
    """One active pair per row, K = 16 channels of which ONE is non-zero: the conv output is a single product x * w, so the split
    kernel's result can be compared with the exact product directly: the dropped terms (<= 3 x 2^-24) and the six fp32 accumulations
    (<= 2^-24 each) bound it by 2^-21 = 4 ulp; magnitudes below ~1e-33 would lose their low piece to denormal flushing (not tested, not
    a range features or gradients live in)."""
    lib = hip_backend.lib
    vals = np.array([1.0, -1.0, 3.0, -3.0, 1.0 + 2.0 ** -23, -(1.0 + 2.0 ** -23), 2.0 - 2.0 ** -23, 0.1, -0.1, 1.0 / 3.0, 16777215.0,
                     1.0e-30, 3.0e38 / 65536, 65504.0, 2.0 ** -20, 1.9999999, 0.99999994, 123456.789], np.float32)
    rng = np.random.default_rng(0)
    n = 4096
    xs = rng.choice(vals, n).astype(np.float32)
    ws = rng.choice(vals[:10], 16).astype(np.float32)      # (the extreme magnitudes only on one side: products stay finite)
    x = np.zeros((n, 16), np.float32)
    x[:, 5] = xs
    w = np.zeros((16, 1, 1, 1, 16), np.float32)
    w[np.arange(16), 0, 0, 0, 5] = ws
    pair = torch.arange(n, dtype=torch.int32).reshape(1, n).cuda()
    try:
        assert lib.vc_debug_set(b"f32_split", 1) == 0
        y = hip_backend.conv_forward(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), pair).cpu().numpy().astype(np.float64)
    finally:
        assert lib.vc_debug_set(b"f32_split", 1) == 0
    ref = xs.astype(np.float64)[:, None] * ws.astype(np.float64)[None, :]
    rel = np.abs(y - ref) / np.abs(ref)
    assert rel.max() <= 2.0 ** -21, rel.max()


def test_split_at_the_edges_of_the_range_is_what_the_cpu_restatement_says(hip_backend):
    """The kernels at the two ends of the exponent range (oracle/split_ref.py, tests/test_split_cpu.py::test_edges_...): operands whose
    low pieces are bf16-subnormal lose those pieces (relative error of the product <= 2^-15 / 2^-7 instead of 2^-21) -- magnitudes below
    2.4e-33, not a range features or gradients live in; operands that round to a bf16 infinity give a non-finite product."""
    n = 64
    e = np.arange(-126, -62)
    xs = (np.float32(1.2345678) * np.float32(2.0) ** e).astype(np.float32)
    x = np.zeros((n, 16), np.float32)
    x[:, 3] = xs
    w = np.zeros((16, 1, 1, 1, 16), np.float32)
    w[:, 0, 0, 0, 3] = np.float32(1.7320508)
    pair = torch.arange(n, dtype=torch.int32).reshape(1, n).cuda()
    y = hip_backend.conv_forward(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), pair).cpu().numpy().astype(np.float64)[:, 0]
    ref = xs.astype(np.float64) * float(np.float32(1.7320508))
    rel = np.abs(y - ref) / np.abs(ref)
    assert np.all(rel[e >= -100] <= 2.0 ** -21), rel[e >= -100].max()
    assert np.all(rel[(e >= -109) & (e < -100)] <= 2.0 ** -14) and np.all(rel[e < -109] <= 2.0 ** -6), rel
    # infinities / values that round to a bf16 infinity: the product is not finite (the fp32-MFMA kernels return +-Inf / the finite product)
    x[:, 3] = np.float32(3.4e38)
    x[0, 3] = np.inf
    y = hip_backend.conv_forward(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), pair).cpu().numpy()[:, 0]
    assert not np.any(np.isfinite(y))
