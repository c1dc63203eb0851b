"""GPU tests of the wave-autonomous gather-GEMM (gather_gemm_v4_kernel, vc_debug_set conv_v4): it issues the same MFMA sequence
per output row as the LDS-staged kernel (v2), so every result must be BIT-IDENTICAL to v2's -- forward, backward-input (mirrored
SubM tables, strided tables with a row order, the duplicate-pixel rule), the three epilogues -- and within the 1e-4 parity bound
of the oracle."""
import numpy as np
import pytest
import torch

import bench
from oracle import sparse_ref
from virconv_amd import synth
from virconv_amd.backbone import VirConvL8x

pytestmark = pytest.mark.gpu

TOL = 1e-4
SHAPE3 = (21, 64, 48)


@pytest.fixture(autouse=True)
def _experiments_only(hip_backend):
    """v4 / v5 / dx shift live in virconv_amd/csrc/experiments/ since round 4: these tests need a -DVC_EXPERIMENTS build."""
    from conftest import require_experiments
    require_experiments(hip_backend)


def _rel_err(a, b):
    """element-wise: < TOL means |a_i - b_i| <= TOL * (|b_i| + 0.1 * max(1, max|b|)) for every element (see test_ops_gpu.py)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0
    return float((np.abs(a - b) / (np.abs(b) + 0.1 * max(1.0, np.abs(b).max()))).max())


class _V4:
    """with _V4(lib, 1): ... -- force the kernel choice, restore the library default (0 = the LDS-staged kernel) afterwards"""

    def __init__(self, lib, value):
        self.lib, self.value = lib, value

    def __enter__(self):
        assert self.lib.vc_debug_set(b"conv_v4", self.value) == 0

    def __exit__(self, *a):
        assert self.lib.vc_debug_set(b"conv_v4", 0) == 0


def _both(lib, fn):
    """ref = the LDS-staged kernel reading the canonical weight layout; got / again = the wave-autonomous kernel; every other
    combination (either kernel with a fragment-ordered weight image) is compared with ref right here, bit for bit."""
    def same(a, b):
        if isinstance(a, (tuple, list)):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                same(x, y)
        elif isinstance(a, dict):
            for k in a:
                same(a[k], b[k])
        elif torch.is_tensor(a):
            assert a.shape == b.shape and torch.equal(a, b)
    with _V4(lib, 0):
        ref = fn()
        assert lib.vc_debug_set(b"conv_autopack", 1) == 0
        try:
            same(ref, fn())
        finally:
            assert lib.vc_debug_set(b"conv_autopack", 0) == 0
    with _V4(lib, 1):
        got = fn()
        again = fn()
        assert lib.vc_debug_set(b"conv_autopack", 1) == 0
        try:
            same(got, fn())
        finally:
            assert lib.vc_debug_set(b"conv_autopack", 0) == 0
    return ref, got, again


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 16), (16, 32), (64, 32), (32, 64), (32, 32), (64, 64), (64, 16), (16, 64)])
def test_v4_subm_and_strided_bit_identical_to_v2_and_within_tolerance_of_the_oracle(hip_backend, cin, cout):
    rng = np.random.default_rng(1000 * cin + cout)
    idx = synth.small_scene_indices(51, 9000, SHAPE3, 2)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
    pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    oi, _, pf, pb = hip_backend.sparse_rulebook(it, SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    go = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).cuda()
    order = hip_backend.row_order(pb)

    def run():
        return (hip_backend.conv_forward(x, w, pair), hip_backend.conv_backward_input(g, w, pair, n, mirror=True),
                hip_backend.conv_forward(x, w, pf), hip_backend.conv_backward_input(go, w, pb, n, mirror=False),
                hip_backend.conv_backward_input(go, w, pb, n, mirror=False, order=order))

    ref, got, again = _both(hip_backend.lib, run)
    for a, b, c in zip(ref, got, again):
        assert torch.equal(a, b) and torch.equal(b, c)
    pref = sparse_ref.subm_rulebook(idx, SHAPE3, (3, 3, 3))
    yref = sparse_ref.conv_forward(x.cpu().double(), w.cpu().double(), pref)
    assert _rel_err(got[0].cpu().numpy(), yref.numpy()) < TOL
    np.testing.assert_allclose(got[0].cpu().numpy(), yref.numpy(), rtol=1e-3, atol=1e-4)
    dxref, _ = sparse_ref.conv_backward(torch.zeros((n, cin), dtype=torch.float64), w.cpu().double(), pref, g.cpu().double())
    assert _rel_err(got[1].cpu().numpy(), dxref.numpy()) < TOL
    ysref = sparse_ref.conv_forward(x.cpu().double(), w.cpu().double(), pf.cpu().numpy())
    assert _rel_err(got[2].cpu().numpy(), ysref.numpy()) < TOL


@pytest.mark.parametrize("case", ["one_row", "ragged", "kv3", "kv9_duplicates", "empty_offsets"])
def test_v4_edge_cases_bit_identical_to_v2(hip_backend, case):
    rng = np.random.default_rng(11)
    cin, cout = 32, 32
    lib = hip_backend.lib
    if case in ("one_row", "ragged", "empty_offsets"):
        idx = synth.small_scene_indices(52, 3000, SHAPE3, 1, surface=(case != "empty_offsets"))
        idx = idx[:1] if case == "one_row" else (idx[:64 + 17] if case == "ragged" else idx[:300])
        it = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
        pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
        n = idx.shape[0]
        x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
        ref, got, _ = _both(lib, lambda: (hip_backend.conv_forward(x, w, pair),
                                          hip_backend.conv_backward_input(x, w, pair, n, mirror=True)))
    elif case == "kv3":
        idx = synth.small_scene_indices(53, 6000, SHAPE3, 2)
        it = torch.from_numpy(idx).cuda()
        oi, _, pf, pb = hip_backend.sparse_rulebook(it, SHAPE3, 2, (3, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1))
        n = idx.shape[0]
        x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
        go = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((cout, 3, 1, 1, cin)) / 3).astype(np.float32)).cuda()
        ref, got, _ = _both(lib, lambda: (hip_backend.conv_forward(x, w, pf),
                                          hip_backend.conv_backward_input(go, w, pb, n, mirror=False)))
    else:  # 2-D image-space SubM over duplicate pixels: centre-only rows, group-summed dy + the row's own dy on the centre tap
        b = rng.integers(0, 2, 7000)
        u, v = rng.integers(0, 40, 7000), rng.integers(0, 15, 7000)
        idx = np.stack([b, u, v], 1).astype(np.int32)
        it = torch.from_numpy(idx).cuda()
        pair, rep = hip_backend.subm_rulebook(it, (160, 60), (3, 3), (1, 1), want_rep=True)
        n = idx.shape[0]
        x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
        g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((cout, 3, 3, cin)) / 3).astype(np.float32)).cuda()
        ref, got, _ = _both(lib, lambda: (hip_backend.conv_forward(x, w, pair),
                                          hip_backend.conv_backward_input(g, w, pair, n, mirror=True, centre=4, rep=rep)))
    for a, b_ in zip(ref, got):
        assert a.shape == b_.shape and torch.equal(a, b_)


@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 32), (32, 64), (64, 64)])
def test_v4_forward_epilogues_bit_identical_to_v2(hip_backend, cin, cout):
    rng = np.random.default_rng(cin + 13 * cout)
    idx = synth.small_scene_indices(54, 7000, SHAPE3, 2)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    gm, bt = torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda()
    mean, var = torch.randn(cout).cuda() * 0.2, torch.rand(cout).cuda() + 0.3

    def run():
        y, partial = hip_backend.conv_forward_stats(x, w, pair)
        return y, partial, hip_backend.conv_forward_affine(x, w, pair, None, mean, var, gm, bt, 1e-3, True)

    ref, got, again = _both(hip_backend.lib, run)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[2], got[2])
    # the partial rows are per 16-row tile in both kernels; an 8-wave v2 block pads the row count to a multiple of 8
    pr, pg = ref[1].view(-1, 2, cout), got[1].view(-1, 2, cout)
    m = min(pr.shape[0], pg.shape[0])
    assert m >= (n + 15) // 16 and torch.equal(pr[:m], pg[:m])
    assert torch.equal(got[1], again[1])


@pytest.mark.parametrize("discard", ["spconv1_inplace"])
def test_v4_whole_train_step_equals_v2(hip_backend, discard, monkeypatch):
    """bench.MODEL_CFG through the native feature pass with every eligible conv on v4 (STATS epilogue forward, BWD epilogue
    backward): outputs bit-equal to the v2 run, gradients equal up to the fold order of the partial rows."""
    dev = torch.device("cuda", 0)
    batch = bench.make_batch([0, 1], dev, training=True)
    lw = bench.make_loss_weights(dev)
    cfg = dict(bench.MODEL_CFG)
    cfg["LAYER_DISCARD_MODE"] = discard
    torch.manual_seed(33)
    model = VirConvL8x(cfg, 8, synth.GRID_SIZE).to(dev).train()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def one():
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(5)       # the layer discard draws from torch's generator
        bd = dict(batch)
        bd["voxel_features"] = batch["voxel_features"].clone()
        out = model(bd)
        loss = (out["encoded_spconv_tensor"].dense() * lw["dense"]).sum()
        for name, t in out["multi_scale_3d_features"].items():
            loss = loss + (t.features * lw[name]).sum()
        loss.backward()
        res = {nm: t.features.detach().clone() for nm, t in out["multi_scale_3d_features"].items()}
        res["out"] = out["encoded_spconv_tensor"].features.detach().clone()
        return float(loss.detach()), res, {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    lib = hip_backend.lib
    # v4 leaves its partial rows to the BatchNorm kernels; v2 would finish the sums in its own launch (another, equally fixed,
    # summation order): compare the two conv kernels on the same route
    assert lib.vc_debug_set(b"conv_bn_finish", 0) == 0
    try:
        _run_v4_vs_v2(lib, one)
    finally:
        assert lib.vc_debug_set(b"conv_bn_finish", 1) == 0


def _run_v4_vs_v2(lib, one):
    with _V4(lib, 0):
        ref = one()                                   # v2, fragment-ordered weight images (the pass packs them)
        assert lib.vc_debug_set(b"conv_packed", 0) == 0
        try:
            canon = one()                             # v2, canonical weight layout
        finally:
            assert lib.vc_debug_set(b"conv_packed", 1) == 0
    assert ref[0] == canon[0]
    for k in ref[1]:
        assert torch.equal(ref[1][k], canon[1][k]), k
    for k in ref[2]:
        assert torch.equal(ref[2][k], canon[2][k]), k
    with _V4(lib, 1):
        got = one()
        again = one()
    assert ref[0] == got[0]
    for k in ref[1]:
        assert torch.equal(ref[1][k], got[1][k]), k
    for k in ref[2]:
        tol = 1e-5 * max(float(ref[2][k].abs().max()), 1e-30)
        assert float((ref[2][k] - got[2][k]).abs().max()) <= tol, k
        assert torch.equal(got[2][k], again[2][k]), k


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 64), (32, 32), (16, 64), (64, 16)])
def test_v5_loader_consumer_kernel_bit_identical_to_v2(hip_backend, cin, cout):
    """gather_gemm_v5_kernel (loader waves + MFMA-only waves, quad-mapped gathers through swizzled LDS stages; a developer
    experiment behind vc_debug_set conv_v5, plain launches with a weight image): same MFMA sequence per row as v2."""
    rng = np.random.default_rng(7 * cin + cout)
    lib = hip_backend.lib
    idx = synth.small_scene_indices(57, 9000, SHAPE3, 2)
    n = idx.shape[0]
    it = torch.from_numpy(idx).cuda()
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
    pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
    oi, _, pf, pb = hip_backend.sparse_rulebook(it, SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
    go = torch.from_numpy(rng.standard_normal((oi.shape[0], cout)).astype(np.float32)).cuda()

    def run():
        return (hip_backend.conv_forward(x, w, pair), hip_backend.conv_backward_input(g, w, pair, n, mirror=True),
                hip_backend.conv_forward(x, w, pf), hip_backend.conv_backward_input(go, w, pb, n, mirror=False),
                hip_backend.conv_forward(x[:70], w, pair[:, :70].contiguous().clamp(max=69)))

    ref = run()
    assert lib.vc_debug_set(b"conv_autopack", 1) == 0 and lib.vc_debug_set(b"conv_v5", 1) == 0
    try:
        got = run()
    finally:
        assert lib.vc_debug_set(b"conv_autopack", 0) == 0 and lib.vc_debug_set(b"conv_v5", 0) == 0
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 32), (32, 64), (64, 64)])
def test_dx_shift_variant_is_bit_identical(hip_backend, cin, cout):
    """conv_dxs (the dx = +-1 fragments of a 27-offset SubM table taken from the group's centre fragment by a DPP lane shift where
    the table says so, gathered otherwise): same operands, same MFMA order -- bit-identical on a coordinate-sorted tensor (where
    most lanes shift) and on a row-permuted one (where almost none does).  A developer experiment, off by default."""
    rng = np.random.default_rng(31 * cin + cout)
    lib = hip_backend.lib
    idx = synth.small_scene_indices(58, 9000, SHAPE3, 2)
    srt = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    for tab in (srt, idx):
        n = tab.shape[0]
        it = torch.from_numpy(np.ascontiguousarray(tab)).cuda()
        pair, _ = hip_backend.subm_rulebook(it, SHAPE3, (3, 3, 3), (1, 1, 1), want_rep=False)
        x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
        g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()

        def run():
            y, partial = hip_backend.conv_forward_stats(x, w, pair)
            return hip_backend.conv_forward(x, w, pair), hip_backend.conv_backward_input(g, w, pair, n, mirror=True), y, partial

        assert lib.vc_debug_set(b"conv_autopack", 1) == 0
        try:
            ref = run()
            assert lib.vc_debug_set(b"conv_dxs", 1) == 0
            got = run()
        finally:
            assert lib.vc_debug_set(b"conv_dxs", 0) == 0 and lib.vc_debug_set(b"conv_autopack", 0) == 0
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
