"""Pair-compacted forward gather-GEMM (gather_gemm_pc_kernel, csrc/conv_kernels.hip): the kernel the strided convs (and their inverses)
take -- tables with n_in != n_out, few active offsets per output row.  Per offset the block's (input row, output row) pairs are
compacted into full 16-pair MFMA tiles and the products are added into LDS output rows in ascending offset order.
Reference semantics: spconv SparseConv3d forward as used by VirConvL8x / VirConv8x (pcdet/models/backbones_3d/spconv_backbone.py:
block(..., stride=2, conv_type='spconv'), conv_out (3,1,1)/(2,1,1)); the oracle is oracle/sparse_ref.py in float64."""
import numpy as np
import pytest
import torch

from oracle import sparse_ref
from virconv_amd import synth

pytestmark = pytest.mark.gpu

SHAPE3 = (21, 64, 48)


@pytest.fixture(autouse=True)
def _pc_on(hip_backend):
    """The kernel is OFF by default (measured: it ties v2 at best, profiles/r03_pair_compacted_kernel.md); these tests turn it on."""
    from conftest import require_experiments
    require_experiments(hip_backend)
    assert hip_backend.lib.vc_debug_set(b"conv_pc", 1) == 0
    yield
    assert hip_backend.lib.vc_debug_set(b"conv_pc", 0) == 0


class _PC:
    def __init__(self, lib, v):
        self.lib, self.v = lib, v

    def __enter__(self):
        assert self.lib.vc_debug_set(b"conv_pc", self.v) == 0

    def __exit__(self, *a):
        assert self.lib.vc_debug_set(b"conv_pc", 1) == 0


def _close(got, ref64, what):
    got = got.double().cpu().numpy()
    ref64 = ref64.numpy() if hasattr(ref64, "numpy") else ref64
    scale = max(float(np.abs(ref64).max()), 1e-30)
    err = np.abs(got - ref64)
    assert np.all(err <= 1e-4 * np.abs(ref64) + 1e-5 * scale), (what, float(err.max() / scale))


def _case(be, kind, n, seed):
    idx = synth.small_scene_indices(seed, n, SHAPE3, 2)
    it = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
    if kind == "stride2":
        oi, _, pf, pb = be.sparse_rulebook(it, SHAPE3, 2, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1))
        wshape = (3, 3, 3)
    else:  # conv_out of the backbone: kernel (3,1,1), stride (2,1,1), no padding
        oi, _, pf, pb = be.sparse_rulebook(it, SHAPE3, 2, (3, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1))
        wshape = (3, 1, 1)
    return idx.shape[0], oi.shape[0], pf, pb, wshape


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 64), (64, 64), (16, 16), (64, 16), (32, 32)])
@pytest.mark.parametrize("kind,n", [("stride2", 9000), ("stride2", 30011), ("kv3", 7001)])
def test_pair_compacted_forward_matches_the_oracle_and_v2_and_is_bit_stable(hip_backend, cin, cout, kind, n):
    be, lib = hip_backend, hip_backend.lib
    rng = np.random.default_rng(cin * 131 + cout + n)
    n_in, n_out, pf, pb, wshape = _case(be, kind, n, 60 + cin)
    assert n_in != n_out
    x = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout,) + wshape + (cin,)) / 4).astype(np.float32)).cuda()
    with _PC(lib, 0):
        y_v2 = be.conv_forward(x, w, pf)
    with _PC(lib, 1):
        y = be.conv_forward(x, w, pf)
        for _ in range(4):
            assert torch.equal(y, be.conv_forward(x, w, pf))
        # the fragment-ordered weight image gives the same bits as the canonical layout
        assert lib.vc_debug_set(b"conv_autopack", 1) == 0
        try:
            assert torch.equal(y, be.conv_forward(x, w, pf))
        finally:
            assert lib.vc_debug_set(b"conv_autopack", 0) == 0
        # the inverse conv walks the backward table as a forward table (n_in and n_out swapped): same kernel
        gi = torch.from_numpy(rng.standard_normal((n_out, cin)).astype(np.float32)).cuda()
        y_inv = be.conv_forward(gi, w, pb)
    ref = sparse_ref.conv_forward(x.cpu().double(), w.cpu().double(), pf.cpu().numpy())
    _close(y, ref, "pc vs oracle")
    _close(y_v2, ref, "v2 vs oracle")
    ref_inv = sparse_ref.conv_forward(gi.cpu().double(), w.cpu().double(), pb.cpu().numpy())
    _close(y_inv, ref_inv, "pc inverse vs oracle")


@pytest.mark.parametrize("cin,cout", [(16, 32), (64, 64)])
def test_pair_compacted_statistics_and_affine_epilogues(hip_backend, cin, cout):
    be, lib = hip_backend, hip_backend.lib
    rng = np.random.default_rng(cin + 7 * cout)
    n_in, n_out, pf, _, wshape = _case(be, "stride2", 20000, 77)
    x = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout,) + wshape + (cin,)) / 4).astype(np.float32)).cuda()
    y = be.conv_forward(x, w, pf)
    y_s, partial = be.conv_forward_stats(x, w, pf)
    assert torch.equal(y, y_s)
    rows = (n_out + 127) // 128 * 8     # one partial row per 16 output rows, whole 128-row blocks
    assert partial.numel() == rows * 2 * cout
    p = partial.view(rows, 2, cout).double().sum(0).cpu()
    yd = y.double().cpu()
    assert float((p[0] - yd.sum(0)).abs().max()) <= 1e-5 * float(yd.abs().sum(0).max())
    assert float((p[1] - (yd * yd).sum(0)).abs().max()) <= 1e-5 * float((yd * yd).sum(0).max())
    # the partial rows really are per 16 output rows
    first = yd[:16].sum(0)
    assert float((partial.view(rows, 2, cout)[0, 0].double().cpu() - first).abs().max()) <= 1e-5 * max(1.0, float(first.abs().max()))
    mean = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
    var = torch.from_numpy(rng.uniform(0.5, 2.0, cout).astype(np.float32)).cuda()
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.uniform(-1, 1, cout).astype(np.float32)).cuda()
    for relu in (False, True):
        ya = be.conv_forward_affine(x, w, pf, None, mean, var, gamma, beta, 1e-3, relu)
        want = (yd - mean.double().cpu()) / torch.sqrt(var.double().cpu() + 1e-3) * gamma.double().cpu() + beta.double().cpu()
        if relu:
            want = want.clamp_min(0)
        _close(ya, want, f"affine relu={relu}")
    # train-mode unit (conv + statistics + apply) on the strided table: the BatchNorm statistics of the compacted kernel's output
    rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    yb, y_raw, m, v = be.post_act_block_forward(x, w, pf, None, "f32", False, gamma, beta, rm, rv, nbt, 0.01, 1e-3, True)
    assert torch.equal(y_raw, y)
    assert float((m.double().cpu() - yd.mean(0)).abs().max()) <= 1e-5 * max(1.0, float(yd.mean(0).abs().max()))
    assert float((v.double().cpu() - yd.var(0, unbiased=False)).abs().max()) <= 1e-5 * float(yd.var(0, unbiased=False).max())


@pytest.mark.parametrize("case", ["one_output_row", "ragged_last_block", "rows_without_pairs", "dense_offset"])
def test_pair_compacted_edge_cases(hip_backend, case):
    """Hand-made tables: a single output row; a row count that is not a multiple of 128; output rows with no pair at all (zeros);
    an offset active in EVERY row of a block (128 pairs = 8 full tiles, every wave busy) next to offsets with a single pair."""
    be = hip_backend
    rng = np.random.default_rng(5)
    cin, cout, kv = 32, 32, 27
    n_in = 5000
    n_out = {"one_output_row": 1, "ragged_last_block": 128 * 3 + 17, "rows_without_pairs": 1000, "dense_offset": 512}[case]
    tbl = np.full((kv, n_out), -1, dtype=np.int32)
    if case == "dense_offset":
        tbl[13] = rng.integers(0, n_in, n_out)
        tbl[0, 5] = 7
        tbl[26, 127] = 9
        tbl[3, 128] = 11
    elif case == "rows_without_pairs":
        act = rng.random((kv, n_out)) < 0.05
        act[:, 100:400] = False
        tbl[act] = rng.integers(0, n_in, int(act.sum()))
    else:
        act = rng.random((kv, n_out)) < 0.3
        tbl[act] = rng.integers(0, n_in, int(act.sum()))
    t = torch.from_numpy(tbl).cuda()
    x = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 4).astype(np.float32)).cuda()
    y = be.conv_forward(x, w, t)
    assert torch.equal(y, be.conv_forward(x, w, t))
    ref = sparse_ref.conv_forward(x.cpu().double(), w.cpu().double(), tbl)
    _close(y, ref, case)
    if case == "rows_without_pairs":
        assert float(y[100:400].abs().max()) == 0.0
