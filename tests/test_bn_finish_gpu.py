"""In-kernel finish of the BatchNorm sums (conv_finish_tail in csrc/conv_kernels.hip): the gather-GEMM's STATS / BWD epilogue carries
its partial sums to the finished per-channel numbers in three levels with arrival tickets instead of leaving per-wave partial rows to
two more launches.  vc_debug_set conv_bn_finish: 1 = in the conv launch, 0 = the two-launch route (partial-row reduce + finalize),
2 = the SAME three levels by a single block after the conv launch (bn_finish_reference_kernel).  1 must be BIT-identical to 2 (any
stale read, lost ticket or wrong group would show) and run-to-run stable, and equal to 0 up to the fp32 rounding of the sums.
Reference semantics: nn.BatchNorm1d in training mode inside post_act_block (pcdet/models/backbones_3d/spconv_backbone.py:86-107)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEFAULT_FINISH = 1   # the library default (restored after every test)


def _finish_launches(lib) -> int:
    v = C.c_int64(0)
    assert lib.vc_debug_get(b"conv_bn_finish_launches", C.byref(v)) == 0
    return int(v.value)


def _scene(rng, n, shape, batch):
    """n distinct voxels of a (batch, *shape) grid, rows in ascending (b, z, y, x) order (as the voxeliser emits them)"""
    cells = int(np.prod(shape)) * batch
    lin = np.sort(rng.choice(cells, size=n, replace=False))
    idx = np.stack(np.unravel_index(lin, (batch,) + tuple(shape)), 1).astype(np.int32)
    return np.ascontiguousarray(idx)


@pytest.mark.parametrize("cin,cout,n", [(8, 8, 20011), (16, 32, 40009), (32, 64, 33333), (64, 64, 25000), (4, 16, 9000),
                                        (64, 32, 131072), (32, 32, 8192 + 64)])
def test_forward_unit_statistics_finished_in_the_conv_launch_are_bit_identical(hip_backend, cin, cout, n):
    be, lib = hip_backend, hip_backend.lib
    rng = np.random.default_rng(cin * 100 + cout)
    shape = (21, 200, 176)
    idx = _scene(rng, n, shape, 2)
    pair, _ = be.subm_rulebook(torch.from_numpy(idx).cuda(), shape, (3, 3, 3), (1, 1, 1), want_rep=False)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.uniform(-0.5, 0.5, cout).astype(np.float32)).cuda()

    def run(finish):
        assert lib.vc_debug_set(b"conv_bn_finish", finish) == 0
        rm = torch.full((cout,), 0.25, device="cuda")
        rv = torch.full((cout,), 0.75, device="cuda")
        nbt = torch.full((1,), 3, dtype=torch.int64, device="cuda")
        before = _finish_launches(lib)
        y, y_raw, mean, var = be.post_act_block_forward(x, w, pair, None, "f32", False, gamma, beta, rm, rv, nbt, 0.01, 1e-3, True)
        torch.cuda.synchronize()
        return (y, y_raw, mean.clone(), var.clone(), rm, rv, nbt), _finish_launches(lib) - before

    names = ("y", "y_raw", "mean", "var", "running_mean", "running_var", "num_batches_tracked")
    try:
        two, took0 = run(0)
        ref, took2 = run(2)
        got, took1 = run(1)
        assert took0 == 0 and took2 == 0
        # > 512 partial rows (one per 16 output rows) is where the finish engages; below, the single small kernel stays
        engaged = (n + 63) // 64 * 4 > 512 and cout >= 8
        assert took1 == (1 if engaged else 0)
        for a, b, what in zip(ref, got, names):
            assert torch.equal(a, b), what
        for a, b, what in zip(two, got, names):   # against the two-launch route: same sums, different (fixed) order
            tol = 2e-6 * max(float(a.double().abs().max()), 1e-30)
            assert float((a.double() - b.double()).abs().max()) <= (0 if what in ("y_raw", "num_batches_tracked") else tol), what
        assert int(got[6]) == 4
        for _ in range(8):   # which block finishes differs from run to run, what it computes does not
            again, _ = run(1)
            for a, b in zip(got, again):
                assert torch.equal(a, b)
        # against torch on the conv output
        m = got[1].double().mean(0)
        v = got[1].double().var(0, unbiased=False)
        assert float((got[2].double() - m).abs().max()) <= 1e-5 * max(1.0, float(m.abs().max()))
        assert float((got[3].double() - v).abs().max()) <= 1e-5 * float(v.abs().max())
    finally:
        lib.vc_debug_set(b"conv_bn_finish", DEFAULT_FINISH)


def test_ticket_slots_are_clean_after_many_launches(hip_backend):
    """300 finishing launches in a row on one stream (more than the 256 ticket slots): every slot is handed out again after its
    launch cleared it; shapes alternate so that group counts differ between the users of a slot."""
    be, lib = hip_backend, hip_backend.lib
    rng = np.random.default_rng(3)
    shape = (21, 200, 176)
    cases = []
    for n, cin, cout in ((9000, 16, 16), (30000, 16, 32), (50000, 32, 32)):
        idx = _scene(rng, n, shape, 1)
        pair, _ = be.subm_rulebook(torch.from_numpy(idx).cuda(), shape, (3, 3, 3), (1, 1, 1), want_rep=False)
        x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
        g = torch.ones(cout, device="cuda")
        b = torch.zeros(cout, device="cuda")
        cases.append((x, w, pair, g, b, cout))

    def run(case, finish):
        x, w, pair, g, b, cout = case
        assert lib.vc_debug_set(b"conv_bn_finish", finish) == 0
        rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
        _, _, mean, var = be.post_act_block_forward(x, w, pair, None, "f32", False, g, b, rm, rv, nbt, 0.01, 1e-3, True)
        return mean.clone(), var.clone()

    try:
        refs = [run(c, 2) for c in cases]
        before = _finish_launches(lib)
        for it in range(300):
            k = it % 3
            m, v = run(cases[k], 1)
            assert torch.equal(m, refs[k][0]) and torch.equal(v, refs[k][1]), it
        assert _finish_launches(lib) - before == 300
    finally:
        lib.vc_debug_set(b"conv_bn_finish", DEFAULT_FINISH)


def test_train_step_gradients_with_and_without_the_in_kernel_finish_are_bit_identical(hip_backend):
    """VirConvL8x train step through the native feature pass (forward statistics AND the backward sums of the 15 units whose
    sums come from a backward-input conv epilogue): every output, every parameter gradient and every BatchNorm buffer is
    bit-identical with conv_bn_finish = 2 (one block, one level after the other) and 1 (tickets), and the finish really ran."""
    import bench
    from virconv_amd import synth
    from virconv_amd.backbone import VirConvL8x
    lib = hip_backend.lib
    dev = torch.device("cuda", 0)
    batch = bench.make_batch([31, 32], dev)
    lw = bench.make_loss_weights(dev)
    torch.manual_seed(2)
    model = VirConvL8x(dict(bench.MODEL_CFG), 8, synth.GRID_SIZE).to(dev).train()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def one(finish):
        assert lib.vc_debug_set(b"conv_bn_finish", finish) == 0
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)   # the layer discard draws its permutations from the torch generator
        before = _finish_launches(lib)
        out = model(dict(batch))
        loss = bench.synthetic_loss(out, lw)
        loss.backward()
        torch.cuda.synchronize()
        feats = {n: t.features.detach().clone() for n, t in out["multi_scale_3d_features"].items()}
        feats["out"] = out["encoded_spconv_tensor"].features.detach().clone()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        bufs = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return float(loss), feats, grads, bufs, _finish_launches(lib) - before

    try:
        two = one(0)
        ref = one(2)
        got = one(1)
        again = one(1)
    finally:
        lib.vc_debug_set(b"conv_bn_finish", DEFAULT_FINISH)
    assert ref[4] == 0 and two[4] == 0 and got[4] >= 20, (ref[4], got[4])   # 20 forward units; + the backward epilogues that qualify
    # against the two-launch route: the BatchNorm sums differ in their last bits (another fixed summation order), which twenty layers
    # of BatchNorm backward and a handful of ReLU decisions at |pre-activation| ~ 1e-8 amplify -- the fp32 noise floor of this
    # network's gradients (6e-3 * max between the fp32 and the float64 oracle); forward features to 1e-5
    for k in two[1]:
        tol = 1e-5 * max(float(two[1][k].abs().max()), 1e-30)
        assert float((two[1][k] - got[1][k]).abs().max()) <= tol, k
    for k in two[2]:
        tol = 1e-2 * max(float(two[2][k].abs().max()), 1e-30)
        assert float((two[2][k] - got[2][k]).abs().max()) <= tol, k
    assert ref[0] == got[0] == again[0]
    for part in (1, 2, 3):
        assert set(ref[part]) == set(got[part])
        for k in ref[part]:
            assert torch.equal(ref[part][k], got[part][k]), (part, k)
            assert torch.equal(got[part][k], again[part][k]), (part, k)


@pytest.mark.parametrize("cin,cout,n", [(16, 32, 40009), (32, 32, 20000), (64, 64, 25000), (32, 64, 61000)])
def test_eight_wave_blocks_of_small_launches_change_no_conv_result(hip_backend, cin, cout, n):
    """Round 6: launches under 62 000 output rows take 8-wave (128-row) blocks for every >= 16-channel shape (vc_debug_set conv_nw8_below;
    VirConv8x's launches are 300-900 four-wave blocks for 256 CUs).  A row's conv result does not depend on the block it is computed in:
    y_raw and the backward-input gradient must be BIT-identical with either block shape; the BatchNorm sums are grouped per wave, so mean /
    var (and with them y) may differ in their last bits, not more."""
    be, lib = hip_backend, hip_backend.lib
    rng = np.random.default_rng(7 * cin + cout)
    shape = (21, 200, 176)
    idx = _scene(rng, n, shape, 2)
    pair, _ = be.subm_rulebook(torch.from_numpy(idx).cuda(), shape, (3, 3, 3), (1, 1, 1), want_rep=False)
    x = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((cout, 3, 3, 3, cin)) / 5).astype(np.float32)).cuda()
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.uniform(-0.5, 0.5, cout).astype(np.float32)).cuda()

    def run(below):
        assert lib.vc_debug_set(b"conv_nw8_below", below) == 0
        rm = torch.full((cout,), 0.25, device="cuda")
        rv = torch.full((cout,), 0.75, device="cuda")
        nbt = torch.full((1,), 3, dtype=torch.int64, device="cuda")
        y, y_raw, mean, var = be.post_act_block_forward(x, w, pair, None, "f32", False, gamma, beta, rm, rv, nbt, 0.01, 1e-3, True)
        dx = be.conv_backward_input(dy, w, pair, n, True, 13, None, order=None, operand="f32")
        torch.cuda.synchronize()
        return y.clone(), y_raw.clone(), mean.clone(), var.clone(), dx.clone()

    try:
        four = run(0)
        eight = run(62000)
    finally:
        assert lib.vc_debug_set(b"conv_nw8_below", 62000) == 0
    assert torch.equal(four[1], eight[1]), "y_raw depends on the block shape"
    assert torch.equal(four[4], eight[4]), "the backward-input gradient depends on the block shape"
    assert torch.allclose(four[2], eight[2], rtol=1e-6, atol=1e-7) and torch.allclose(four[3], eight[3], rtol=1e-6, atol=1e-7)
    assert torch.allclose(four[0], eight[0], rtol=1e-5, atol=1e-6)
