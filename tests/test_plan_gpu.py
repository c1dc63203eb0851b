"""Native geometry plan (vc_plan_begin / vc_plan_wait / vc_plan_finish, csrc/plan.hip, virconv_amd/native_plan.py) against the
operator-by-operator plan of round 3 (backbone._plan_nrconv_chain over vc_hash_build / vc_subm_rulebook / vc_spconv_* /
vc_project_uv / vc_group_plan), which is itself pinned bit-exactly to the oracle (tests/test_ops_gpu.py).

Reference semantics: the indice generation behind every conv of VirConvL8x.forward (pcdet/models/backbones_3d/
spconv_backbone.py:609-699; NRConvBlock :150-229; layer_voxel_discard :134-147; conv_out :561-567).  Integer work: every table,
coordinate list, pixel list, representative list, group plan and keep list must be BIT-IDENTICAL; row orders are scheduling hints
(checked to be permutations that stay inside their 2048-row window and group the rows they are meant to group)."""
import numpy as np
import pytest
import torch

import bench
from helpers import GRID, MODEL_CFG, golden_batch, load_golden
from virconv_amd import native_plan, ops, synth
from virconv_amd import backbone as bb
from virconv_amd.backbone import VirConvL8x

pytestmark = pytest.mark.gpu


def _rb_tensors(rb):
    return {"pair_fwd": rb.pair_fwd, "pair_bwd": rb.pair_bwd, "rep": rb.rep, "grp_plan": rb.grp_plan, "in": rb.in_indices,
            "out": rb.out_indices}


def _assert_same_plan(pn, pp, where=""):
    assert torch.equal(pn["in_indices"], pp["in_indices"])
    assert len(pn["stages"]) == len(pp["stages"])
    for si, (sn, sp) in enumerate(zip(pn["stages"], pp["stages"])):
        tag = f"{where} stage {si}"
        assert list(sn["out_shape"]) == list(sp["out_shape"]), tag
        assert torch.equal(sn["out_indices"], sp["out_indices"]), tag
        assert torch.equal(sn["uv"], sp["uv"]), tag
        assert (sn["keep"] is None) == (sp["keep"] is None), tag
        if sn["keep"] is not None:
            assert torch.equal(sn["keep"], sp["keep"]) and torch.equal(sn["kept_indices"], sp["kept_indices"]), tag
        for group in ("rb3d", "rb2d"):
            assert sn[group].keys() == sp[group].keys(), tag
            for key in sn[group]:
                _assert_same_rulebook(sn[group][key], sp[group][key], f"{tag} {key}")
    assert pn["conv_out"].keys() == pp["conv_out"].keys()
    for key in pn["conv_out"]:
        _assert_same_rulebook(pn["conv_out"][key], pp["conv_out"][key], f"{where} conv_out")


def _assert_same_rulebook(a, b, tag):
    assert (a.kind, a.n_in, a.n_out, tuple(a.in_shape), tuple(a.out_shape), tuple(a.ksize), tuple(a.stride), tuple(a.padding),
            tuple(a.dilation)) == (b.kind, b.n_in, b.n_out, tuple(b.in_shape), tuple(b.out_shape), tuple(b.ksize), tuple(b.stride),
                                   tuple(b.padding), tuple(b.dilation)), tag
    ta, tb = _rb_tensors(a), _rb_tensors(b)
    for name in ta:
        assert (ta[name] is None) == (tb[name] is None), f"{tag}: {name} presence"
        if ta[name] is not None:
            assert ta[name].shape == tb[name].shape and torch.equal(ta[name], tb[name]), f"{tag}: {name} differs"
    for name in ("order_fwd", "order_bwd"):
        oa, ob = getattr(a, name), getattr(b, name)
        assert (oa is None) == (ob is None), f"{tag}: {name} presence"
        if oa is not None:
            n = oa.shape[0]
            assert n == ob.shape[0]
            o = oa.long().cpu().numpy()
            assert np.array_equal(np.sort(o), np.arange(n)), f"{tag}: {name} is not a permutation"
            assert np.array_equal(o // 2048, np.arange(n) // 2048), f"{tag}: {name} leaves its 2048-row window"


def _plans(model, batch, monkeypatch, seed=5):
    """(native plan, operator-by-operator plan) of the same model / batch / torch seed."""
    bd = dict(batch)
    calib = bd["calib"] if torch.is_tensor(bd["calib"]) else ops.calib_tensor(bd["calib"], bd["voxel_coords"].device)
    out = []
    for native in (True, False):
        monkeypatch.setattr(native_plan, "NATIVE_PLAN", native)
        torch.manual_seed(seed)
        p = model.build_plan(bd["voxel_coords"], bd["batch_size"], calib, bd.get("aug_param"), bd)
        bb.join_plan(p)   # the backward-only structures are enqueued at the end of a forward pass
        torch.cuda.synchronize()
        assert ("_arenas" in p) == native
        out.append(p)
    return out


@pytest.mark.parametrize("mode", ["train_random_keep", "train_noop_discard", "eval"])
def test_native_plan_equals_the_operator_by_operator_plan_small(hip_backend, monkeypatch, mode):
    g = load_golden()
    batch = golden_batch(g, "cuda")
    cfg = dict(MODEL_CFG, LAYER_DISCARD_MODE="spconv1_inplace" if mode == "train_random_keep" else "spconv2_noop")
    model = VirConvL8x(cfg, 8, GRID).cuda()
    model.train(mode != "eval")
    with torch.set_grad_enabled(mode != "eval"):
        pn, pp = _plans(model, batch, monkeypatch)
    _assert_same_plan(pn, pp, mode)


@pytest.mark.parametrize("frames,inject", [([0, 1], False), ([0, 1, 2, 3], True), ([3], False)])
def test_native_plan_equals_the_operator_by_operator_plan_full_size(hip_backend, monkeypatch, frames, inject):
    """KITTI-sized synthetic frames (the bench batch for [0, 1, 2, 3]), layer discard on; `inject`: the keeps are injected
    permutation prefixes instead of drawn ones (the row counts they need come from a first plan)."""
    dev = torch.device("cuda", 0)
    batch = bench.make_batch(frames, dev, training=True)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    if inject:   # the keeps of a first (drawn) plan, injected into both routes as the tests / benchmarks do
        monkeypatch.setattr(native_plan, "NATIVE_PLAN", False)
        torch.manual_seed(3)
        p0 = model.build_plan(batch["voxel_coords"], len(frames), batch["calib"], batch["aug_param"], batch)
        batch["layer_discard_keep"] = {f"x_conv{bi + 1}": p0["stages"][bi]["keep"].flip(0).cpu() for bi in range(3)}
    pn, pp = _plans(model, batch, monkeypatch)
    _assert_same_plan(pn, pp, f"frames {frames}")
    # the strided backward row orders of the native plan group the rows by stride-residue class inside every window: rows of one
    # class have identical active-offset sets away from the grid border, which is what the order is for
    for si in (1, 2, 3):
        rb = [r for r in pn["stages"][si]["rb3d"].values() if r.kind == "sparse"][0]
        o = rb.order_bwd.long()
        c = rb.in_indices.long()[o]
        cls = (((c[:, 1] + rb.padding[0]) % rb.stride[0]) * rb.stride[1] + ((c[:, 2] + rb.padding[1]) % rb.stride[1])) * rb.stride[2] + \
              ((c[:, 3] + rb.padding[2]) % rb.stride[2])
        win = torch.arange(o.shape[0], device=o.device) // 2048
        key = win * 64 + cls
        assert bool((key[1:] >= key[:-1]).all()), "rows of a window are not grouped by residue class"
        same = key[1:] == key[:-1]
        assert bool((o[1:][same] > o[:-1][same]).all()), "the order is not stable inside a class"


@pytest.mark.parametrize("key", ["plan_subm_bitmap", "plan_image_2d", "plan_parity_order", "sp_mark_variant", "plan_group_multi",
                                 "group_plan_multi_onesweep"])
def test_plan_kernel_switches_do_not_change_any_table(hip_backend, monkeypatch, key):
    """Every chain-aware index kernel has the generic operator as its A/B alternative (vc_debug_set): same tables either way."""
    dev = torch.device("cuda", 0)
    batch = bench.make_batch([0, 1], dev, training=True)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    monkeypatch.setattr(native_plan, "NATIVE_PLAN", True)
    lib = hip_backend.lib
    plans = []
    try:
        for v in ((2, 1) if key == "sp_mark_variant" else (1, 0)):
            assert lib.vc_debug_set(key.encode(), v) == 0
            torch.manual_seed(11)
            plans.append(model.build_plan(batch["voxel_coords"], 2, batch["calib"], batch["aug_param"], batch))
            bb.join_plan(plans[-1])
            torch.cuda.synchronize()
    finally:
        assert lib.vc_debug_set(key.encode(), 2 if key == "sp_mark_variant" else 1) == 0
    _assert_same_plan(plans[0], plans[1], key)


@pytest.mark.parametrize("n,groups", [(1, 1), (63, 5), (4096, 4096), (4097, 300), (70001, 9000), (310351, 60000), (200000, 3),
                                      (2100000, 500000)])
def test_group_plan_is_the_stable_sort_by_representative(hip_backend, n, groups):
    """vc_group_plan against a stable argsort; in a -DVC_EXPERIMENTS build also the hand-written LDS radix sort of
    csrc/experiments/group_plan_radix.inc (vc_debug_set plan_radix_sort = 1; the last case is beyond its range)."""
    rng = np.random.default_rng(n + groups)
    dev = torch.device("cuda", 0)
    rep_groups = rng.integers(0, groups, n)
    last = np.full(groups, -1, np.int64)
    last[rep_groups] = np.arange(n)            # later rows overwrite: the highest row of each group = its representative
    rep_np = last[rep_groups].astype(np.int32)
    if n > 100:
        rep_np[::17] = -1                      # "own representative" marker of some tables: key = the row itself
    rep = torch.from_numpy(rep_np).to(dev)
    keys = np.where(rep_np < 0, np.arange(n), rep_np)
    order = np.argsort(keys, kind="stable")
    lib = hip_backend.lib
    got = hip_backend.group_plan(rep)
    assert np.array_equal(got[0].cpu().numpy(), order) and np.array_equal(got[1].cpu().numpy(), keys[order])
    for _ in range(2):
        assert torch.equal(hip_backend.group_plan(rep), got)
    if lib.vc_debug_set(b"plan_radix_sort", 1) == 0:      # experiment builds only
        try:
            assert torch.equal(hip_backend.group_plan(rep), got)
        finally:
            assert lib.vc_debug_set(b"plan_radix_sort", 0) == 0


def test_train_step_with_the_native_plan_matches_the_operator_by_operator_plan(hip_backend, monkeypatch):
    """Whole bench train step: with the residue-class row order switched off the two plans hold identical structures, so loss,
    outputs and every gradient are BIT-identical; with it on (the default) only the tile composition of the strided
    backward-input convs changes -- the BatchNorm-backward sums formed in their epilogue re-associate: <= 1e-5 of max."""
    dev = torch.device("cuda", 0)
    batch = bench.make_batch([0, 1], dev, training=True)
    lw = bench.make_loss_weights(dev)
    lib = hip_backend.lib

    def run(native, parity):
        monkeypatch.setattr(native_plan, "NATIVE_PLAN", native)
        assert lib.vc_debug_set(b"plan_parity_order", parity) == 0
        torch.manual_seed(0)
        model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
        torch.manual_seed(77)
        opt.zero_grad(set_to_none=True)
        bd = dict(batch)
        bd["voxel_features"] = batch["voxel_features"].clone()
        out = model(bd)
        loss = bench.synthetic_loss(out, lw)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {k: p.grad.clone() for k, p in model.named_parameters()}

    try:
        l_py, g_py = run(False, 1)
        l_n0, g_n0 = run(True, 0)
        l_n1, g_n1 = run(True, 1)
    finally:
        assert lib.vc_debug_set(b"plan_parity_order", 1) == 0
    assert l_py == l_n0 == l_n1
    for k in g_py:
        assert torch.equal(g_py[k], g_n0[k]), k
        scale = float(g_py[k].abs().max()) + 1e-30
        assert float((g_py[k] - g_n1[k]).abs().max()) <= 1e-5 * scale, k


def test_training_with_the_plan_built_a_step_ahead_is_bit_identical(hip_backend):
    """VirConvL8x.plan_ahead_begin (bench.train_step(next_batch=...)): the first half of the plan of step t + 1 enqueued before step t's
    forward, the tables built when step t + 1 asks for them.  With the layer discards injected (the random draws would otherwise be consumed in a different
    order) three optimiser steps give the same losses and the same parameters, bit for bit; a batch the early plan was NOT begun for
    falls back to the in-place plan."""
    dev = torch.device("cuda", 0)
    batch = bench.make_batch([0, 1], dev, training=True)
    other = bench.make_batch([2, 3], dev, training=True)
    lw = bench.make_loss_weights(dev)
    torch.manual_seed(3)
    probe = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    for b in (batch, other):
        p0 = probe.build_plan(b["voxel_coords"], 2, b["calib"], b["aug_param"], b)
        bb.join_plan(p0)
        b["layer_discard_keep"] = {f"x_conv{bi + 1}": p0["stages"][bi]["keep"].clone() for bi in range(3)}
    torch.cuda.synchronize()

    def run(ahead):
        torch.manual_seed(0)
        model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
        seq = [batch, batch, other, batch]
        losses, took = [], 0
        for t, b in enumerate(seq):
            nxt = seq[t + 1] if (ahead and t + 1 < len(seq)) else None
            if ahead and t == 1:
                nxt = batch          # a WRONG guess: step 2 runs `other`, the early plan must be dropped
            had = len(model._ahead) > 0
            losses.append(float(bench.train_step(model, opt, b, lw, next_batch=nxt)))
            took += int(had)
        torch.cuda.synchronize()
        return losses, {k: p.detach().clone() for k, p in model.named_parameters()}, took

    l0, p0_, _ = run(False)
    l1, p1_, took = run(True)
    assert took == 3                      # steps 1, 2 (dropped: wrong batch), 3 entered forward with an early plan pending
    assert l0 == l1
    for k in p0_:
        assert torch.equal(p0_[k], p1_[k]), k


def test_a_plan_begun_and_never_finished_does_not_block_the_count_ring(hip_backend):
    """Plans begun ahead for batches that never come may stay referenced (the model <-> plan cycle waits for the cyclic collector).  The
    pinned count ring then wraps onto their slot: the newer plan takes it, the abandoned one is marked stale -- finishing it raises,
    a forward that finds it plans in place instead."""
    dev = torch.device("cuda", 0)
    batch = bench.make_batch([0], dev, training=True)
    model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
    kept = []
    for _ in range(native_plan._RING + 2):
        assert model.plan_ahead_begin(batch)
        kept.append(model._ahead[-1])          # keeps the begun plan alive although the model drops all but the last two
    cps = [a["entries"][""][3] for a in kept]
    assert cps[0].stale and cps[1].stale and not cps[-1].stale
    with pytest.raises(RuntimeError, match="recycled"):
        cps[0].finish()
    model._ahead[:] = [kept[1]]                # the forward finds only a stale early plan for this batch
    out = model(dict(batch, voxel_features=batch["voxel_features"].clone()))
    assert out["encoded_spconv_tensor"].features.shape[0] > 0
    assert not cps[1].counts_read              # ... and did not touch it
    torch.cuda.synchronize()


def test_native_plan_of_virconv8x_equals_the_operator_by_operator_plan(hip_backend, monkeypatch):
    """VirConv8x, training (spconv_backbone.py:339-535): the LiDAR stream (conv_input / conv1..4 / conv_out: one SubM table per
    stage, no image-space branch) and the virtual-point stream (input discard :488-489 + four NRConvBlocks + layer discards) are
    one native chain plan each; every structure equals the operator-by-operator plan's."""
    from virconv_amd.backbone import VirConv8x
    dev = torch.device("cuda", 0)
    batch = bench.make_batch_8x([0, 1], dev)
    model = VirConv8x(bench.MODEL_CFG_8X, 8, synth.GRID_SIZE).cuda().train()
    assert model._discard_active() and model.mm
    plans = []
    for native in (True, False):
        monkeypatch.setattr(native_plan, "NATIVE_PLAN", native)
        torch.manual_seed(9)
        p = model.build_plan(batch, [""], 2, batch["calib"])
        bb.join_plan(p)
        torch.cuda.synchronize()
        assert ("_arenas" in p) == native
        plans.append(p)
    pn, pp = plans
    (idx_n, rbs_n), (idx_p, rbs_p) = pn["lidar"][""], pp["lidar"][""]
    assert torch.equal(idx_n, idx_p) and rbs_n.keys() == rbs_p.keys()
    for key in rbs_n:
        _assert_same_rulebook(rbs_n[key], rbs_p[key], f"lidar {key}")
    mn, mp_ = pn["mm"][""], pp["mm"][""]
    assert torch.equal(mn["keep0"], mp_["keep0"]) and torch.equal(mn["in_indices"], mp_["in_indices"])
    _assert_same_plan({"in_indices": mn["in_indices"], "stages": mn["stages"], "conv_out": {}},
                      {"in_indices": mp_["in_indices"], "stages": mp_["stages"], "conv_out": {}}, "8x mm")
