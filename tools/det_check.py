"""LOG.md A.15 / A.17 diagnostics in ONE script (replaces det_check{,2,3,4}.py).

Re-creates the form that failed in round 4 -- the TABLES of the next step's geometry plan (vc_plan_finish: pixel projection, pair
tables) enqueued on the plan stream right after `loss.backward()` returns, i.e. beside the GPU's backward pass -- and compares every
structure of that plan with one built on an idle GPU.  Each variant changes one thing about WHO could be writing WHAT:

    base        the failing form (no guard)
    reprepare   vc_debug_set plan_reprepare = 1: the projection parameter block is written again right in front of every projection
    pad         vc_debug_set plan_params_pad = 4096: the parameter block moved 4 KB into its arena
    dwmain      weight gradients on the main stream (no third stream)
    exactmfma   vc_debug_set f32_split = 0, bw_split = 0 (the form that never failed)
    private     plan arenas from a private, never-freed pool (rules torch's caching allocator out)
    guard       the shipped guard (the plan stream waits for the backward pass in front of its first table kernel)
    select      vc_debug_set plan_uv_mode = 1: project_uv_kernel<1>, no divergent branch on the loaded flag word (per-lane select)
    argflag     vc_debug_set plan_uv_mode = 2: project_uv_kernel<2>, "has augmentation" as a kernel argument (the flag word is not read)

Every projection runs as project_uv_kernel<MODE, true>: a thread whose flag word P[28] reads "no augmentation" although the plan has
one logs the bits it saw, its wave's ballot, and what the same word reads again (vc_plan_desc.debug_buf).  Usage:  python tools/det_check.py [--reps 12] [--bs 2] [variants ...]
"""
import argparse
import os
import struct
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import backbone as bb, native_plan, ops, synth  # noqa: E402
from virconv_amd.backbone import NRConvBlock, VirConvL8x  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--bs", type=int, default=2)
ap.add_argument("--stress", type=int, default=0,
                help="N > 0: instead of the variants, run the benchmark's own unsynchronised loop for N steps WITHOUT the guard (plans kept), "
                     "compare every step's pixel coordinates with a rebuild on an idle GPU and match the wrong rows against an fp32 emulation "
                     "of the projection with single steps of its arithmetic knocked out")
ap.add_argument("variants", nargs="*", default=["base", "select", "argflag", "reprepare", "pad", "dwmain", "exactmfma", "private", "guard"])
args = ap.parse_args()

dev = torch.device("cuda", 0)
be = ops.get_backend()
lw = bench.make_loss_weights(dev)
batch = bench.make_batch(list(range(args.bs)), dev, training=True)
bs = args.bs
torch.manual_seed(3)
probe = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
bb.PLAN_GUARD = 0
p0 = probe.build_plan(batch["voxel_coords"], bs, batch["calib"], batch["aug_param"], batch)
bb.join_plan(p0)
batch["layer_discard_keep"] = {f"x_conv{bi + 1}": p0["stages"][bi]["keep"].clone() for bi in range(3)}
torch.cuda.synchronize()
SIDE_ROWS = 400_000
DBG_INTS = 64 + 32 * 4096 + 4 * SIDE_ROWS * 8


def dset(key, val):
    assert be.lib.vc_debug_set(key.encode(), int(val)) == 0, key


def begin(model, guard=None):
    """vc_plan_begin of the batch on the plan stream, tables of block 0 included (the round-4 form) -> (ChainPlan, idx, dbg)."""
    blocks = [(model.vir_conv1, 1), (model.vir_conv2, 2), (model.vir_conv3, 4), (model.vir_conv4, 8)]
    co = model.conv_out[0]
    side = bb._plan_stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dbg = torch.zeros((DBG_INTS,), dtype=torch.int32, device=dev)
        dbg[3] = SIDE_ROWS
        idx = batch["voxel_coords"].int()
        tags = [f"x_conv{bi + 1}" if bi < 3 else None for bi in range(4)]
        cp = native_plan.ChainPlan(model, native_plan.nrconv_kind(blocks, co, None), native_plan.nrconv_blocks(blocks), co, idx, bs,
                                   batch["calib"], batch["aug_param"], tags, model.layer_discard_rate, batch, NRConvBlock.IMAGE_SHAPE,
                                   None, [], guard, False, dbg)
    return cp, idx, dbg, blocks


def tensors(cp, blocks):
    res, rb_out, _, _, _ = cp.result
    out = {}
    for si, r in enumerate(res):
        for k in ("out_indices", "uv", "keep", "kept_indices"):
            if r.get(k) is not None:
                out[f"s{si}.{k}"] = r[k]
        for key in ("down", "subm3d", "subm2d"):
            rb = r[key]
            if rb is None:
                continue
            for name in ("pair_fwd", "pair_bwd", "rep", "order_fwd"):
                t = getattr(rb, name, None)
                if t is not None:
                    out[f"s{si}.{key}.{name}"] = t
    for name in ("pair_fwd", "pair_bwd", "out_indices"):
        out[f"out.{name}"] = getattr(rb_out, name)
    return out


def show_records(dbg, label):
    n = int(dbg[0])
    if n == 0:
        return
    recs = dbg[64: 64 + 32 * min(n, 4096)].view(-1, 32).cpu()
    f = lambda v: struct.unpack("f", struct.pack("i", int(v)))[0]   # noqa: E731
    print(f"      {label}: {n} projection threads took the flag word P[28] for 'no augmentation'; first / last records:")
    for r in list(recs[:6]) + list(recs[-2:]):
        bal = ((int(r[7]) & 0xffffffff) << 32) | (int(r[6]) & 0xffffffff)
        print(f"        row {int(r[0])} sample {int(r[1])} block {int(r[2])} lane {int(r[3])} xcd {int(r[5]) & 15} stride {int(r[15])} mode {int(r[16])}"
              f" | flag bits as used {int(r[4]) & 0xffffffff:#010x} | wave ballot(has) {bal:#018x} | re-read at once {int(r[8]) & 0xffffffff:#010x},"
              f" later {int(r[9]) & 0xffffffff:#010x} | P[24:29] re-read {[round(f(v), 5) for v in r[10:15]]}")
    print(f"        rows {int(recs[:, 0].min())} .. {int(recs[:, 0].max())}, lanes {sorted(set(recs[:, 3].tolist()))}, flag bits seen "
          f"{sorted({hex(int(v) & 0xffffffff) for v in recs[:, 4].tolist()})}, strides {sorted(set(recs[:, 15].tolist()))}")


class PrivatePool:
    """Plan arenas from buffers that are never returned to torch's caching allocator (four sets, round robin)."""

    def __init__(self):
        self.bufs = [torch.empty((96 << 20,), dtype=torch.int32, device=dev) for _ in range(8)]
        self.i = 0

    def __call__(self, nwords):
        b = self.bufs[self.i % len(self.bufs)]
        self.i += 1
        assert nwords <= b.numel()
        return b[:nwords]


def run_variant(name):
    dset("plan_uv_mode", {"select": 1, "argflag": 2, "rcp": 3}.get(name, 0))
    dset("plan_reprepare", 1 if name == "reprepare" else 0)
    dset("plan_params_pad", 4096 if name == "pad" else 0)
    exact = name == "exactmfma"
    dset("f32_split", 0 if exact else 1)
    dset("bw_split", 0 if exact else 1)
    os.environ["VIRCONV_PASS_OVERLAP_DW"] = "0" if name == "dwmain" else "1"
    native_plan.ARENA_ALLOC = PrivatePool() if name == "private" else None
    bad_reps = 0
    for rep in range(args.reps):
        torch.manual_seed(0)
        model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).cuda().train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
        bench.train_step(model, opt, batch, lw)          # a step in front (allocator, clocks)
        # --- the step under test: plan of the NEXT step begun before the forward, finished right after backward() returns
        opt.zero_grad(set_to_none=True)
        cp, idx, dbg, blocks = begin(model)
        bd = dict(batch)
        bd["voxel_features"] = batch["voxel_features"].clone()
        loss = bench.synthetic_loss(model(bd), lw)
        loss.backward()
        guard = None
        if name == "guard":
            guard = torch.cuda.Event()
            guard.record(torch.cuda.current_stream())
        with torch.cuda.stream(bb._plan_stream(dev)):
            cp.finish(guard)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        torch.cuda.synchronize()
        # --- the same plan on an idle GPU
        cf, _, dbg_f, _ = begin(model)
        with torch.cuda.stream(bb._plan_stream(dev)):
            cf.finish()
        torch.cuda.synchronize()
        ta, tf = tensors(cp, blocks), tensors(cf, blocks)
        assert ta.keys() == tf.keys()
        bad = [(k, int((ta[k] != tf[k]).sum())) for k in ta if ta[k].shape != tf[k].shape or not torch.equal(ta[k], tf[k])]
        pa = cp.arena_a[(256 + (4096 if name == "pad" else 0)) // 4:][: bs * 32]
        pf = cf.arena_a[(256 + (4096 if name == "pad" else 0)) // 4:][: bs * 32]
        if bad or int(dbg[0]) or int(dbg_f[0]):
            print(f"  [{name}] rep {rep}: {len(ta)} structures, differing: {bad}; parameter blocks of the two plans equal now: {bool(torch.equal(pa, pf))};"
                  f" threads that took the flag word for 0: beside backward {int(dbg[0])}, idle {int(dbg_f[0])}")
        for k, _ in bad:
            if k.endswith(".uv"):
                # which part of the inverse augmentation did the wrong rows lose?  Stand-alone projections of the same coordinates
                # (idle GPU) with parts of the augmentation parameters [rot, flip, scale] neutralised
                rows_ = (ta[k] != tf[k]).any(1).nonzero().squeeze(1)
                si_ = int(k[1])
                co_ = ta[f"s{si_}.out_indices"] if f"s{si_}.out_indices" in ta else idx
                aug = batch["aug_param"].float()
                hyp = {"no augmentation at all": None, "rotation lost": aug * torch.tensor([0., 1., 1.], device=dev),
                       "flip lost": aug * torch.tensor([1., 0., 1.], device=dev),
                       "scale lost": aug * torch.tensor([1., 1., 0.], device=dev) + torch.tensor([0., 0., 1.], device=dev),
                       "rotation + flip lost": aug * torch.tensor([0., 0., 1.], device=dev)}
                match = {}
                for name_h, tp in hyp.items():
                    alt = ops.project_uv(co_, batch["calib"], tp, bs, 2 ** si_)
                    match[name_h] = int((alt[rows_] == ta[k][rows_]).all(1).sum())
                torch.cuda.synchronize()
                print(f"      {k}: of {rows_.numel()} wrong rows, equal to the projection with ... {match}")
                # the intermediates both kernels wrote for these rows: [X, Y, Z after the inverse augmentation, rect xyz, hom x y]
                sa_ = dbg[64 + 32 * 4096:].view(torch.float32).view(4, SIDE_ROWS, 8)[si_]
                sf_ = dbg_f[64 + 32 * 4096:].view(torch.float32).view(4, SIDE_ROWS, 8)[si_]
                for r_ in rows_[:3].tolist() + rows_[-1:].tolist():
                    print(f"        row {r_} lane {r_ % 64} coords {co_[r_].tolist()}: beside {[round(float(x), 4) for x in sa_[r_]]}")
                    print(f"        {'':>{len(str(r_)) + 30}} idle   {[round(float(x), 4) for x in sf_[r_]]}")
                cols = (sa_[rows_] != sf_[rows_]).any(0).tolist()
                print(f"        intermediates that differ in these rows (X Y Z | rect | hom): {cols}; rows whose X, Y, Z are bit-equal: {int((sa_[rows_, :3] == sf_[rows_, :3]).all(1).sum())}")
                rows = (ta[k] != tf[k]).any(1).nonzero().squeeze(1)
                print(f"      {k}: {rows.numel()} rows differ, span {int(rows.min())} .. {int(rows.max())}, lanes {sorted(set((rows % 64).tolist()))}, samples {sorted(set(ta[k][rows, 0].tolist()))};"
                      f" beside-backward {ta[k][rows[:3]].tolist()} idle {tf[k][rows[:3]].tolist()}")
        show_records(dbg, "beside backward")
        show_records(dbg_f, "idle")
        bad_reps += bool(bad)
        del cp, cf, model, opt
    print(f"[{name}] {bad_reps} of {args.reps} reps differ")
    native_plan.ARENA_ALLOC = None


def emulate(coords, calib, aug, stride, knock):
    """fp32 restatement of project_uv_kernel on the CPU, op for op (every product and sum rounded separately); `knock` names one
    step of the inverse augmentation that is skipped or altered.  -> (n, 2) int64 [u, v]"""
    f32 = torch.float32
    c = coords.cpu()
    b = c[:, 0].long()
    vs = torch.tensor(0.05 * stride, dtype=torch.float64).to(f32)
    mn = [torch.tensor(v + 0.05 * stride / 2, dtype=torch.float64).to(f32) for v in (0.0, -40.0, -3.0)]
    X = c[:, 3].to(f32) * vs + mn[0]
    Y = c[:, 2].to(f32) * vs + mn[1]
    Z = c[:, 1].to(f32) * vs + mn[2]
    cal = calib.cpu().to(f32)
    v2c, r0, p2 = cal[:, :12].view(-1, 3, 4), cal[:, 12:21].view(-1, 3, 3), cal[:, 21:33].view(-1, 3, 4)
    a = aug.cpu().to(f32)
    ang = -(a[:, 0].double())
    ca, sa = torch.cos(ang).to(f32)[b], torch.sin(ang).to(f32)[b]
    flip, sc = (a[:, 1] != 0)[b], a[:, 2][b]
    Xs, Ys, Zs = (X / sc, Y / sc, Z / sc) if knock != "scale" else (X, Y, Z)
    if knock == "scale_z_only_kept":
        Xs, Ys = X, Y
    if knock != "flip":
        Ys = torch.where(flip, -Ys, Ys)
    nsa = -sa
    X2 = Xs * ca + Ys * nsa
    Y2 = Xs * sa + Ys * ca
    if knock == "Y2=Y":
        Y2 = Ys
    elif knock == "X2=X":
        X2 = Xs
    elif knock == "rotation":
        X2, Y2 = Xs, Ys
    elif knock == "Y2=Xsa-Yca":
        Y2 = Xs * sa - Ys * ca
    elif knock == "X2=Xca+Ysa":
        X2 = Xs * ca + Ys * sa
    elif knock == "other sample":
        bo = (b + 1) % a.shape[0]
        ang2 = -(a[:, 0].double())
        ca2, sa2 = torch.cos(ang2).to(f32)[bo], torch.sin(ang2).to(f32)[bo]
        X2 = Xs * ca2 + Ys * (-sa2)
        Y2 = Xs * sa2 + Ys * ca2
    elif knock == "all":
        X2, Y2, Zs = X, Y, Z
    # M1 = V2C^T @ R0^T (4 x 3), per sample, each element three products summed in order
    M1 = torch.zeros((cal.shape[0], 4, 3), dtype=f32)
    for r in range(4):
        for cc in range(3):
            M1[:, r, cc] = (v2c[:, 0, r] * r0[:, cc, 0] + v2c[:, 1, r] * r0[:, cc, 1]) + v2c[:, 2, r] * r0[:, cc, 2]
    P = M1[b]
    rect = [((X2 * P[:, 0, k] + Y2 * P[:, 1, k]) + Zs * P[:, 2, k]) + P[:, 3, k] for k in range(3)]
    p2t = p2[b]   # (n, 3, 4): P2T[r][c] = p2[c][r]
    hom = [((rect[0] * p2t[:, k, 0] + rect[1] * p2t[:, k, 1]) + rect[2] * p2t[:, k, 2]) + p2t[:, k, 3] for k in range(2)]
    u = torch.nan_to_num(hom[0] / rect[2], nan=0.0, posinf=2.0e9, neginf=-2.0e9).trunc().clamp(-2147483648, 2147483647).long()
    v = torch.nan_to_num(hom[1] / rect[2], nan=0.0, posinf=2.0e9, neginf=-2.0e9).trunc().clamp(-2147483648, 2147483647).long()
    u = u.clamp(0, 1399) // stride
    v = v.clamp(0, 599) // stride
    return torch.stack([u, v], 1)


KNOCKS = ["none", "Y2=Y", "X2=X", "rotation", "Y2=Xsa-Yca", "X2=Xca+Ysa", "flip", "scale", "scale_z_only_kept", "other sample", "all"]


def stress(n_steps):
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    tps = importlib.import_module("test_plan_stress_gpu")
    _, _, b4, model, opt, lw4 = tps._setup()
    seen = []
    for t in range(3):
        bench.train_step(model, opt, b4, lw4)
    b4["plan_observer"] = lambda rid, plan: seen.append(plan)
    for t in range(n_steps):
        torch.manual_seed(5000 + t)
        bench.train_step(model, opt, b4, lw4)
    torch.cuda.synchronize()
    del b4["plan_observer"]
    tally = {k: 0 for k in KNOCKS}
    total, emu_ok, emu_n = 0, 0, 0
    for t in range(n_steps):
        ref = tps._rebuild(model, b4, t)
        for si, (st_g, st_r) in enumerate(zip(seen[t]["stages"], ref["stages"])):
            g, r = st_g["uv"], st_r["uv"]
            rows = (g != r).any(1).nonzero().squeeze(1)
            if rows.numel() == 0:
                continue
            co = st_r["out_indices"][rows]
            got = g[rows][:, 1:].cpu().long()
            if emu_n < 4:   # the emulation itself against the kernel on rows that are right
                some = torch.arange(0, r.shape[0], max(1, r.shape[0] // 2000), device=r.device)
                e = emulate(st_r["out_indices"][some], b4["calib"], b4["aug_param"], 2 ** si, "none")
                emu_ok += int((e == r[some][:, 1:].cpu().long()).all(1).sum())
                emu_n += 1
                print(f"  emulation vs kernel on {some.numel()} right rows of step {t} stage {si}: {int((e == r[some][:, 1:].cpu().long()).all(1).sum())} equal")
            total += rows.numel()
            for k in KNOCKS:
                e = emulate(co, b4["calib"], b4["aug_param"], 2 ** si, k)
                tally[k] += int((e == got).all(1).sum())
        seen[t] = None
    print(f"[stress {n_steps} steps, no guard] {total} wrong pixel rows; rows reproduced exactly by the emulation with ... {tally}")


if args.stress:
    stress(args.stress)
    sys.exit(0)
for v in args.variants:
    run_variant(v)
