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

Every projection runs as project_uv_kernel<true>: each thread compares the parameter block it loads with a golden copy saved behind
project_prepare_kernel and logs what it saw (vc_plan_desc.debug_buf).  Usage:  python tools/det_check.py [--reps 12] [--bs 2] [variants ...]
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
ap.add_argument("variants", nargs="*", default=["base", "reprepare", "pad", "dwmain", "exactmfma", "private", "guard"])
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
DBG_INTS = 64 + bs * 32 + 32 * 4096


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
    recs = dbg[64 + bs * 32: 64 + bs * 32 + 32 * min(n, 4096)].view(-1, 32).cpu()
    gold = dbg[64: 64 + bs * 32].view(bs, 32).cpu()
    f = lambda v: struct.unpack("f", struct.pack("i", int(v)))[0]   # noqa: E731
    print(f"      {label}: {n} projection threads read a parameter block that differs from the golden copy; first / last records:")
    for r in list(recs[:4]) + list(recs[-2:]):
        b = int(r[1])
        print(f"        row {int(r[0])} sample {b} block {int(r[2])} first bad word {int(r[3])} xcd {int(r[5]) & 15} stride {int(r[29])} t {(int(r[7]) << 32) | (int(r[6]) & 0xffffffff)}"
              f" | saw P[24:32] = {[round(f(v), 5) for v in r[8:16]]} | golden {[round(f(v), 5) for v in gold[b, 24:32]]}"
              f" | {int(r[28])} ticks later: {[round(f(v), 5) for v in r[20:28]]} | P[0], P[11], P[12], P[23] = {[round(f(v), 4) for v in r[16:20]]}"
              f" golden {[round(f(gold[b, j]), 4) for j in (0, 11, 12, 23)]}")
    rows = recs[:, 0].long()
    print(f"        rows {int(rows.min())} .. {int(rows.max())}, samples {sorted(set(recs[:, 1].tolist()))}, blocks {int(recs[:, 2].min())} .. {int(recs[:, 2].max())},"
          f" strides {sorted(set(recs[:, 29].tolist()))}, bad words {sorted(set(recs[:, 3].tolist()))}")


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
        gold = dbg[64: 64 + bs * 32]
        print(f"  [{name}] rep {rep}: {len(ta)} structures, differing: {bad}; parameter block now == golden copy: {bool(torch.equal(pa, gold))};"
              f" threads that saw something else: beside backward {int(dbg[0])}, idle {int(dbg_f[0])}")
        if name == "pad":   # the 4 KB in front of the moved block: poison check (vc_plan_begin never writes it)
            padw = cp.arena_a[64: 64 + 1024]
            print(f"      words of the padding changed since allocation cannot be told (uninitialised); nonzero now: {int((padw != 0).sum())}")
        for k, _ in bad:
            if k.endswith(".uv"):
                rows = (ta[k] != tf[k]).any(1).nonzero().squeeze(1)
                print(f"      {k}: {rows.numel()} rows differ, span {int(rows.min())} .. {int(rows.max())}, samples {sorted(set(ta[k][rows, 0].tolist()))};"
                      f" beside-backward {ta[k][rows[:3]].tolist()} idle {tf[k][rows[:3]].tolist()}")
        show_records(dbg, "beside backward")
        show_records(dbg_f, "idle")
        bad_reps += bool(bad)
        del cp, cf, model, opt
    print(f"[{name}] {bad_reps} of {args.reps} reps differ")
    native_plan.ARENA_ALLOC = None


for v in args.variants:
    run_variant(v)
