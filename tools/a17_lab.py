"""LOG.md A.17 lab (round 6): WHERE must the pixel projection be, and WHAT must run beside it, for lanes 48-63 to go wrong?

Two experiments, both with the fence off (the round-4 loop), both printing one JSON line per run so that a shell loop can tabulate them:

  micro   `vc_project_uv` over the benchmark batch's coordinates, launched again and again on a "victim" stream while an "aggressor" runs on
          the main stream; every launch's pixels are compared on the device with the idle-GPU result.  Aggressors: none | step (the whole
          train step, fence on for ITS plan) | fwd / dw (one bf16-split gather-GEMM / weight-gradient layer in a loop) | fwd_exact / dw_exact
          (the same with exact-fp32 MFMA products) | valu (a torch fp32 elementwise kernel) | mm_bf16 / mm_f32 (rocBLAS / hipBLASLt GEMMs:
          somebody else's MFMA kernels).  Victim variants: --mode (project_uv_kernel<MODE>: 0 product, 4 s_setprio 3, 5 padded with wait
          states, 6 computed twice + compared in the kernel), --lds BYTES (dynamic LDS of the launch: 163840 = a block owns its CU's LDS).
  stress  the 64-step unsynchronised training loop of tests/test_plan_stress_gpu.py (plans kept alive), fence off.

--mask confines the victim (plan) stream and the aggressor streams (main + weight-gradient) to DISJOINT sets of compute units through
hipExtStreamCreateWithCUMask.  Bit i of a CU mask belongs to XCD i % 8 on this chip (the KFD deals the bits round-robin over the XCCs):
  none      torch streams as shipped (plan stream high priority)
  all       CU-mask streams with every bit set on both sides (control: is it the stream TYPE that changes the outcome?)
  spread32  victim = bits 0..31 (4 CUs of every XCD), aggressors = the other 224
  xcd0      victim = every bit with i % 8 == 0 (the 32 CUs of ONE XCD), aggressors = the other seven XCDs
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from virconv_amd import backbone as bb, ops, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

N_CU = 256


def mask_words(bits):
    w = [0] * (N_CU // 32)
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return (ctypes.c_uint32 * len(w))(*w)


def masked_stream(bits):
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    words = mask_words(bits)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value)


def make_streams(kind, dev):
    """-> (victim, main, side) torch streams."""
    if kind == "none":
        return (torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=0))
    every = list(range(N_CU))
    if kind == "all":
        v = a = every
    elif kind == "spread32":
        v, a = every[:32], every[32:]
    elif kind == "xcd0":
        v, a = [i for i in every if i % 8 == 0], [i for i in every if i % 8 != 0]
    else:
        raise SystemExit(f"unknown mask {kind}")
    return masked_stream(v), masked_stream(a), masked_stream(a)


def dset(be, key, val):
    assert be.lib.vc_debug_set(key.encode(), int(val)) == 0, key


LAYERS = {   # name -> (strided convs in front, cin, cout, the table: "subm" of that level | "down" = the next strided conv's)
    "s2": (1, 32, 16, "subm"), "s2b": (1, 16, 16, "subm"), "s3": (2, 64, 32, "subm"), "s3b": (2, 32, 32, "subm"), "s3_64": (2, 64, 64, "subm"),
    "s3_1616": (2, 16, 16, "subm"), "s4": (3, 64, 32, "subm"), "s3down": (1, 32, 64, "down"), "s4down": (2, 64, 64, "down")}


def layer_inputs(be, batch, bs, dev, which):
    """One conv layer on the batch's coordinates: (x, w, dy, rulebook); see LAYERS."""
    import numpy as np
    idx = batch["voxel_coords"].int()
    shape = [int(v) for v in (np.asarray(synth.GRID_SIZE)[::-1] + [1, 0, 0])]
    cur_idx, cur_shape = idx, shape
    n_down, cin, cout, kind = LAYERS[which]
    for _ in range(n_down):
        rb = ops.build_sparse_rulebook(cur_idx, cur_shape, bs, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1)
        cur_idx, cur_shape = rb.out_indices, list(rb.out_shape)
    if kind == "subm":
        rb3 = ops.build_subm_rulebook(cur_idx, cur_shape, (3, 3, 3), 1, False)
    else:
        rb3 = ops.build_sparse_rulebook(cur_idx, cur_shape, bs, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1)
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((rb3.n_in, cin), generator=g).to(dev)
    w = (torch.randn((cout, 27, cin), generator=g) / (27 * cin) ** 0.5).to(dev).reshape((cout, 3, 3, 3, cin))
    dy = torch.randn((rb3.n_out, cout), generator=g).to(dev)
    return x, w, dy, rb3


def emulate(coords, calib, aug, stride, knock, want_intermediates=False):
    """fp32 restatement of project_uv_kernel on the CPU, op for op (every product and sum rounded separately); `knock` names one
    alteration of the inverse augmentation.  -> (n, 2) int64 [u, v]"""
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
    Xs, Ys, Zs = X / sc, Y / sc, Z / sc
    Yf = torch.where(flip, -Ys, Ys)
    X2 = Xs * ca + Yf * (-sa)
    Y2 = Xs * sa + Yf * ca
    if knock == "Y=flipped scaled y":
        Y2 = Yf
    elif knock == "Y=unscaled y":
        Y2 = Y
    elif knock == "Y=scaled y (no flip)":
        Y2 = Ys
    elif knock == "X,Y=unscaled":
        X2, Y2 = X, Y
    elif knock == "X,Y,Z=unscaled (no augmentation)":
        X2, Y2, Zs = X, Y, Z
    elif knock == "rotation lost":
        X2, Y2 = Xs, Yf
    elif knock == "Y=Xs*sa-Yf*ca":
        Y2 = Xs * sa - Yf * ca
    elif knock == "X=Xs":
        X2 = Xs
    elif knock == "X=unscaled x":
        X2 = X
    elif knock == "Z=unscaled z":
        Zs = Z
    M1 = torch.zeros((cal.shape[0], 4, 3), dtype=f32)
    for r in range(4):
        for cc in range(3):
            M1[:, r, cc] = (v2c[:, 0, r] * r0[:, cc, 0] + v2c[:, 1, r] * r0[:, cc, 1]) + v2c[:, 2, r] * r0[:, cc, 2]
    P = M1[b]
    rect = [((X2 * P[:, 0, k] + Y2 * P[:, 1, k]) + Zs * P[:, 2, k]) + P[:, 3, k] for k in range(3)]
    p2t = p2[b]
    hom = [((rect[0] * p2t[:, k, 0] + rect[1] * p2t[:, k, 1]) + rect[2] * p2t[:, k, 2]) + p2t[:, k, 3] for k in range(2)]
    u = torch.nan_to_num(hom[0] / rect[2], nan=0.0, posinf=2.0e9, neginf=-2.0e9).trunc().clamp(-2147483648, 2147483647).long()
    v = torch.nan_to_num(hom[1] / rect[2], nan=0.0, posinf=2.0e9, neginf=-2.0e9).trunc().clamp(-2147483648, 2147483647).long()
    if want_intermediates:
        return {"X0": X, "Y0": Y, "Z0": Z, "Xs": Xs, "Ys": Ys, "Yf": Yf, "Zs": Zs, "X2": X2, "Y2": Y2, "junk": Xs * sa - Yf * ca,
                "Xs*sa": Xs * sa, "Xs*ca": Xs * ca, "ca*Yf": ca * Yf, "sa*Yf": sa * Yf, "ca": ca, "sa": sa, "sc": sc,
                "rect0": rect[0], "rect1": rect[1], "rect2": rect[2], "hom0": hom[0], "hom1": hom[1]}
    return torch.stack([u.clamp(0, 1399) // stride, v.clamp(0, 599) // stride], 1)


KNOCKS = ["none", "Y=flipped scaled y", "Y=unscaled y", "Y=scaled y (no flip)", "X,Y=unscaled", "X,Y,Z=unscaled (no augmentation)",
          "rotation lost", "Y=Xs*sa-Yf*ca", "X=Xs", "X=unscaled x", "Z=unscaled z"]


def run_micro(args):
    dev = torch.device("cuda", 0)
    be = ops.get_backend()
    bs = 4
    batch = bench.make_batch(list(range(bs)), dev, training=True)
    victim, main, side = make_streams(args.mask, dev)
    torch.cuda.set_stream(main)
    be._side = side
    idx = batch["voxel_coords"].int().repeat(args.rep, 1).contiguous()
    n = idx.shape[0]
    params = torch.empty((bs, 32), dtype=torch.float32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    st0 = torch.cuda.current_stream().cuda_stream
    aug = batch["aug_param"].float().contiguous()
    assert be.lib.vc_project_prepare(P(batch["calib"]), P(aug), bs, P(params), ctypes.c_void_p(st0)) == 0
    torch.cuda.synchronize()
    refs = {}
    for s in (1, 2, 4, 8):     # idle GPU, product kernel
        r = torch.empty((n, 3), dtype=torch.int32, device=dev)
        assert be.lib.vc_project_uv(P(idx), n, P(params), bs, s, P(r), None, ctypes.c_void_p(st0)) == 0
        refs[s] = r
    torch.cuda.synchronize()
    if args.explain:
        emu = emulate(idx, batch["calib"], aug, 1, "none")
        same = int((emu.to(dev) == refs[1][:, 1:].long()).all(1).sum())
        print(f"# emulation vs the idle-GPU kernel: {same} of {n} rows equal", flush=True)

    layers = {}

    def layer(which):
        if which not in layers:
            layers[which] = layer_inputs(be, batch, bs, dev, which)
        return layers[which]

    model_state = {}

    def make_chunk(aggr, which):
        """-> closure that enqueues one chunk of aggressor work on the current (main) stream, or None."""
        dset(be, "conv_autopack", 1)
        exact = aggr.endswith("_exact")
        dset(be, "f32_split", 0 if exact else 1)
        dset(be, "bw_split", 0 if exact else 1)
        base = aggr[:-6] if exact else aggr
        if base in ("fwd", "dw", "bwd"):
            x, w, dy, rb = layer(which)
            if base == "fwd":
                return lambda: [be.conv_forward(x, w, rb.pair_fwd, order=rb.order_fwd) for _ in range(args.chunk)]
            if base == "dw":
                return lambda: [be.conv_backward_weight(x, dy, rb.pair_fwd, w.shape) for _ in range(args.chunk)]
            if rb.kind == "subm":
                return lambda: [be.conv_backward_input(dy, w, rb.pair_fwd, rb.n_in, True, rb.centre, rb.rep, order=rb.order_bwd, grp_plan=rb.grp_plan)
                                for _ in range(args.chunk)]
            return lambda: [be.conv_backward_input(dy, w, rb.pair_bwd, rb.n_in, False, order=rb.order_bwd) for _ in range(args.chunk)]
        if base == "valu":
            big = torch.randn((64 << 20,), device=dev)
            return lambda: [torch.add(torch.mul(big, 1.5), 2.0) for _ in range(max(1, args.chunk // 4))]
        if base in ("mm_bf16", "mm_f32"):
            dt = torch.bfloat16 if base == "mm_bf16" else torch.float32
            a_ = torch.randn((8192, 8192), device=dev, dtype=dt)
            b_ = torch.randn((8192, 8192), device=dev, dtype=dt)
            return lambda: [torch.mm(a_, b_) for _ in range(max(1, args.chunk // 8))]
        if base == "step":
            if "m" not in model_state:
                torch.manual_seed(0)
                model = VirConvL8x(bench.MODEL_CFG, 8, synth.GRID_SIZE).to(dev).train()
                opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=True)
                lw = bench.make_loss_weights(dev)
                bb._PLAN_STREAMS[dev.index] = side if args.mask != "none" else bb._plan_stream(dev)   # its own plan: fenced, off the victim stream
                torch.cuda.synchronize()
                batch["inputs_ready_event"] = torch.cuda.Event()
                batch["inputs_ready_event"].record()
                for _ in range(3):
                    bench.train_step(model, opt, batch, lw)
                model_state["m"] = (model, opt, lw)
            model, opt, lw = model_state["m"]
            return lambda: bench.train_step(model, opt, batch, lw)
        assert base == "none", aggr
        return None

    if args.dump:
        return run_dump(args, be, batch, idx, params, refs, victim, make_chunk, aug, dev)
    outs = [torch.empty((n, 3), dtype=torch.int32, device=dev) for _ in range(args.victims)]
    strides = (1, 2, 4, 8)
    for aggr_spec in args.aggr.split(","):
        aggr, _, which = aggr_spec.partition(":")
        which = which or args.layer
        for kv_ in filter(None, args.set.split(",")):     # aggressor-side library switches: key=value
            k_, v_ = kv_.split("=")
            dset(be, k_, int(v_))
        dset(be, "plan_uv_mode", 0); dset(be, "plan_uv_lds", 0)
        chunk = make_chunk(aggr, which)
        if chunk is not None:
            chunk()
        torch.cuda.synchronize()
        for mode in [int(m) for m in args.mode.split(",")]:
            for lds in [int(v) for v in args.lds.split(",")]:
                dset(be, "plan_uv_mode", mode); dset(be, "plan_uv_lds", lds)
                mrefs = refs
                if 300 <= mode < 400:      # a PART of the block: this variant's own idle-GPU result is its reference
                    torch.cuda.synchronize()
                    mrefs = {}
                    for s_ in (1, 2, 4, 8):
                        r = torch.empty((n, 3), dtype=torch.int32, device=dev)
                        assert be.lib.vc_project_uv(P(idx), n, P(params), bs, s_, P(r), None, ctypes.c_void_p(st0)) == 0
                        mrefs[s_] = r
                    torch.cuda.synchronize()
                elif mode in (0, 4, 5, 6) or mode >= 100:  # (200 + k: code shifted by 4 k bytes; 4xx / 5xx: loads restated / checked)   # these share the product kernel's arithmetic: equal on an idle GPU
                    r = torch.empty((n, 3), dtype=torch.int32, device=dev)
                    assert be.lib.vc_project_uv(P(idx), n, P(params), bs, 1, P(r), None, ctypes.c_void_p(st0)) == 0
                    torch.cuda.synchronize()
                    assert torch.equal(r, refs[1]), f"variant kernel {mode} differs from the product kernel on an idle GPU"
                bad_rows = torch.zeros((), dtype=torch.int64, device=dev)
                bad_launches = torch.zeros((), dtype=torch.int64, device=dev)
                lanes = torch.zeros((64,), dtype=torch.int64, device=dev)
                kept = []
                mm0 = ctypes.c_int64(0)
                be.lib.vc_debug_get(b"a17_mismatch", ctypes.byref(mm0))
                marks = []
                t0 = time.perf_counter()
                for r_ in range(args.rounds):
                    if chunk is not None:
                        chunk()
                    with torch.cuda.stream(victim):
                        for k in range(args.victims):
                            assert be.lib.vc_project_uv(P(idx), n, P(params), bs, strides[k % 4], P(outs[k]), None,
                                                        ctypes.c_void_p(victim.cuda_stream)) == 0
                        for k in range(args.victims):
                            bad = (outs[k] != mrefs[strides[k % 4]]).any(1)
                            cnt = bad.sum()
                            bad_rows += cnt
                            bad_launches += (cnt > 0).to(torch.int64)
                            lanes += torch.bincount(torch.nonzero(bad).squeeze(1) % 64, minlength=64)
                            if args.explain and k % 4 == 0 and len(kept) < 64:
                                kept.append((bad.nonzero().squeeze(1), outs[k][bad]))      # stride-1 launches: rows + what they hold
                        ev = torch.cuda.Event()
                        ev.record(victim)
                    marks.append(ev)
                    if len(marks) > 2:
                        marks.pop(0).synchronize()      # the host stays at most two rounds ahead
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                mm = ctypes.c_int64(0)
                be.lib.vc_debug_get(b"a17_mismatch", ctypes.byref(mm))
                ln = lanes.cpu().tolist()
                res = {"exp": "micro", "mask": args.mask, "aggr": aggr, "layer": which if aggr.startswith(("fwd", "dw", "bwd")) else None,
                       "set": args.set or None, "mode": mode, "lds": lds, "launches": args.rounds * args.victims, "rows_per_launch": n,
                       "bad_launches": int(bad_launches), "bad_rows": int(bad_rows), "bad_rows_lanes_48_63": int(sum(ln[48:])),
                       "bad_rows_other_lanes": int(sum(ln[:48])),
                       "in_kernel_mismatch_threads": int(mm.value - mm0.value) if mode in (6, 501) else None, "seconds": round(dt, 2)}
                print(json.dumps(res), flush=True)
                if args.explain and kept:
                    rows = torch.cat([k_[0] for k_ in kept]).cpu()
                    got = torch.cat([k_[1] for k_ in kept]).cpu()[:, 1:].long()
                    expl = {}
                    left = torch.ones((rows.shape[0],), dtype=torch.bool)
                    sub = idx.cpu()[rows]
                    for kn in KNOCKS[1:]:
                        e = emulate(sub, batch["calib"], aug, 1, kn)
                        hit = (e == got).all(1) & left
                        expl[kn] = int(hit.sum())
                        left &= ~hit
                    print(json.dumps({"exp": "explain", "aggr": aggr, "mode": mode, "wrong_rows_examined": int(rows.shape[0]), "reproduced_by": expl,
                                      "unexplained": int(left.sum()), "waves_touched": int(torch.unique(rows // 64).numel())}), flush=True)


def run_dump(args, be, batch, idx, params, refs, victim, make_chunk, aug, dev):
    """project_uv_kernel<MODE, true>: every row stores X, Y, Z (behind the inverse augmentation), rect[0..2], hom[0..1]; the wrong rows'
    stored values against the fp32 emulation -- WHICH register holds WHAT in the lanes that go wrong."""
    n, bs = idx.shape[0], 4
    P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    aggr, _, which = args.aggr.split(",")[0].partition(":")
    chunk = make_chunk(aggr, which or args.layer)
    dbg = torch.zeros((64 + 4 * n * 8,), dtype=torch.int32, device=dev)
    dbg[3] = n
    torch.cuda.synchronize()
    mode = int(args.mode.split(",")[0])
    dset(be, "plan_uv_mode", mode)
    ptr = dbg.data_ptr()
    lo, hi = ptr & 0xFFFFFFFF, ptr >> 32
    dset(be, "plan_uv_dbg_lo", lo - (1 << 32) if lo >= (1 << 31) else lo)
    dset(be, "plan_uv_dbg_hi", hi)
    out = torch.empty((n, 3), dtype=torch.int32, device=dev)
    kept, launches, bad_launches = [], 0, 0
    for r_ in range(args.rounds):
        chunk()
        with torch.cuda.stream(victim):
            for k in range(args.victims):
                assert be.lib.vc_project_uv(P(idx), n, P(params), bs, 1, P(out), None, ctypes.c_void_p(victim.cuda_stream)) == 0
                bad = (out != refs[1]).any(1).nonzero().squeeze(1)
                s8 = dbg[64: 64 + n * 8].view(torch.float32).view(n, 8)[bad]
                kept.append((bad, s8, out[bad]))
                launches += 1
        if r_ % 4 == 3:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dset(be, "plan_uv_dbg_lo", 0); dset(be, "plan_uv_dbg_hi", 0)
    rows = torch.cat([k[0] for k in kept]).cpu()
    s8 = torch.cat([k[1] for k in kept]).cpu()
    bad_launches = sum(1 for k in kept if k[0].numel())
    print(json.dumps({"exp": "dump", "aggr": aggr, "mode": mode, "launches": launches, "bad_launches": bad_launches, "wrong_rows": int(rows.numel()),
                      "lanes": sorted(set((rows % 64).tolist()))[:8]}), flush=True)
    if rows.numel() == 0:
        return
    E = emulate(idx.cpu()[rows], batch["calib"], aug, 1, "none", True)
    names = ["X", "Y", "Z", "rect0", "rect1", "rect2", "hom0", "hom1"]
    want = [E["X2"], E["Y2"], E["Zs"], E["rect0"], E["rect1"], E["rect2"], E["hom0"], E["hom1"]]
    bits = lambda t: t.contiguous().view(torch.int32)   # noqa: E731
    summary = {}
    for c, (nm, w) in enumerate(zip(names, want)):
        summary[nm + " wrong"] = int((bits(s8[:, c]) != bits(w)).sum())
    # what do the wrong X / Y / Z hold?
    for c, nm in ((0, "X"), (1, "Y"), (2, "Z")):
        wrong = bits(s8[:, c]) != bits(want[c])
        holds = {}
        for cand, t in E.items():
            if cand in ("rect0", "rect1", "rect2", "hom0", "hom1"):
                continue
            holds[cand] = int(((bits(s8[:, c]) == bits(t)) & wrong).sum())
        summary[nm + " holds"] = {k: v for k, v in holds.items() if v}
    print(json.dumps({"exp": "dump-summary", **summary}), flush=True)
    for j in range(min(6, rows.numel())):
        print("# row", int(rows[j]), "lane", int(rows[j]) % 64, "stored", [float(v) for v in s8[j]], "| expected", [float(w[j]) for w in want],
              "| Xs Yf Y0", float(E["Xs"][j]), float(E["Yf"][j]), float(E["Y0"][j]), "ca sa sc", float(E["ca"][j]), float(E["sa"][j]), float(E["sc"][j]), flush=True)


def run_generic(args):
    """Is the projection kernel special?  Generic vector-ALU-dense victims -- one torch kernel each, 50-150 dependent fp32 operations per element
    behind one load -- on the victim stream beside the aggressor; every launch compared bit for bit with the same kernel on an idle GPU."""
    dev = torch.device("cuda", 0)
    be = ops.get_backend()
    bs = 4
    batch = bench.make_batch(list(range(bs)), dev, training=True)
    victim, main, side = make_streams(args.mask, dev)
    torch.cuda.set_stream(main)
    be._side = side
    layers = {}
    x_, w_, dy_, rb_ = layer_inputs(be, batch, bs, dev, args.layer)
    dset(be, "conv_autopack", 1); dset(be, "f32_split", 1)
    chunk = (lambda: [be.conv_forward(x_, w_, rb_.pair_fwd, order=rb_.order_fwd) for _ in range(args.chunk)]) if args.aggr != "none" else (lambda: None)
    g = torch.Generator(device="cpu").manual_seed(1)
    n = 1 << 20
    a = (torch.rand((n,), generator=g) * 4 + 0.1).to(dev)
    b = (torch.rand((n,), generator=g) * 2 - 1).to(dev)
    victims = {
        "lgamma": lambda: torch.lgamma(a),
        "erfinv": lambda: torch.special.erfinv(b * 0.999),
        "digamma": lambda: torch.digamma(a),
        "atan2": lambda: torch.atan2(b, a),
        "sinh": lambda: torch.sinh(b * 3),
        "pow": lambda: torch.pow(a, b),
        "div_chain (a / (a + 1) / (a + 2) ...: one fused addcdiv each, 8 launches)": lambda: torch.addcdiv(b, a, a + 1.0),
    }
    chunk()
    torch.cuda.synchronize()
    for name, fn in victims.items():
        ref = fn()
        torch.cuda.synchronize()
        bad_l = torch.zeros((), dtype=torch.int64, device=dev)
        bad_e = torch.zeros((), dtype=torch.int64, device=dev)
        marks = []
        for r_ in range(args.rounds):
            chunk()
            with torch.cuda.stream(victim):
                for k in range(args.victims):
                    y = fn()
                    cnt = (y.view(torch.int32) != ref.view(torch.int32)).sum()
                    bad_e += cnt
                    bad_l += (cnt > 0).to(torch.int64)
                ev = torch.cuda.Event(); ev.record(victim)
            marks.append(ev)
            if len(marks) > 2:
                marks.pop(0).synchronize()
        torch.cuda.synchronize()
        print(json.dumps({"exp": "generic", "aggr": args.aggr + ":" + args.layer, "victim": name, "launches": args.rounds * args.victims,
                          "elements_per_launch": n, "bad_launches": int(bad_l), "bad_elements": int(bad_e)}), flush=True)


def run_stress(args):
    import test_plan_stress_gpu as T
    dev = torch.device("cuda", 0)
    be = ops.get_backend()
    victim, main, side = make_streams(args.mask, dev)
    torch.cuda.set_stream(main)
    be._side = side
    bb._PLAN_STREAMS[dev.index] = victim
    bb.PLAN_GUARD = args.guard
    bench_, dev, batch, model, opt, lw = T._setup()
    if args.exact:
        dset(be, "f32_split", 0); dset(be, "bw_split", 0)
    seen = []
    for t in range(3):
        bench.train_step(model, opt, batch, lw)
    batch["plan_observer"] = lambda rid, plan: seen.append(plan)
    t0 = time.perf_counter()
    for t in range(args.steps):
        torch.manual_seed(5000 + t)
        bench.train_step(model, opt, batch, lw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del batch["plan_observer"]
    bad, kinds, lanes = [], set(), [0] * 64
    bb.PLAN_GUARD = 1
    for t in range(args.steps):
        ref = T._structures(T._rebuild(model, batch, t))
        got = T._structures(seen[t])
        for k in ref:
            if got[k].shape != ref[k].shape or not torch.equal(got[k], ref[k]):
                bad.append((t, k))
                kinds.add(k.split(".")[-1])
                if k.endswith(".uv") and got[k].shape == ref[k].shape:
                    rows = (got[k] != ref[k]).any(1).nonzero().squeeze(1)
                    for ln, c in zip(*[x.tolist() for x in torch.unique(rows % 64, return_counts=True)]):
                        lanes[ln] += c
        seen[t] = None
    print(json.dumps({"exp": "stress", "mask": args.mask, "guard": args.guard, "exact_mfma": bool(args.exact), "steps": args.steps,
                      "bad_structures": len(bad), "bad_steps": len({t for t, _ in bad}), "kinds": sorted(kinds),
                      "wrong_uv_rows_lanes_48_63": sum(lanes[48:]), "wrong_uv_rows_other_lanes": sum(lanes[:48]),
                      "ms_per_step": round(dt / args.steps * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("micro")
    m.add_argument("--aggr", default="step", help="comma list; a conv aggressor may name its layer: fwd:s3")
    m.add_argument("--layer", default="s3")
    m.add_argument("--mode", default="0", help="comma list of victim kernel modes")
    m.add_argument("--lds", default="0", help="comma list of dynamic LDS bytes of the victim launch")
    m.add_argument("--set", default="", help="library switches for the aggressor: key=value,...")
    m.add_argument("--dump", action="store_true", help="diagnostics build of the kernel (every row stores its intermediates): what do the wrong rows hold?")
    m.add_argument("--explain", action="store_true", help="match the wrong rows against an fp32 emulation with single steps altered")
    m.add_argument("--mask", default="none")
    m.add_argument("--rounds", type=int, default=100)
    m.add_argument("--victims", type=int, default=8)
    m.add_argument("--chunk", type=int, default=24, help="aggressor launches per round")
    m.add_argument("--rep", type=int, default=2, help="the batch's coordinate list repeated this many times per victim launch")
    gq = sub.add_parser("generic")
    gq.add_argument("--aggr", default="fwd")
    gq.add_argument("--layer", default="s3")
    gq.add_argument("--mask", default="none")
    gq.add_argument("--rounds", type=int, default=100)
    gq.add_argument("--victims", type=int, default=8)
    gq.add_argument("--chunk", type=int, default=24)
    s = sub.add_parser("stress")
    s.add_argument("--mask", default="none")
    s.add_argument("--guard", type=int, default=0)
    s.add_argument("--steps", type=int, default=64)
    s.add_argument("--exact", type=int, default=0)
    a = ap.parse_args()
    {"micro": run_micro, "stress": run_stress, "generic": run_generic}[a.cmd](a)
