"""Kernel micro-benchmark: every conv layer shape of VirConv-L on a synthetic KITTI batch, one kernel at a time.

    python tools/kbench.py [--bs 4] [--iters 20] [--only fwd|bwd|dw|all]

For each layer: N_in, N_out, active pairs P, and per kernel (forward gather-GEMM, backward-input gather-GEMM,
weight-gradient) the average HIP-event time, achieved algorithmic TFLOP/s (2*P*Cin*Cout / t) and the fraction of the
157.3 TFLOP/s fp32-MFMA peak, plus algorithmic GB/s (SURVEY §8d byte formulas).  Used to iterate on kernel variants;
bench.py remains the whole-step measurement.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from virconv_amd import ops, synth  # noqa: E402

PEAK = 157.3


def timeit(fn, iters, reps=3):
    """us per call: the best of `reps` averages over `iters` back-to-back calls.  (One average is not robust: two round-6 tables carried a
    single 27-30 ms pause in one layer's 20 launches -- 1400-1600 us where every rerun says 56 / 92.  The pause was Python's: a generation-2
    garbage collection over torch's long-lived objects; main() freezes the collector now, as bench.py does.)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    import gc
    gc.collect()
    gc.freeze()      # as bench.py: keeps generation-2 collections (tens of ms over torch's ~10^6 long-lived objects) out of the timed loops
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="all")
    ap.add_argument("--variant", type=int, default=0, help="gather-GEMM variant (vc_debug_set conv_variant); 0 = default")
    ap.add_argument("--no-xcd", action="store_true", help="disable the XCD-aware block swizzle of the gather-GEMM")
    ap.add_argument("--rt", type=int, default=0, help="v2 row tiles per wave (vc_debug_set conv_rt); 0 = heuristic")
    ap.add_argument("--operand", default="f32", choices=["f32", "f16", "bf16"], help="MFMA operand type")
    ap.add_argument("--bw-legacy", action="store_true", help="offset-major block order in the weight-gradient kernel")
    ap.add_argument("--no-window", action="store_true", help="never use the LDS row-window gather-GEMM (A/B)")
    ap.add_argument("--window", action="store_true", help="use the LDS row-window gather-GEMM where the table is sorted (A/B)")
    ap.add_argument("--wdma", type=int, default=0, help="window kernel: W images through the LDS-DMA engine (vc_debug_set conv_wdma)")
    ap.add_argument("--winrows", type=int, default=32, help="window kernel: rows per wave window, 32 | 24 (vc_debug_set conv_winrows)")
    ap.add_argument("--nw", type=int, default=0, help="direct kernel: waves per block 4 | 8 (vc_debug_set conv_nw); 0 = library default")
    ap.add_argument("--v4", type=int, default=-1, help="wave-autonomous gather-GEMM (vc_debug_set conv_v4): 0 never | 1 every eligible shape | 2 library table; -1 = leave the default")
    ap.add_argument("--autopack", action="store_true", help="repack the weights into fragment order before every conv launch (vc_debug_set conv_autopack; the pack launch is inside the timing)")
    ap.add_argument("--ablate", type=int, default=0, help="v4 ablations (vc_debug_set conv_v4_ablate; wrong results; library built with VIRCONV_HIPCC_EXTRA=-DVC_EXPERIMENTS): 1 no MFMA | 2 coalesced gathers | 3 W from one image | 4 = 2 + 3")
    ap.add_argument("--pf", type=int, default=1, help="v4 gather prefetch distance (vc_debug_set conv_v4_pf; values other than 1 need a -DVC_EXPERIMENTS build): 1 | 2 | 4 | 11 = interleaved")
    ap.add_argument("--v5", type=int, default=0, help="loader / MFMA wave-role gather-GEMM (vc_debug_set conv_v5; needs --autopack)")
    ap.add_argument("--dxs", type=int, default=-1, help="dx shift in the LDS-staged kernel (vc_debug_set conv_dxs): 0 | 1; -1 = library default; needs --autopack")
    ap.add_argument("--layers", default="", help="comma-separated substrings: only time layers whose name contains one of them")
    ap.add_argument("--split", type=int, default=1, help="gather-GEMM products: 1 (library default) = six bf16 split terms on the MFMA (vc_debug_set f32_split; needs a weight image: implies --autopack), also prints each layer's max deviation from the exact-fp32 kernels; 0 = v_mfma_f32_16x16x4_f32")
    ap.add_argument("--bwsplit", type=int, default=1, help="weight-gradient products: 1 (library default) = six bf16 split terms (vc_debug_set bw_split), prints each layer's deviation from the exact-fp32 kernel; 0 = v_mfma_f32_16x16x4_f32")
    ap.add_argument("--il", action="store_true", help="forward convs (channel counts multiples of 16) also with the source features in the 16-row interleaved layout (VC_CONV_SRC_INTERLEAVED; implies --autopack)")
    args = ap.parse_args()
    ops.WINDOW_GATHER = bool(args.window)
    dev = torch.device("cuda", 0)
    be = ops.get_backend()
    if args.variant:
        assert be.lib.vc_debug_set(b"conv_variant", args.variant) == 0
    assert be.lib.vc_debug_set(b"conv_rt", args.rt) == 0
    assert be.lib.vc_debug_set(b"bw_legacy_order", 1 if args.bw_legacy else 0) == 0
    assert be.lib.vc_debug_set(b"conv_wdma", args.wdma) == 0 and be.lib.vc_debug_set(b"conv_winrows", args.winrows) == 0
    if args.nw:
        assert be.lib.vc_debug_set(b"conv_nw", args.nw) == 0
    if args.v4 >= 0:
        assert be.lib.vc_debug_set(b"conv_v4", args.v4) == 0
    assert be.lib.vc_debug_set(b"conv_autopack", 1 if (args.autopack or args.il or args.split) else 0) == 0
    assert be.lib.vc_debug_set(b"f32_split", args.split) == 0
    assert be.lib.vc_debug_set(b"bw_split", args.bwsplit) == 0
    assert be.lib.vc_debug_set(b"conv_v4_ablate", args.ablate) == 0
    assert be.lib.vc_debug_set(b"conv_v4_pf", args.pf) == 0
    assert be.lib.vc_debug_set(b"conv_v5", args.v5) == 0
    if args.dxs >= 0:
        assert be.lib.vc_debug_set(b"conv_dxs", args.dxs) == 0
    torch.zeros(1, device=dev)
    assert be.lib.vc_debug_set(b"xcd_swizzle_off", 1 if args.no_xcd else 0) == 0
    for kv_ in filter(None, os.environ.get("VIRCONV_DEBUG_SET", "").split(",")):   # "key=value,..." -> vc_debug_set
        key, val = kv_.split("=")
        assert be.lib.vc_debug_set(key.encode(), int(val)) == 0, kv_
    batch = bench.make_batch(list(range(args.bs)), dev, training=True)
    idx = batch["voxel_coords"].int()
    shape = [int(v) for v in (np.asarray(synth.GRID_SIZE)[::-1] + [1, 0, 0])]
    bs = args.bs
    calib, trans = batch["calib"], batch["aug_param"]
    g = torch.Generator(device="cpu").manual_seed(0)

    layers = []  # (name, rulebook, cin, cout)
    cur_idx, cur_shape = idx, shape
    chans = [(8, 16), (16, 32), (32, 64), (64, 64)]
    for stage, (cin, cout) in enumerate(chans):
        stride = 2 ** stage
        if stage > 0:
            pad = (0, 1, 1) if stage == 3 else (1, 1, 1)
            rb = ops.build_sparse_rulebook(cur_idx, cur_shape, bs, (3, 3, 3), (2, 2, 2), pad, 1)
            layers.append((f"s{stage + 1}.down {cin}->{cout}", rb, cin, cout))
            cur_idx, cur_shape = rb.out_indices, list(rb.out_shape)
            c1 = cout
        else:
            c1 = cin
        rb3 = ops.build_subm_rulebook(cur_idx, cur_shape, (3, 3, 3), 1, False)
        layers.append((f"s{stage + 1}.d3_conv1 {c1}->{cout // 2}", rb3, c1, cout // 2))
        layers.append((f"s{stage + 1}.d3_conv2 {cout // 2}->{cout // 2}", rb3, cout // 2, cout // 2))
        uv = ops.project_uv(cur_idx, calib, trans, bs, stride)
        rb2 = ops.build_subm_rulebook(uv, [1600, 600], (3, 3), 1, True)
        layers.append((f"s{stage + 1}.d2_conv {cout // 2}->{cout // 2} (2D)", rb2, cout // 2, cout // 2))
    rbo = ops.build_sparse_rulebook(cur_idx, cur_shape, bs, (3, 1, 1), (2, 1, 1), (0, 0, 0), 1)
    layers.append(("conv_out 64->64", rbo, 64, 64))

    if args.only in ("all", "rulebook"):
        # geometry kernels (plan stream): hash build + SubM rulebook, and the whole strided rulebook (incl. its count read)
        print("rulebook timings (us per build, includes the hash build / the host count read):")
        seen = set()
        for name, rb, _, _ in layers:
            if id(rb) in seen:
                continue
            seen.add(id(rb))
            if rb.kind == "subm":
                t = timeit(lambda: be.subm_rulebook(rb.in_indices, rb.in_shape, rb.ksize, rb.dilation, want_rep=rb.rep is not None), args.iters)
            else:
                t = timeit(lambda: be.sparse_rulebook(rb.in_indices, rb.in_shape, bs, rb.ksize, rb.stride, rb.padding, rb.dilation), args.iters)
            print(f"  {name:34s} {rb.kind:6s} N_in {rb.n_in:7d} N_out {rb.n_out:7d}  {t:8.1f}")
        if args.only == "rulebook":
            return
    print(f"{'layer':34s} {'N_in':>7s} {'N_out':>7s} {'P/N':>5s} | {'fwd us':>8s} {'TF':>6s} {'%pk':>5s} {'GB/s':>6s} | "
          f"{'bwd us':>8s} {'TF':>6s} {'%pk':>5s} | {'dW us':>8s} {'TF':>6s} {'%pk':>5s}")
    tot = {"fwd": 0.0, "bwd": 0.0, "dw": 0.0, "flops": 0.0}
    for name, rb, cin, cout in layers:
        if args.layers and not any(t in name for t in args.layers.split(",")):
            continue
        kv = rb.kv
        x = torch.randn((rb.n_in, cin), generator=g).to(dev)
        w = (torch.randn((cout, kv, cin), generator=g) / np.sqrt(kv * cin)).to(dev).reshape((cout,) + tuple(rb.ksize) + (cin,))
        dy = torch.randn((rb.n_out, cout), generator=g).to(dev)
        pairs = int((rb.pair_fwd >= 0).sum().item())
        flops = 2.0 * pairs * cin * cout
        byts = 4.0 * (rb.n_in * cin + rb.n_out * cout + kv * cin * cout) + 4.0 * kv * rb.n_out
        res = {}
        srt = rb.sorted_rows and not args.no_window
        dev_note = ""
        if args.split and args.only in ("all", "fwd", "bwd"):
            def both(fn):
                assert be.lib.vc_debug_set(b"f32_split", 0) == 0
                a = fn()
                assert be.lib.vc_debug_set(b"f32_split", args.split) == 0
                b = fn()
                return float((a - b).abs().max() / a.abs().max())
            if args.only in ("all", "fwd"):
                dev_note += f" | split dev fwd {both(lambda: be.conv_forward(x, w, rb.pair_fwd, order=rb.order_fwd)):.1e}"
            if args.only in ("all", "bwd"):
                if rb.kind == "subm":
                    dev_note += f" bwd {both(lambda: be.conv_backward_input(dy, w, rb.pair_fwd, rb.n_in, True, rb.centre, rb.rep, order=rb.order_bwd, grp_plan=rb.grp_plan)):.1e}"
                else:
                    dev_note += f" bwd {both(lambda: be.conv_backward_input(dy, w, rb.pair_bwd, rb.n_in, False, order=rb.order_bwd)):.1e}"
        if args.only in ("all", "fwd"):
            res["fwd"] = timeit(lambda: be.conv_forward(x, w, rb.pair_fwd, order=rb.order_fwd, operand=args.operand, sorted_rows=srt), args.iters)
        if args.il and args.only in ("all", "fwd") and cin % 16 == 0 and cout % 16 == 0:
            xil = be.interleave_rows(x)
            y_ref = be.conv_forward(x, w, rb.pair_fwd, order=rb.order_fwd, operand=args.operand)
            y_il = be.conv_forward(xil, w, rb.pair_fwd, order=rb.order_fwd, operand=args.operand, interleaved_rows=rb.n_in)
            assert torch.equal(y_ref, y_il), f"{name}: interleaved source changed the result"
            res["il"] = timeit(lambda: be.conv_forward(xil, w, rb.pair_fwd, order=rb.order_fwd, operand=args.operand,
                                                       interleaved_rows=rb.n_in), args.iters)
        if args.only in ("all", "bwd"):
            if rb.kind == "subm":
                res["bwd"] = timeit(lambda: be.conv_backward_input(dy, w, rb.pair_fwd, rb.n_in, True, rb.centre, rb.rep, order=rb.order_bwd, operand=args.operand, sorted_rows=srt, grp_plan=rb.grp_plan), args.iters)
            else:
                res["bwd"] = timeit(lambda: be.conv_backward_input(dy, w, rb.pair_bwd, rb.n_in, False, order=rb.order_bwd, operand=args.operand), args.iters)
        if args.bwsplit and args.only in ("all", "dw") and not (rb.kind == "subm" and rb.rep is not None):
            assert be.lib.vc_debug_set(b"bw_split", 0) == 0
            d0 = be.conv_backward_weight(x, dy, rb.pair_fwd, w.shape)
            assert be.lib.vc_debug_set(b"bw_split", args.bwsplit) == 0
            d1 = be.conv_backward_weight(x, dy, rb.pair_fwd, w.shape)
            dev_note += f" | dW split dev {float((d0 - d1).abs().max() / d0.abs().max()):.1e}"
        if args.only in ("all", "dw"):
            if rb.kind == "subm" and rb.rep is not None and rb.grp_plan is not None:   # duplicate-pixel table: dW over representatives
                grp = be.group_sum_sorted(dy, rb.grp_plan)
                res["dw"] = timeit(lambda: be.conv_backward_weight(x, dy, rb.pair_fwd, w.shape, operand=args.operand, rep=rb.rep,
                                                                   centre=rb.centre, dy_grp=grp), args.iters)
            else:
                res["dw"] = timeit(lambda: be.conv_backward_weight(x, dy, rb.pair_fwd, w.shape, operand=args.operand), args.iters)

        def tf(us):
            return flops / (us * 1e-6) / 1e12

        f, b, d = res.get("fwd"), res.get("bwd"), res.get("dw")
        line = f"{name + (' [win]' if srt else ''):34s} {rb.n_in:7d} {rb.n_out:7d} {pairs / max(rb.n_out, 1):5.2f} | "
        line += (f"{f:8.1f} {tf(f):6.2f} {100 * tf(f) / PEAK:5.1f} {byts / (f * 1e-6) / 1e9:6.0f} | " if f else " " * 34 + "| ")
        line += (f"{b:8.1f} {tf(b):6.2f} {100 * tf(b) / PEAK:5.1f} | " if b else " " * 23 + "| ")
        line += (f"{d:8.1f} {tf(d):6.2f} {100 * tf(d) / PEAK:5.1f}" if d else "")
        if res.get("il"):
            line += f" | fwd interleaved src {res['il']:8.1f} us ({100 * tf(res['il']) / PEAK:4.1f} %pk, {res['fwd'] / res['il']:.2f}x)"
        line += dev_note
        if ops.ROW_ORDER != "none" and not args.layers and not args.split:
            line += f" | ord {timeit(lambda: be.row_order(rb.pair_fwd, window=ops.ROW_ORDER_WINDOW), args.iters):6.1f}"
        print(line)
        for k in ("fwd", "bwd", "dw"):
            if res.get(k):
                tot[k] += res[k]
        tot["flops"] += flops
    print(f"TOTAL  fwd {tot['fwd']:.0f} us  bwd-in {tot['bwd']:.0f} us  dW {tot['dw']:.0f} us   algorithmic GFLOP/pass {tot['flops'] / 1e9:.2f}"
          f"  -> fwd {tot['flops'] / max(tot['fwd'], 1e-9) / 1e6:.1f} TF, all three {3 * tot['flops'] / max(tot['fwd'] + tot['bwd'] + tot['dw'], 1e-9) / 1e6:.1f} TF")


if __name__ == "__main__":
    main()
