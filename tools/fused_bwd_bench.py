"""A/B of the fused backward experiment (tools/ubench/fused_bwd.hip): ONE gather for dX and dW of a 3-D SubM conv, against the two
launches the product library issues (vc_conv_backward_input + vc_conv_backward_weight), on the real tables of the benchmark batch.

    python tools/fused_bwd_bench.py [--bs 4] [--iters 20]

Prints per layer: separate bwd-input / dW / sum, fused (dX + dW), fused dX half alone, max deviations of the fused results.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bench  # noqa: E402
from virconv_amd import ops, synth  # noqa: E402
from kbench import timeit  # noqa: E402

SRC = os.path.join(ROOT, "tools", "ubench", "fused_bwd.hip")
LIB = os.path.join(ROOT, "tools", "ubench", "libfused_bwd.so")


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
                               "-mllvm", "-amdgpu-mfma-vgpr-form", SRC, "-o", LIB])
    lib = C.CDLL(LIB)
    lib.fb_wp_bytes.restype = C.c_size_t
    lib.fb_wp_bytes.argtypes = [C.c_int] * 3
    lib.fb_partial_bytes.restype = C.c_size_t
    lib.fb_partial_bytes.argtypes = [C.c_int] * 4
    lib.fb_lds_bytes.restype = C.c_size_t
    lib.fb_lds_bytes.argtypes = [C.c_int] * 3
    lib.fb_fused_bwd.restype = C.c_int
    lib.fb_fused_bwd.argtypes = [C.c_void_p] * 8 + [C.c_int64] + [C.c_int] * 5 + [C.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    lib = load()
    dev = torch.device("cuda", 0)
    be = ops.get_backend()
    bs = args.bs
    batch = bench.make_batch(list(range(bs)), dev, training=True)
    idx = batch["voxel_coords"].int()
    shape = [int(v) for v in (np.asarray(synth.GRID_SIZE)[::-1] + [1, 0, 0])]
    g = torch.Generator().manual_seed(0)
    layers = []
    cur_idx, cur_shape = idx, shape
    chans = [(8, 16), (16, 32), (32, 64), (64, 64)]
    for stage, (cin, cout) in enumerate(chans):
        if stage > 0:
            pad = (0, 1, 1) if stage == 3 else (1, 1, 1)
            rb = ops.build_sparse_rulebook(cur_idx, cur_shape, bs, (3, 3, 3), (2, 2, 2), pad, 1)
            cur_idx, cur_shape = rb.out_indices, list(rb.out_shape)
            c1 = cout
        else:
            c1 = cin
        rb3 = ops.build_subm_rulebook(cur_idx, cur_shape, (3, 3, 3), 1, False)
        layers.append((f"s{stage + 1}.d3_conv1 {c1}->{cout // 2}", rb3, c1, cout // 2))
        layers.append((f"s{stage + 1}.d3_conv2 {cout // 2}->{cout // 2}", rb3, cout // 2, cout // 2))
    print(f"{'layer':26s} {'rows':>7s} | {'bwd-in':>7s} {'dW':>7s} {'sum':>7s} | {'fused':>7s} {'blocks':>6s} {'LDS KB':>6s} {'fused dX only':>13s} | "
          f"max |dx - ref| / max|ref|, same for dW")
    st = torch.cuda.current_stream().cuda_stream
    for name, rb, cin, cout in layers:
        if (cin, cout) not in ((8, 8), (16, 16), (32, 16), (32, 32)):
            continue
        kv, n = rb.kv, rb.n_in
        x = torch.randn((n, cin), generator=g).to(dev)
        w = (torch.randn((cout, kv, cin), generator=g) / np.sqrt(kv * cin)).to(dev).reshape((cout,) + tuple(rb.ksize) + (cin,))
        dy = torch.randn((n, cout), generator=g).to(dev)
        dx_ref = be.conv_backward_input(dy, w, rb.pair_fwd, n, True, rb.centre, None, order=rb.order_bwd)
        dw_ref = be.conv_backward_weight(x, dy, rb.pair_fwd, w.shape)
        t_b = timeit(lambda: be.conv_backward_input(dy, w, rb.pair_fwd, n, True, rb.centre, None, order=rb.order_bwd), args.iters)
        t_w = timeit(lambda: be.conv_backward_weight(x, dy, rb.pair_fwd, w.shape), args.iters)
        lds = lib.fb_lds_bytes(kv, cin, cout)
        per_cu = max(1, min(8, (160 * 1024) // lds))
        best = None
        for nblocks in (256 * per_cu, 128 * per_cu):
            nblocks = min(nblocks, (n + 63) // 64)
            wp = torch.empty((lib.fb_wp_bytes(kv, cin, cout) // 4,), dtype=torch.float32, device=dev)
            partial = torch.empty((lib.fb_partial_bytes(kv, cin, cout, nblocks) // 4,), dtype=torch.float32, device=dev)
            dx = torch.empty((n, cin), dtype=torch.float32, device=dev)
            dw = torch.empty((cout, kv, cin), dtype=torch.float32, device=dev)

            def run(do_dw):
                rc = lib.fb_fused_bwd(dy.data_ptr(), x.data_ptr(), rb.pair_fwd.data_ptr(), w.data_ptr(), wp.data_ptr(), dx.data_ptr(),
                                      partial.data_ptr(), dw.data_ptr(), n, kv, cin, cout, nblocks, do_dw, st)
                assert rc == 0, rc

            run(1)
            torch.cuda.synchronize()
            e_dx = float((dx - dx_ref).abs().max() / dx_ref.abs().max())
            e_dw = float((dw - dw_ref.reshape(cout, kv, cin)).abs().max() / dw_ref.abs().max())
            t_f = timeit(lambda: run(1), args.iters)
            t_h = timeit(lambda: run(0), args.iters)
            t_a = timeit(lambda: run(2), args.iters)   # ablation: transpose + dW MFMAs, no LDS accumulation
            if best is None or t_f < best[0]:
                best = (t_f, nblocks, t_h, e_dx, e_dw, t_a)
        t_f, nblocks, t_h, e_dx, e_dw, t_a = best
        print(f"{name:26s} {n:7d} | {t_b:7.1f} {t_w:7.1f} {t_b + t_w:7.1f} | {t_f:7.1f} {nblocks:6d} {lds / 1024:6.1f} {t_h:13.1f} | "
              f"{e_dx:.2e} {e_dw:.2e} | without the LDS accumulation {t_a:7.1f}")
        assert e_dx < 1e-5 and e_dw < 1e-4, "fused results deviate"


if __name__ == "__main__":
    main()
