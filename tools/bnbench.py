"""Stand-alone rate of the BatchNorm apply / backward kernels at the benchmark's tensor sizes (rows x channels of VirConv-L, bs 4).
    python tools/bnbench.py
bn_apply: reads x, writes y (2 x 4 n c bytes); bn_relu_backward = reduce (reads x, dy) + finalize + dx (reads x, dy, writes dx: 3 x 4 n c)."""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from virconv_amd import ops  # noqa: E402
from virconv_amd._lib import check  # noqa: E402

SIZES = [(133578, 8), (267156, 16), (310351, 16), (310351, 32), (194944, 32), (194944, 64), (75991, 32), (75991, 64), (64058, 64)]


def timeit(fn, iters=50, reps=3):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e30
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
    gc.collect()
    gc.freeze()
    be = ops.get_backend()
    lib = be.lib
    dev = torch.device("cuda", 0)
    st = lambda: torch._C._cuda_getCurrentRawStream(0)   # noqa: E731
    print(f"{'rows':>8} {'c':>3} | {'apply us':>9} {'TB/s':>6} | {'bwd (3 launches) us':>20} {'TB/s of 5 passes':>17}")
    for n, c in SIZES:
        x = torch.randn((n, c), device=dev)
        dy = torch.randn((n, c), device=dev)
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        mean, var = x.mean(0).contiguous(), x.var(0, unbiased=False).contiguous()
        gamma, beta = torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) - 0.5
        dgamma, dbeta = torch.empty(c, device=dev), torch.empty(c, device=dev)
        wsb = lib.vc_bn_workspace_bytes(n, c)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        t_apply = timeit(lambda: check(lib.vc_bn_apply_relu(x.data_ptr(), n, c, mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                           1e-3, 1, y.data_ptr(), c, 0, st()), "apply"))
        t_bwd = timeit(lambda: check(lib.vc_bn_relu_backward(x.data_ptr(), dy.data_ptr(), c, 0, n, c, mean.data_ptr(), var.data_ptr(),
                                                             gamma.data_ptr(), beta.data_ptr(), 1e-3, 1, dx.data_ptr(), dgamma.data_ptr(),
                                                             dbeta.data_ptr(), None, ws.data_ptr(), wsb, st()), "bwd"))
        b = 4.0 * n * c
        print(f"{n:8d} {c:3d} | {t_apply:9.1f} {2 * b / t_apply * 1e-6:6.2f} | {t_bwd:20.1f} {5 * b / t_bwd * 1e-6:17.2f}")


if __name__ == "__main__":
    main()
