"""HeightCompression -> first BEV conv (ZeroPad2d(1) + Conv2d(256 -> 64, k3, p0), base_bev_backbone.py:31-36),
measured end to end from the backbone's sparse output on one GPU:

    python tools/bevbench.py [--bs 4]

  A  reference recipe   .dense() (write-once fill) -> view (B, 256, 200, 176) -> nn.ZeroPad2d(1) -> conv
  B  fused border       vc_to_dense_fill_padded -> (B, 256, 202, 178) -> conv (padding 0, no pad kernel)
each in NCHW and channels_last (what MIOpen picks differs per layout), forward and forward+backward, and
  C  sparse stem        vc_bev_pairs -> gather-GEMM passes over the sparse rows -> vc_nhwc_to_nchw (forward; virconv_amd/bev_stem.py),
                        conv only and conv + BatchNorm(eval) + ReLU, against the same ops on the dense map.
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
from virconv_amd.backbone import VirConvL8x  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    batch = bench.make_batch(list(range(args.bs)), dev, training=False)
    torch.manual_seed(0)
    model = VirConvL8x(dict(bench.MODEL_CFG, LAYER_DISCARD_MODE="spconv2_noop"), 8, synth.GRID_SIZE).to(dev).eval()
    with torch.no_grad():
        t = model(dict(batch, voxel_features=batch["voxel_features"].clone()))["encoded_spconv_tensor"]
    feats, idx, shape, bs = t.features.detach().clone(), t.indices, t.spatial_shape, t.batch_size
    print(f"encoded tensor: {feats.shape[0]} rows x {feats.shape[1]} ch, spatial {shape}, bs {bs}")
    conv = torch.nn.Conv2d(256, 64, 3, stride=1, padding=0, bias=False).to(dev)
    pad = torch.nn.ZeroPad2d(1)

    for fmt_name, fmt in (("NCHW", torch.contiguous_format), ("channels_last", torch.channels_last)):
        conv_f = conv.to(memory_format=fmt)

        def recipe_a(train):
            f = feats.clone().requires_grad_(train)
            d = ops.to_dense(f, idx, shape, bs)
            x = pad(d.view(bs, -1, d.shape[-2], d.shape[-1]))
            if fmt is torch.channels_last:
                x = x.contiguous(memory_format=fmt)
            y = conv_f(x)
            if train:
                y.sum().backward()
            return y

        def recipe_b(train):
            f = feats.clone().requires_grad_(train)
            d = ops.to_dense(f, idx, shape, bs, pad=(1, 1))
            x = d.view(bs, -1, d.shape[-2], d.shape[-1])
            if fmt is torch.channels_last:
                x = x.contiguous(memory_format=fmt)
            y = conv_f(x)
            if train:
                y.sum().backward()
            return y

        with torch.no_grad():
            ya, yb = recipe_a(False), recipe_b(False)
            assert torch.equal(ya, yb), "padded dense changed the conv result"
            ta, tb = timeit(lambda: recipe_a(False)), timeit(lambda: recipe_b(False))
        tta, ttb = timeit(lambda: recipe_a(True)), timeit(lambda: recipe_b(True))
        only_conv = timeit(lambda: conv_f(ya.new_zeros((bs, 256, 202, 178)).contiguous(memory_format=fmt)))
        print(f"{fmt_name:14s} forward: dense+ZeroPad2d+conv {ta:8.1f} us | padded dense+conv {tb:8.1f} us ({ta - tb:+.1f}) ; "
              f"fwd+bwd: {tta:8.1f} vs {ttb:8.1f} us ({tta - ttb:+.1f}) ; zeros+conv alone {only_conv:8.1f} us")


    # ---- C: the conv (and conv + BN + ReLU) on the sparse rows
    from virconv_amd.bev_stem import SparseBEVStem, pack_stem_weight
    be = ops.get_backend()
    passes = pack_stem_weight(conv.weight, int(shape[0]))
    with torch.no_grad():
        dense_map = ops.to_dense(feats, idx, shape, bs, pad=(1, 1))
        want = conv(dense_map.view(bs, -1, dense_map.shape[-2], dense_map.shape[-1]))
        got = be.bev_stem_conv(feats, idx, shape, bs, passes, 64)
        err = float((got - want).abs().max() / want.abs().max())
        t_sparse = timeit(lambda: be.bev_stem_conv(feats, idx, shape, bs, passes, 64))
        t_dense = timeit(lambda: conv(ops.to_dense(feats, idx, shape, bs, pad=(1, 1)).view(bs, -1, 202, 178)))
        blk = torch.nn.Sequential(torch.nn.ZeroPad2d(1), conv, torch.nn.BatchNorm2d(64, eps=1e-3, momentum=0.01).to(dev), torch.nn.ReLU()).eval()
        stem = SparseBEVStem(blk)

        class _T:   # the three attributes the stem reads
            features, indices, spatial_shape, batch_size = feats, idx, shape, bs
        t_stem = timeit(lambda: stem(_T))
        t_blk = timeit(lambda: blk[3](blk[2](conv(ops.to_dense(feats, idx, shape, bs, pad=(1, 1)).view(bs, -1, 202, 178)))))
    print(f"sparse stem    forward: conv on the sparse rows {t_sparse:8.1f} us vs padded dense + MIOpen conv {t_dense:8.1f} us "
          f"(max rel diff {err:.1e}); conv+BN+ReLU {t_stem:8.1f} us vs {t_blk:8.1f} us")

    # ---- D (round 5): TRAINING through the stem -- forward + backward (d features, d weight, BatchNorm in train mode), sparse vs dense
    blk_t = torch.nn.Sequential(torch.nn.ZeroPad2d(1), conv, torch.nn.BatchNorm2d(64, eps=1e-3, momentum=0.01).to(dev), torch.nn.ReLU()).train()
    stem_t = SparseBEVStem(blk_t)

    def train_sparse():
        f = feats.clone().requires_grad_(True)

        class _Tt:
            features, indices, spatial_shape, batch_size = f, idx, shape, bs
        conv.weight.grad = None
        stem_t(_Tt).sum().backward()

    def train_dense():
        f = feats.clone().requires_grad_(True)
        conv.weight.grad = None
        d = ops.to_dense(f, idx, shape, bs, pad=(1, 1))
        blk_t[3](blk_t[2](conv(d.view(bs, -1, 202, 178)))).sum().backward()

    t_ts, t_td = timeit(train_sparse, 10), timeit(train_dense, 10)
    print(f"sparse stem    fwd+bwd (train-mode BatchNorm, d features + d weight): {t_ts:8.1f} us vs the dense recipe {t_td:8.1f} us")


if __name__ == "__main__":
    main()
