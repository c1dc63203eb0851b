"""Micro-benchmark of the RoI-grid-pooling operators at VirConv-L scale: x_conv3 / x_conv4 sized tensors of
the bench frames, 128 RoIs x 6^3 grid points per frame (ROI_GRID_POOL), both query scales, nsample 16, C = 32."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from virconv_amd import ops, synth  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--rois", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    be = ops.get_backend()
    batch = bench.make_batch(list(range(args.bs)), dev, training=False)
    idx0 = batch["voxel_coords"].int()
    g = torch.Generator(device="cpu").manual_seed(0)
    print(f"{'level':8s} {'N':>7s} {'M':>7s} | {'index us':>9s} | {'range':>9s} {'query us':>9s} {'empty %':>8s} {'Mq/s':>7s} | "
          f"{'group us':>9s} {'GB/s':>6s} | {'grad us':>8s}")
    for name, stride, shape, radii in (("x_conv3", 4, (21, 400, 352), (0.4, 0.8)), ("x_conv4", 8, (10, 200, 176), (0.8, 1.6))):
        c = torch.cat([idx0[:, :1], idx0[:, 1:] // stride], 1)
        lin = ((c[:, 0].long() * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]
        lin = torch.unique(lin)                                            # ascending, like a strided conv's outputs
        x = lin % shape[2]; r = lin // shape[2]; y = r % shape[1]; r = r // shape[1]; z = r % shape[0]; b = r // shape[0]
        idx = torch.stack([b, z, y, x], 1).int().contiguous()
        n = idx.shape[0]
        vs = torch.tensor(synth.VOXEL_SIZE, device=dev).float() * stride
        lo = torch.tensor(synth.POINT_CLOUD_RANGE[:3], device=dev).float()
        xyz = ((idx[:, [3, 2, 1]].float() + 0.5) * vs + lo).contiguous()
        m_per = args.rois * 216
        cnt = torch.bincount(idx[:, 0].long(), minlength=args.bs).int()
        qs, cs = [], []
        for bb in range(args.bs):
            sel = torch.nonzero(idx[:, 0] == bb)[:, 0]
            pick = sel[torch.randint(0, sel.numel(), (m_per,), generator=g).to(dev)]
            q = xyz[pick] + (torch.rand((m_per, 3), generator=g).to(dev) - 0.5) * 2.0
            qc = torch.floor((q - lo) / vs).int()
            qs.append(q)
            cs.append(torch.cat([torch.full((m_per, 1), bb, dtype=torch.int32, device=dev), qc[:, [2, 1, 0]]], 1))
        q, coords = torch.cat(qs).contiguous(), torch.cat(cs).contiguous()
        m = q.shape[0]
        new_cnt = torch.full((args.bs,), m_per, dtype=torch.int32, device=dev)
        t_index = timeit(lambda: be.voxel_index_build(idx, args.bs, shape), args.iters)
        ws = be.voxel_index_build(idx, args.bs, shape)
        feats = torch.randn((n, 32), generator=g).to(dev)
        for rng, radius in (([2, 2, 2], radii[0]), ([4, 4, 4], radii[1])):
            t_q = timeit(lambda: be.voxel_query(ws, n, args.bs, shape, xyz, q, coords, rng, radius, 16), args.iters)
            out, empty = be.voxel_query(ws, n, args.bs, shape, xyz, q, coords, rng, radius, 16)
            starts = torch.cumsum(cnt.long(), 0) - cnt.long()
            loc = (out - torch.repeat_interleave(starts, new_cnt.long()).int()[:, None]).contiguous()
            loc[empty] = 0
            t_g = timeit(lambda: be.group_points(feats, cnt, loc, new_cnt), args.iters)
            go = torch.randn((m, 32, 16), generator=g).to(dev)
            go[empty] = 0                                                    # the module zeroes emptied balls (keep mask)
            t_b = timeit(lambda: be.group_points_grad(go, loc, new_cnt, cnt, n), args.iters)
            gbytes = m * 16 * 32 * 4 * 2 / 1e9                               # rows read + (M, C, nsample) written
            print(f"{name:8s} {n:7d} {m:7d} | {t_index:9.1f} | {str(rng):>9s} {t_q:9.1f} {100 * float(empty.float().mean()):8.1f} "
                  f"{m / t_q:7.1f} | {t_g:9.1f} {gbytes / (t_g * 1e-6):6.0f} | {t_b:8.1f}")


if __name__ == "__main__":
    main()
