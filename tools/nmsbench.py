"""Timing of the rotated IoU / NMS kernels on KITTI-like proposal sets (events, average of `iters` launches).

    python tools/nmsbench.py [--iters 20]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_iou3d_cpu import kitti_like_boxes  # noqa: E402
from virconv_amd import iou3d_nms  # noqa: E402


def timeit(fn, iters):
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
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    for n, spread in ((512, 40.0), (4096, 70.0), (9000, 70.0)):
        boxes = torch.from_numpy(kitti_like_boxes(n, n, spread=spread)).cuda()
        scores = torch.rand(n, device="cuda")
        order = scores.sort(0, descending=True)[1]
        sb = boxes[order].contiguous()
        t_iou = timeit(lambda: iou3d_nms.boxes_iou3d_gpu(boxes[:128], sb), args.iters)
        t_nms = timeit(lambda: iou3d_nms.nms_sorted(sb, 0.1), args.iters)
        t_nrm = timeit(lambda: iou3d_nms.nms_sorted(sb, 0.1, rotated=False), args.iters)
        keep, num = iou3d_nms.nms_sorted(sb, 0.1)
        print(f"n = {n:5d}: iou3d (128 x n) {t_iou:8.1f} us   rotated NMS {t_nms:8.1f} us ({int(num)} kept)   axis-aligned NMS {t_nrm:8.1f} us")


if __name__ == "__main__":
    main()
