"""Rotated-box IoU and NMS -- host-side mirror of the reference interface
pcdet/ops/iou3d_nms/iou3d_nms_utils.py, over the HIP kernels of csrc/nms_kernels.hip (vc_boxes_* / vc_nms).

Same function names, arguments and return values:

  boxes_iou_bev(boxes_a, boxes_b)              iou3d_nms_utils.py:31-45     (N, M) BEV IoU
  boxes_iou3d_gpu(boxes_a, boxes_b)            iou3d_nms_utils.py:67-99     (N, M) 3-D IoU -- ONE launch instead of 1 + 5 torch kernels
  nms_gpu(boxes, scores, thresh, pre_maxsize)  iou3d_nms_utils.py:102-118   -> (selected indices, None)
  nms_normal_gpu(boxes, scores, thresh)        iou3d_nms_utils.py:121-135   -> (selected indices, None)
  boxes_bev_iou_cpu(boxes_a, boxes_b)          iou3d_nms_utils.py:12-28     numpy / CPU tensors in and out; computed on the GPU
                                               (this package has no CPU arithmetic; the reference's CPU loop is the oracle's job)
  boxes_dis(boxes_a, boxes_b)                  iou3d_nms_utils.py:47-64     plain torch, as the reference

MI355X-first difference in NMS: the reference copies the N x N/64 suppression mask to the host and selects there
(src/iou3d_nms.cpp:125-150); here the selection runs on the device (one wave, register-resident removal words) and only the
count of survivors crosses PCIe -- 8 bytes, because the caller slices `order[keep[:num_out]]` like the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .backend_hip import _need, _ptr, _stream


def _lib_handle():
    return _lib.load()


def _pair_matrix(fn_name: str, boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a = _need(boxes_a, torch.float32, "boxes_a")
    b = _need(boxes_b, torch.float32, "boxes_b")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.check(getattr(_lib_handle(), fn_name)(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream()), fn_name)
    return out


def boxes_overlap_bev(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """(N, M) BEV overlap areas (what the reference's boxes_iou3d_gpu gets from iou3d_nms_cuda.boxes_overlap_bev_gpu)."""
    return _pair_matrix("vc_boxes_overlap_bev", boxes_a, boxes_b)


def boxes_iou_bev(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    return _pair_matrix("vc_boxes_iou_bev", boxes_a, boxes_b)


def boxes_iou3d_gpu(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    return _pair_matrix("vc_boxes_iou3d", boxes_a, boxes_b)


def boxes_dis(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    return torch.cdist(boxes_a[:, 0:2].unsqueeze(0), boxes_b[:, 0:2].unsqueeze(0)).squeeze(0)


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    is_numpy = isinstance(boxes_a, np.ndarray)
    a = torch.as_tensor(boxes_a, dtype=torch.float32)
    b = torch.as_tensor(boxes_b, dtype=torch.float32)
    assert not (a.is_cuda or b.is_cuda), "Only support CPU tensors"
    assert a.shape[1] == 7 and b.shape[1] == 7
    out = boxes_iou_bev(a.cuda().contiguous(), b.cuda().contiguous()).cpu()
    return out.numpy() if is_numpy else out


_NMS_MAX = 65536


def nms_sorted(boxes: torch.Tensor, thresh: float, rotated: bool = True):
    """NMS over boxes already sorted by descending score -> (keep positions (n) int64 on the device, count () int64 on the
    device).  No host synchronisation.  Limit: n <= 65536 boxes (vc_nms; the suppression-word workspace is n^2 / 8 bytes, 512 MB
    at the cap) -- the reference's nms_gpu has no such limit but is used with NMS_PRE_MAXSIZE 4096-9000 (VirConv-L.yaml); more
    boxes are handled by `nms_gpu` below by keeping the 65536 best-scored ones first (what pre_maxsize does)."""
    boxes = _need(boxes, torch.float32, "boxes")
    n = boxes.shape[0]
    keep = torch.empty((n,), dtype=torch.int64, device=boxes.device)
    num = torch.empty((), dtype=torch.int64, device=boxes.device)
    lib = _lib_handle()
    ws_bytes = lib.vc_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=boxes.device)
    _lib.check(lib.vc_nms(_ptr(boxes), n, float(thresh), 1 if rotated else 0, _ptr(keep), _ptr(num), _ptr(ws), ws_bytes, _stream()),
               "vc_nms")
    return keep, num


def nms_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, pre_maxsize=None, **kwargs):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    order = order[:_NMS_MAX]          # vc_nms cap (see nms_sorted)
    keep, num = nms_sorted(boxes[order].contiguous(), thresh, rotated=True)
    return order[keep[:int(num)]].contiguous(), None


def nms_normal_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, **kwargs):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1][:_NMS_MAX]
    keep, num = nms_sorted(boxes[order].contiguous(), thresh, rotated=False)
    return order[keep[:int(num)]].contiguous(), None
