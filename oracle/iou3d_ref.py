"""TEST INFRASTRUCTURE (oracle): rotated BEV IoU / 3-D IoU / NMS, restated from the reference for the HIP kernels of
virconv_amd/csrc/nms_kernels.hip to be checked against.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; nothing under virconv_amd/ does.

Two oracles:
  * `ref_boxes_iou_bev_cpu` -- the REFERENCE ITSELF: /root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp compiled unmodified by
    oracle/build_ref.py into oracle/_ref/libiou3d_ref.so (boxes_iou_bev_cpu, iou3d_cpu.cpp:222-251).  Parity of the IoU is pinned
    against this one.
  * a float32 numpy restatement, vectorised over box pairs, of box_overlap / iou_bev (iou3d_cpu.cpp:137-220 =
    iou3d_nms_kernel.cu:127-233), iou_normal (iou3d_nms_kernel.cu:321-331), the 3-D IoU composition
    (iou3d_nms_utils.py:67-99) and the greedy selection loop of nms_gpu (iou3d_nms.cpp:98-150, iou3d_nms_utils.py:102-135);
    it is itself checked against the compiled reference in tests/test_iou3d_cpu.py.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

F = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = None


def ref_lib():
    """ctypes handle of the compiled reference (None when it was never built and /root/reference is absent)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libiou3d_ref.so")
        if not os.path.exists(path):
            from . import build_ref
            if build_ref.build(verbose=False) is None:
                return None
        import torch  # noqa: F401  (libtorch must be in the process before the reference library resolves its symbols)
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:   # built against another libtorch: treat as "not built" (the numpy restatement still checks the kernels)
            import warnings
            warnings.warn(f"oracle/_ref/libiou3d_ref.so does not load here ({e}); rebuild with python -m oracle.build_ref")
            return None
        lib.ref_boxes_iou_bev_cpu.restype = ctypes.c_int
        lib.ref_boxes_iou_bev_cpu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _REF = lib
    return _REF


def ref_boxes_iou_bev_cpu(boxes_a: np.ndarray, boxes_b: np.ndarray) -> np.ndarray:
    lib = ref_lib()
    assert lib is not None, "oracle/_ref/libiou3d_ref.so is missing (python -m oracle.build_ref in the build container)"
    a = np.ascontiguousarray(boxes_a[:, :7], dtype=np.float32)
    b = np.ascontiguousarray(boxes_b[:, :7], dtype=np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    if a.shape[0] and b.shape[0]:
        lib.ref_boxes_iou_bev_cpu(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data)
    return out


def _corners(box):
    """(P, 7) -> (P, 5, 2) oriented corners, last = first (iou3d_cpu.cpp:143-172)."""
    hx, hy = box[:, 3] / F(2), box[:, 4] / F(2)
    cs, sn = np.cos(box[:, 6]).astype(F), np.sin(box[:, 6]).astype(F)
    out = np.zeros((box.shape[0], 5, 2), F)
    for k, (sx, sy) in enumerate(((-1, -1), (1, -1), (1, 1), (-1, 1))):
        px, py = box[:, 0] + F(sx) * hx, box[:, 1] + F(sy) * hy
        out[:, k, 0] = (px - box[:, 0]) * cs + (py - box[:, 1]) * (-sn) + box[:, 0]
        out[:, k, 1] = (px - box[:, 0]) * sn + (py - box[:, 1]) * cs + box[:, 1]
    out[:, 4] = out[:, 0]
    return out


def _cross3(p1, p2, p0):
    return (p1[:, 0] - p0[:, 0]) * (p2[:, 1] - p0[:, 1]) - (p2[:, 0] - p0[:, 0]) * (p1[:, 1] - p0[:, 1])


def _intersection(p1, p0, q1, q0):
    """iou3d_cpu.cpp:86-117 -> (valid (P,), point (P, 2))"""
    mn, mx = np.minimum, np.maximum
    meet = ((mn(p0[:, 0], p1[:, 0]) <= mx(q0[:, 0], q1[:, 0])) & (mn(q0[:, 0], q1[:, 0]) <= mx(p0[:, 0], p1[:, 0])) &
            (mn(p0[:, 1], p1[:, 1]) <= mx(q0[:, 1], q1[:, 1])) & (mn(q0[:, 1], q1[:, 1]) <= mx(p0[:, 1], p1[:, 1])))
    s1, s2, s3, s4 = _cross3(q0, p1, p0), _cross3(p1, q1, p0), _cross3(p0, q1, q0), _cross3(q1, p1, q0)
    valid = meet & (s1 * s2 > 0) & (s3 * s4 > 0)
    s5 = _cross3(q1, p1, p0)
    with np.errstate(all="ignore"):
        x1 = (s5 * q0[:, 0] - s1 * q1[:, 0]) / (s5 - s1)
        y1 = (s5 * q0[:, 1] - s1 * q1[:, 1]) / (s5 - s1)
        a0, b0, c0 = p0[:, 1] - p1[:, 1], p1[:, 0] - p0[:, 0], p0[:, 0] * p1[:, 1] - p1[:, 0] * p0[:, 1]
        a1, b1, c1 = q0[:, 1] - q1[:, 1], q1[:, 0] - q0[:, 0], q0[:, 0] * q1[:, 1] - q1[:, 0] * q0[:, 1]
        d = a0 * b1 - a1 * b0
        x2, y2 = (b0 * c1 - b1 * c0) / d, (a1 * c0 - a0 * c1) / d
    first = np.abs(s5 - s1) > F(1e-8)
    return valid, np.stack([np.where(first, x1, x2), np.where(first, y1, y2)], 1).astype(F)


def _in_box(box, p):
    """iou3d_cpu.cpp:74-84 (MARGIN 1e-2)"""
    c, s = np.cos(-box[:, 6]).astype(F), np.sin(-box[:, 6]).astype(F)
    dx, dy = p[:, 0] - box[:, 0], p[:, 1] - box[:, 1]
    rx, ry = dx * c + dy * (-s), dx * s + dy * c
    return (np.abs(rx) < box[:, 3] / F(2) + F(1e-2)) & (np.abs(ry) < box[:, 4] / F(2) + F(1e-2))


def overlap_pairs(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """box_overlap (iou3d_cpu.cpp:137-211) for P pairs: a, b (P, 7) float32 -> (P,) overlap areas."""
    a, b = a.astype(F), b.astype(F)
    P = a.shape[0]
    ca, cb = _corners(a), _corners(b)
    pts = np.zeros((P, 24, 2), F)
    ok = np.zeros((P, 24), bool)
    s = 0
    for i in range(4):
        for j in range(4):
            ok[:, s], pts[:, s] = _intersection(ca[:, i + 1], ca[:, i], cb[:, j + 1], cb[:, j])
            s += 1
    for k in range(4):
        ok[:, s], pts[:, s] = _in_box(a, cb[:, k]), cb[:, k]
        s += 1
        ok[:, s], pts[:, s] = _in_box(b, ca[:, k]), ca[:, k]
        s += 1
    cnt = ok.sum(1)
    cx, cy = np.zeros(P, F), np.zeros(P, F)
    for s in range(24):  # the reference accumulates in this order
        cx = np.where(ok[:, s], cx + pts[:, s, 0], cx)
        cy = np.where(ok[:, s], cy + pts[:, s, 1], cy)
    with np.errstate(all="ignore"):
        cx, cy = cx / cnt.astype(F), cy / cnt.astype(F)
        ang = np.arctan2(pts[:, :, 1] - cy[:, None], pts[:, :, 0] - cx[:, None]).astype(F)
    ang = np.where(ok, ang, np.inf)
    order = np.argsort(ang, axis=1, kind="stable")  # bubble sort with a strict comparator = stable ascending sort
    sp = np.take_along_axis(pts, order[:, :, None], 1)
    area = np.zeros(P, F)
    with np.errstate(all="ignore"):  # slots past cnt hold the sort's +inf keys: masked out below
        for k in range(23):
            term = ((sp[:, k, 0] - sp[:, 0, 0]) * (sp[:, k + 1, 1] - sp[:, 0, 1]) -
                    (sp[:, k, 1] - sp[:, 0, 1]) * (sp[:, k + 1, 0] - sp[:, 0, 0]))
            area = np.where(k < cnt - 1, area + term, area)
    return (np.abs(area) / F(2)).astype(F)


def _all_pairs(boxes_a, boxes_b):
    a = np.ascontiguousarray(boxes_a[:, :7], dtype=F)
    b = np.ascontiguousarray(boxes_b[:, :7], dtype=F)
    n, m = a.shape[0], b.shape[0]
    return np.repeat(a, m, 0), np.tile(b, (n, 1)), n, m


def boxes_overlap_bev(boxes_a, boxes_b) -> np.ndarray:
    pa, pb, n, m = _all_pairs(boxes_a, boxes_b)
    return overlap_pairs(pa, pb).reshape(n, m) if n and m else np.zeros((n, m), F)


def boxes_iou_bev(boxes_a, boxes_b) -> np.ndarray:
    """iou_bev (iou3d_cpu.cpp:213-220)"""
    pa, pb, n, m = _all_pairs(boxes_a, boxes_b)
    if not (n and m):
        return np.zeros((n, m), F)
    ov = overlap_pairs(pa, pb)
    return (ov / np.maximum(pa[:, 3] * pa[:, 4] + pb[:, 3] * pb[:, 4] - ov, F(1e-8))).reshape(n, m).astype(F)


def boxes_iou3d(boxes_a, boxes_b) -> np.ndarray:
    """iou3d_nms_utils.py:67-99"""
    pa, pb, n, m = _all_pairs(boxes_a, boxes_b)
    if not (n and m):
        return np.zeros((n, m), F)
    ov = overlap_pairs(pa, pb)
    hmax = np.minimum(pa[:, 2] + pa[:, 5] / F(2), pb[:, 2] + pb[:, 5] / F(2))
    hmin = np.maximum(pa[:, 2] - pa[:, 5] / F(2), pb[:, 2] - pb[:, 5] / F(2))
    ov3 = ov * np.maximum(hmax - hmin, F(0))
    va, vb = pa[:, 3] * pa[:, 4] * pa[:, 5], pb[:, 3] * pb[:, 4] * pb[:, 5]
    return (ov3 / np.maximum(va + vb - ov3, F(1e-6))).reshape(n, m).astype(F)


def boxes_iou_normal(boxes_a, boxes_b) -> np.ndarray:
    """iou_normal (iou3d_nms_kernel.cu:321-331): axis-aligned, heading ignored"""
    pa, pb, n, m = _all_pairs(boxes_a, boxes_b)
    if not (n and m):
        return np.zeros((n, m), F)
    left = np.maximum(pa[:, 0] - pa[:, 3] / F(2), pb[:, 0] - pb[:, 3] / F(2))
    right = np.minimum(pa[:, 0] + pa[:, 3] / F(2), pb[:, 0] + pb[:, 3] / F(2))
    top = np.maximum(pa[:, 1] - pa[:, 4] / F(2), pb[:, 1] - pb[:, 4] / F(2))
    bottom = np.minimum(pa[:, 1] + pa[:, 4] / F(2), pb[:, 1] + pb[:, 4] / F(2))
    inter = np.maximum(right - left, F(0)) * np.maximum(bottom - top, F(0))
    return (inter / np.maximum(pa[:, 3] * pa[:, 4] + pb[:, 3] * pb[:, 4] - inter, F(1e-8))).reshape(n, m).astype(F)


def nms_from_iou(iou: np.ndarray, thresh: float) -> np.ndarray:
    """The selection loop of nms_gpu on boxes ALREADY sorted by descending score (iou3d_nms.cpp:125-150): box i is kept iff no
    kept box j < i has iou[j, i] > thresh."""
    n = iou.shape[0]
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > F(thresh)
    return np.asarray(keep, np.int64)


def nms(boxes: np.ndarray, scores: np.ndarray, thresh: float, pre_maxsize=None, rotated: bool = True, iou_fn=None):
    """nms_gpu / nms_normal_gpu (iou3d_nms_utils.py:102-135) -> selected indices into `boxes` (descending score)."""
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order]
    iou = (iou_fn or (boxes_iou_bev if rotated else boxes_iou_normal))(b, b)
    return order[nms_from_iou(iou, thresh)], iou
