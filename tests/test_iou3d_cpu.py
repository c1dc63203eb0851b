"""CPU tests of the IoU / NMS oracle (oracle/iou3d_ref.py): the numpy restatement is pinned against the REFERENCE ITSELF -- its
own iou3d_cpu.cpp compiled unmodified into oracle/_ref/libiou3d_ref.so -- on random KITTI-like boxes and on the edge cases the
geometry has (identical boxes, containment, touching edges, axis-aligned and 90-degree boxes, disjoint boxes)."""
import numpy as np
import pytest

from oracle import iou3d_ref as R


def kitti_like_boxes(seed, n, spread=30.0):
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = rng.uniform(0, spread, n)
    b[:, 1] = rng.uniform(-spread / 3, spread / 3, n)
    b[:, 2] = rng.uniform(-2, 0, n)
    b[:, 3] = rng.uniform(1, 5, n)
    b[:, 4] = rng.uniform(1, 3, n)
    b[:, 5] = rng.uniform(1, 2, n)
    b[:, 6] = rng.uniform(-3.2, 3.2, n)
    return b


def edge_case_boxes():
    base = np.array([[10, 0, -1, 4, 2, 1.5, 0.3]], np.float32)
    cases = [base[0].copy() for _ in range(10)]
    cases[1][3:5] = [2, 1]                      # contained in the first
    cases[2][0] += 4 * np.cos(0.3)              # shifted by exactly one length along the heading: touching edges
    cases[2][1] += 4 * np.sin(0.3)
    cases[3][6] = 0.3 + np.pi / 2               # same centre, rotated by 90 degrees
    cases[4][6] = 0.0                           # axis aligned
    cases[5][0] += 100                          # disjoint
    cases[6][6] = 0.3 + np.pi                   # rotated by 180 degrees: the same rectangle
    cases[7][0] += 0.005                        # inside the reference's 1e-2 corner margin
    cases[8][3:5] = [1e-3, 1e-3]                # degenerate size
    cases[9][6] = 0.3 + 1e-4                    # almost parallel edges
    return np.stack(cases).astype(np.float32)


needs_ref = pytest.mark.skipif(R.ref_lib() is None, reason="oracle/_ref/libiou3d_ref.so not built (no /root/reference here)")


@needs_ref
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_numpy_restatement_equals_the_compiled_reference_on_random_boxes(seed):
    a, b = kitti_like_boxes(seed, 150), kitti_like_boxes(100 + seed, 120)
    ref, mine = R.ref_boxes_iou_bev_cpu(a, b), R.boxes_iou_bev(a, b)
    assert (ref > 0).mean() > 0.02                 # the boxes do overlap
    np.testing.assert_allclose(mine, ref, rtol=0, atol=5e-6)
    dense = kitti_like_boxes(7 + seed, 100, spread=6.0)       # heavy overlap (an NMS input)
    np.testing.assert_allclose(R.boxes_iou_bev(dense, dense), R.ref_boxes_iou_bev_cpu(dense, dense), rtol=0, atol=5e-6)


@needs_ref
def test_numpy_restatement_equals_the_compiled_reference_on_edge_cases():
    e = edge_case_boxes()
    ref, mine = R.ref_boxes_iou_bev_cpu(e, e), R.boxes_iou_bev(e, e)
    assert np.isfinite(ref).all() and np.isfinite(mine).all()
    np.testing.assert_allclose(mine, ref, rtol=0, atol=5e-6)
    assert abs(ref[0, 0] - 1) < 1e-5 and abs(ref[0, 6] - 1) < 1e-5      # identical rectangle -> 1
    assert abs(ref[0, 1] - 0.25) < 1e-5                                   # contained: area ratio
    assert ref[0, 5] == 0                                                 # disjoint


def test_known_answers_without_the_reference():
    a = np.array([[0, 0, 0, 2, 2, 2, 0.0]], np.float32)
    b = np.array([[1, 0, 0, 2, 2, 2, 0.0], [0, 0, 1, 2, 2, 2, np.pi / 4], [5, 5, 0, 1, 1, 1, 0.2]], np.float32)
    iou = R.boxes_iou_bev(a, b)
    assert abs(iou[0, 0] - 2 / 6) < 1e-6                                   # half overlap: 2 / (4 + 4 - 2)
    octagon = 8 * (np.sqrt(2) - 1)                                        # square with its 45-degree copy: regular octagon
    assert abs(iou[0, 1] - octagon / (8 - octagon)) < 1e-5
    assert iou[0, 2] == 0
    iou3 = R.boxes_iou3d(a, b)
    assert abs(iou3[0, 0] - (2 * 2) / (8 + 8 - 4)) < 1e-6                  # full height overlap
    assert abs(iou3[0, 1] - octagon / (16 - octagon)) < 1e-5              # height overlap 1 of 2
    n = R.boxes_iou_normal(a, b)
    assert abs(n[0, 0] - 2 / 6) < 1e-6 and abs(n[0, 1] - 1) < 1e-6        # heading ignored


def test_nms_selection_loop():
    iou = np.array([[1, .8, .1, .0], [.8, 1, .9, .0], [.1, .9, 1, .6], [0, 0, .6, 1]], np.float32)
    assert R.nms_from_iou(iou, 0.5).tolist() == [0, 2]      # 1 is suppressed by 0, so it cannot suppress 2; 3 falls to 2
    assert R.nms_from_iou(iou, 0.95).tolist() == [0, 1, 2, 3]
    boxes = kitti_like_boxes(3, 60, spread=8.0)
    scores = np.random.default_rng(4).uniform(size=60).astype(np.float32)
    sel, _ = R.nms(boxes, scores, 0.1, pre_maxsize=40)
    assert len(sel) <= 40 and len(set(sel.tolist())) == len(sel)
    assert (np.diff(scores[sel]) <= 0).all()                # descending score
    iou_sel = R.boxes_iou_bev(boxes[sel], boxes[sel])
    assert (np.triu(iou_sel, 1) <= 0.1).all()               # survivors do not overlap beyond the threshold
