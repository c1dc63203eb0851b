"""GPU parity tests of the rotated IoU / NMS kernels (csrc/nms_kernels.hip through virconv_amd/iou3d_nms.py, the mirror of the
reference's iou3d_nms_utils.py): IoU matrices within 1e-5 of the compiled reference (oracle/_ref) and of the numpy oracle, NMS
keep lists identical to the reference's selection loop run on the oracle IoU matrix."""
import numpy as np
import pytest
import torch

from oracle import iou3d_ref as R
from test_iou3d_cpu import edge_case_boxes, kitti_like_boxes

pytestmark = pytest.mark.gpu

ATOL = 1e-5


@pytest.fixture
def nms_ops(hip_backend):
    from virconv_amd import iou3d_nms
    return iou3d_nms


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n_a,n_b", [(150, 120), (1, 1), (65, 3), (3, 257), (0, 5), (5, 0)])
def test_iou_matrices_vs_oracle_and_compiled_reference(nms_ops, n_a, n_b):
    a, b = kitti_like_boxes(1, n_a), kitti_like_boxes(2, n_b)
    got = nms_ops.boxes_iou_bev(_t(a), _t(b)).cpu().numpy()
    assert got.shape == (n_a, n_b)
    np.testing.assert_allclose(got, R.boxes_iou_bev(a, b), rtol=0, atol=ATOL)
    if R.ref_lib() is not None:
        np.testing.assert_allclose(got, R.ref_boxes_iou_bev_cpu(a, b), rtol=0, atol=ATOL)
    np.testing.assert_allclose(nms_ops.boxes_overlap_bev(_t(a), _t(b)).cpu().numpy(), R.boxes_overlap_bev(a, b), rtol=0, atol=5e-5)
    np.testing.assert_allclose(nms_ops.boxes_iou3d_gpu(_t(a), _t(b)).cpu().numpy(), R.boxes_iou3d(a, b), rtol=0, atol=ATOL)


def test_iou_edge_cases(nms_ops):
    e = edge_case_boxes()
    got = nms_ops.boxes_iou_bev(_t(e), _t(e)).cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, R.boxes_iou_bev(e, e), rtol=0, atol=ATOL)
    if R.ref_lib() is not None:
        np.testing.assert_allclose(got, R.ref_boxes_iou_bev_cpu(e, e), rtol=0, atol=ATOL)
    dense = kitti_like_boxes(9, 200, spread=6.0)
    np.testing.assert_allclose(nms_ops.boxes_iou_bev(_t(dense), _t(dense)).cpu().numpy(), R.boxes_iou_bev(dense, dense), rtol=0, atol=ATOL)
    np.testing.assert_allclose(nms_ops.boxes_bev_iou_cpu(e, e), got, rtol=0, atol=0)   # the CPU-facing entry point, numpy in/out


def _scene(seed, n, spread, need_gap):
    """boxes + scores (+ the reference IoU matrices of the score-sorted boxes).  need_gap: search for a scene whose pairwise IoUs
    all stay 1e-4 away from the thresholds used below, so that the selection does not depend on the last bits of an IoU."""
    for s in range(seed, seed + 50):
        boxes = kitti_like_boxes(s, n, spread=spread)
        scores = np.random.default_rng(s + 1000).permutation(n).astype(np.float32) / n     # distinct scores: unambiguous order
        order = np.argsort(-scores.astype(np.float64), kind="stable")
        ref_iou = R.ref_boxes_iou_bev_cpu if R.ref_lib() is not None else R.boxes_iou_bev
        iou_r = ref_iou(boxes[order], boxes[order])
        iou_n = R.boxes_iou_normal(boxes[order], boxes[order])
        if not need_gap or all(np.abs(m - t).min() > 1e-4 for m in (iou_r, iou_n) for t in (0.1, 0.7)):
            return boxes, scores, order, iou_r, iou_n
    raise AssertionError("no unambiguous scene found")


@pytest.mark.parametrize("n,spread", [(1, 10.0), (63, 10.0), (64, 10.0), (65, 8.0), (1000, 40.0), (4100, 90.0)])
@pytest.mark.parametrize("thresh", [0.1, 0.7])
def test_nms_equals_the_reference_selection_loop(nms_ops, n, spread, thresh):
    """Up to 65 boxes: a scene with every IoU at least 1e-4 away from the threshold, expected list = the reference's selection
    loop over the COMPILED REFERENCE's IoU matrix.  Thousands of boxes (millions of pairs: some IoU always sits within 1e-5 of
    any threshold): the selection loop over the IoU matrix of the HIP pair kernel -- the same device function the NMS mask kernel
    evaluates, itself within 1e-5 of the reference (checked here too)."""
    small = n <= 65
    boxes, scores, order, iou_r, iou_n = _scene(n, n, spread, need_gap=small)
    if not small:
        sb = _t(boxes[order])
        hip_r = nms_ops.boxes_iou_bev(sb, sb).cpu().numpy()
        # 1e-5 holds for all but a few of 17 M pairs; the worst (1.5e-5, a box against itself: eight near-coincident polygon
        # vertices whose angular order hangs on the last bit of atan2f) sets the declared bound for matrices of this size
        np.testing.assert_allclose(hip_r, iou_r, rtol=0, atol=5e-5)
        assert (np.abs(hip_r - iou_r) > ATOL).mean() < 1e-6
        iou_r = hip_r
        flips = (hip_r > thresh) != (R.boxes_iou_normal(boxes[order], boxes[order]) > thresh)   # (only to show the test has teeth)
        assert flips.any()
    sel, none = nms_ops.nms_gpu(_t(boxes), _t(scores), thresh)
    assert none is None and sel.dtype == torch.int64
    assert sel.cpu().numpy().tolist() == order[R.nms_from_iou(iou_r, thresh)].tolist()
    sel_n, _ = nms_ops.nms_normal_gpu(_t(boxes), _t(scores), thresh)
    exp_n = order[R.nms_from_iou(iou_n, thresh)]
    if small:
        assert sel_n.cpu().numpy().tolist() == exp_n.tolist()
    else:   # the axis-aligned IoU is a handful of float ops; allow the near-threshold pairs of a float32 numpy vs device evaluation
        near = np.abs(iou_n - thresh) < 1e-6
        assert sel_n.cpu().numpy().tolist() == exp_n.tolist() or near.any()
    if n >= 64:
        pre = n // 2
        sel_p, _ = nms_ops.nms_gpu(_t(boxes), _t(scores), thresh, pre_maxsize=pre)
        assert sel_p.cpu().numpy().tolist() == order[:pre][R.nms_from_iou(iou_r[:pre, :pre], thresh)].tolist()


def test_nms_empty_and_device_side_count(nms_ops):
    keep, num = nms_ops.nms_sorted(torch.zeros((0, 7), device="cuda"), 0.5)
    assert keep.numel() == 0 and int(num) == 0
    boxes = kitti_like_boxes(5, 300, spread=10.0)
    keep, num = nms_ops.nms_sorted(_t(boxes), 0.3)
    assert num.is_cuda and num.dtype == torch.int64 and 0 < int(num) <= 300
    k = keep[:int(num)].cpu().numpy()
    assert (np.diff(k) > 0).all()                       # ascending positions = descending score
    again, num2 = nms_ops.nms_sorted(_t(boxes), 0.3)
    assert int(num2) == int(num) and torch.equal(again[:int(num)], keep[:int(num)])
