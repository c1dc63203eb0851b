"""Harness that imports the REFERENCE's own composition layer (pcdet/models/backbones_3d/spconv_backbone.py,
unmodified, from /root/reference) on top of the virconv_amd spconv facade.

Build-container only: /root/reference does not exist on the GPU box, so nothing that runs there imports this module
without first checking ``available()``.  Used by tests/golden/make_golden.py to generate the committed fixtures and by
tests/test_reference_composition.py (skipped when the reference tree is absent).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"

_STUBS = [
    "pcdet.ops.iou3d_nms.iou3d_nms_cuda",
    "pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda",
    "pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda",
    "pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda",
    "skimage", "skimage.io", "skimage.transform", "prefetch_generator", "cv2", "easydict", "tensorboardX", "numba",
    "SharedArray",
]


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pcdet"))


def _install_stubs():
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            if name.split(".")[0] not in ("pcdet",):
                importlib.import_module(name)
                continue
        except Exception:
            pass
        m = types.ModuleType(name)
        sys.modules[name] = m
        parent, _, child = name.rpartition(".")
        if parent and parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    nb = sys.modules["numba"]
    if not hasattr(nb, "jit"):
        def _passthrough(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        nb.jit = nb.njit = _passthrough
        nb.cuda = types.SimpleNamespace(jit=_passthrough)
        nb.float32 = nb.int32 = None
    ed = sys.modules["easydict"]
    if not hasattr(ed, "EasyDict"):
        class EasyDict(dict):
            def __init__(self, d=None, **kw):
                super().__init__()
                for k, v in dict(d or {}, **kw).items():
                    self[k] = v
            def __setitem__(self, k, v):
                if isinstance(v, dict) and not isinstance(v, EasyDict):
                    v = EasyDict(v)
                super().__setitem__(k, v)
            __setattr__ = __setitem__
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError as e:
                    raise AttributeError(k) from e
        ed.EasyDict = EasyDict
    pg = sys.modules["prefetch_generator"]
    if not hasattr(pg, "BackgroundGenerator"):
        pg.BackgroundGenerator = object


def import_reference_backbone():
    """-> the reference module pcdet.models.backbones_3d.spconv_backbone, bound to the virconv_amd facade."""
    assert available(), "reference tree not present"
    import virconv_amd.spconv as facade
    facade.install(force=True)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module("pcdet.models.backbones_3d.spconv_backbone")


class Calib:
    """Minimal stand-in for pcdet.utils.calibration_kitti.Calibration built from a dict (it accepts dicts itself)."""


def make_reference_calib(calib_dict):
    from pcdet.utils.calibration_kitti import Calibration
    return Calibration(dict(calib_dict))


# ------------------------------------------------------------------------------------------------ RoI grid pooling
def import_reference_voxel_pool():
    """-> (voxel_pool_modules, spconv_utils, common_utils) of the REFERENCE, unmodified, with its compiled extension
    `pointnet2_stack_cuda` replaced by the CPU oracle (oracle/pooling_ref.py) and torch.cuda.{Int,Float}Tensor mapped to
    CPU constructors (the wrappers allocate their outputs with them: voxel_query_utils.py:33, pointnet2_utils.py:77,97)."""
    assert available(), "reference tree not present"
    import numpy as np
    import torch
    from oracle import pooling_ref
    import_reference_backbone()  # facade + stubs + sys.path
    ext = sys.modules["pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda"]

    def voxel_query_wrapper(M, Z, Y, X, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                            point_indices, idx):
        out, empty = pooling_ref.voxel_query((z_range, y_range, x_range), radius, nsample, xyz.numpy(), new_xyz.numpy(),
                                             new_coords.numpy(), point_indices.numpy())
        out = out.copy()
        out[empty, 0] = -1  # raw kernel output: `if (cnt == 0) idx[0] = -1` on a zero-initialised row
        idx.copy_(torch.from_numpy(out))
        return 1

    def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, output):
        output.copy_(torch.from_numpy(pooling_ref.group_points(features.detach().numpy(), features_batch_cnt.numpy(),
                                                               idx.numpy(), idx_batch_cnt.numpy())))
        return 1

    def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
        grad_features.copy_(torch.from_numpy(pooling_ref.group_points_grad(
            grad_out.numpy(), idx.numpy(), idx_batch_cnt.numpy(), features_batch_cnt.numpy(), N)))
        return 1

    ext.voxel_query_wrapper = voxel_query_wrapper
    ext.group_points_wrapper = group_points_wrapper
    ext.group_points_grad_wrapper = group_points_grad_wrapper
    torch.cuda.IntTensor = lambda *shape: torch.zeros(*shape, dtype=torch.int32)
    torch.cuda.FloatTensor = lambda *shape: torch.zeros(*shape, dtype=torch.float32)
    vpm = importlib.import_module("pcdet.ops.pointnet2.pointnet2_stack.voxel_pool_modules")
    su = importlib.import_module("pcdet.utils.spconv_utils")
    cu = importlib.import_module("pcdet.utils.common_utils")
    return vpm, su, cu
