"""Dense oracle O1 / O3: an INDEPENDENT restatement sharing no code with oracle/sparse_ref.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A sparse convolution equals ``torch.nn.functional.conv{2,3}d(dense_x, W.permute(Cout, Cin, *k), stride, pad)``
read at the active output sites; the active set of a strided conv is the support of
``conv(ones_mask, ones_kernel) > 0`` in row-major (= ascending linear-index) order (App-A.3).
Only usable on small grids and duplicate-free inputs.  Backward oracle O3 = torch autograd through this.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _nt(v, n):
    return tuple(int(x) for x in v) if isinstance(v, (list, tuple, np.ndarray)) else (int(v),) * n


def densify(features: torch.Tensor, indices: np.ndarray, spatial_shape, batch_size: int) -> torch.Tensor:
    """(N, C) rows at (b, *coord) -> (B, C, *spatial); requires unique coordinates."""
    shape = tuple(int(s) for s in spatial_shape)
    dense = features.new_zeros((batch_size, features.shape[1]) + shape)
    return _put(dense, torch.from_numpy(indices.astype(np.int64)), features)


def _put(dense, idx, features):
    nd = idx.shape[1] - 1
    dense = dense.movedim(1, -1).contiguous()  # (B, *spatial, C)
    flat = dense.reshape(-1, dense.shape[-1])
    lin = idx[:, 0]
    for a in range(nd):
        lin = lin * dense.shape[a + 1] + idx[:, a + 1]
    assert torch.unique(lin).numel() == lin.numel(), "dense oracle needs duplicate-free coordinates"
    flat = flat.index_copy(0, lin, features)
    return flat.reshape(dense.shape).movedim(-1, 1)


def _conv(x, w, stride, padding, dilation):
    nd = x.dim() - 2
    fn = F.conv3d if nd == 3 else F.conv2d
    perm = (0, nd + 1) + tuple(range(1, nd + 1))
    return fn(x, w.permute(*perm), None, stride, padding, dilation)


def subm_conv(features, indices, spatial_shape, batch_size, weight, dilation=1):
    """SubM conv: out rows == in rows; effective padding (k//2)*dil, stride 1 (App-A.1)."""
    nd = indices.shape[1] - 1
    ks = tuple(weight.shape[1:-1])
    dil = _nt(dilation, nd)
    pad = tuple((k // 2) * d for k, d in zip(ks, dil))
    y = _conv(densify(features, indices, spatial_shape, batch_size), weight, (1,) * nd, pad, dil)
    idx = torch.from_numpy(indices.astype(np.int64))
    sl = (idx[:, 0], slice(None)) + tuple(idx[:, a + 1] for a in range(nd))
    return y[sl]


def sparse_conv(features, indices, spatial_shape, batch_size, weight, stride, padding, dilation=1):
    """Regular sparse conv -> (out_features (M, Cout), out_indices (M, nd+1) int32 ascending, out_shape)."""
    nd = indices.shape[1] - 1
    ks = tuple(weight.shape[1:-1])
    stride, padding, dil = _nt(stride, nd), _nt(padding, nd), _nt(dilation, nd)
    x = densify(features, indices, spatial_shape, batch_size)
    y = _conv(x, weight, stride, padding, dil)
    mask = densify(torch.ones((features.shape[0], 1), dtype=features.dtype), indices, spatial_shape, batch_size)
    ones = torch.ones((1,) + ks + (1,), dtype=features.dtype)
    act = _conv(mask, ones, stride, padding, dil)[:, 0] > 0.5
    out_idx = torch.nonzero(act)  # row-major == ascending linear index
    sl = (out_idx[:, 0], slice(None)) + tuple(out_idx[:, a + 1] for a in range(nd))
    return y[sl], out_idx.numpy().astype(np.int32), tuple(y.shape[2:])
