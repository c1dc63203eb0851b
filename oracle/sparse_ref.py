"""Sparse CPU restatement (oracle O2) of the spconv operators the VirConv backbone uses.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED against spconv itself (absent);
anchored on oracle/dense_ref.py.

Restates, from SURVEY.md Appendix A (spconv v2.1.22 semantics, [spconv-knowledge]) and the reference
call sites:
  * coordinate conventions / output shapes                       App-A.1   (spconv_backbone.py:552,639-644)
  * cross-correlation, weight layout (Cout, kz, ky, kx, Cin)     App-A.2   (detector3d_template.py:358-370)
  * active output set in ascending linear order                  App-A.3
  * dense pair tables pair_fwd[KV, N_out] / pair_bwd[KV, N_in]   App-A.4
  * duplicate-coordinate rule rep(c) = max row index             App-A.5   (spconv_backbone.py:217-222)
  * dense()                                                      App-A.6   (height_compression.py:29)

Algorithm class = the reference's CPU path: coordinate lookup -> per-kernel-offset rulebook ->
index_select gather -> mm -> index_add_ scatter (this is also what bench.py times as cpu_baseline).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch


def ntuple(v, n: int) -> Tuple[int, ...]:
    if isinstance(v, (list, tuple, np.ndarray)):
        assert len(v) == n, (v, n)
        return tuple(int(x) for x in v)
    return (int(v),) * n


def kernel_offsets(ksize: Sequence[int]) -> np.ndarray:
    """All kernel indices kappa in row-major (kz, ky, kx) order -> (KV, ndim) int64."""
    grids = np.meshgrid(*[np.arange(k) for k in ksize], indexing="ij")
    return np.stack([g.ravel() for g in grids], axis=1).astype(np.int64)


def conv_out_shape(in_shape, ksize, stride, padding, dilation) -> Tuple[int, ...]:
    """App-A.1: out = floor((in + 2*pad - dil*(k-1) - 1)/stride) + 1 per axis."""
    return tuple((int(i) + 2 * p - d * (k - 1) - 1) // s + 1
                 for i, k, s, p, d in zip(in_shape, ksize, stride, padding, dilation))


def linear_index(indices: np.ndarray, spatial_shape: Sequence[int]) -> np.ndarray:
    """L = ((b*S0 + c0)*S1 + c1)*S2 + c2, 64-bit (App-A.1; int32 overflows at bs>=3 on the T/S eval tensor)."""
    idx = indices.astype(np.int64)
    lin = idx[:, 0]
    for a, s in enumerate(spatial_shape):
        lin = lin * int(s) + idx[:, a + 1]
    return lin


class CoordLookup:
    """coordinate -> representative row, rep(c) = max{ i : coord_i = c }  (App-A.5)."""

    def __init__(self, indices: np.ndarray, spatial_shape: Sequence[int]):
        self.shape = tuple(int(s) for s in spatial_shape)
        keys = linear_index(indices, self.shape)
        order = np.lexsort((np.arange(keys.shape[0]), keys))  # by key, then row
        ks = keys[order]
        last = np.ones(ks.shape[0], dtype=bool)
        if ks.shape[0] > 1:
            last[:-1] = ks[1:] != ks[:-1]
        self.keys = ks[last]
        self.rows = order[last].astype(np.int64)

    def find(self, batch: np.ndarray, coords: np.ndarray) -> np.ndarray:
        """rows (or -1) for query coordinates; out-of-bounds queries give -1."""
        inb = np.ones(coords.shape[0], dtype=bool)
        lin = batch.astype(np.int64)
        for a, s in enumerate(self.shape):
            c = coords[:, a]
            inb &= (c >= 0) & (c < s)
            lin = lin * s + c
        res = np.full(coords.shape[0], -1, dtype=np.int64)
        if self.keys.shape[0] == 0:
            return res
        pos = np.searchsorted(self.keys, lin)
        pos_c = np.minimum(pos, self.keys.shape[0] - 1)
        hit = inb & (self.keys[pos_c] == lin)
        res[hit] = self.rows[pos_c[hit]]
        return res


def subm_rulebook(indices: np.ndarray, spatial_shape, ksize, dilation=1) -> np.ndarray:
    """Submanifold rulebook (App-A.3/A.4/A.5): pair_fwd[k, i] = rep(coord_i + (kappa_k - k//2)*dil) or -1.

    The centre tap of row i is row i itself (every row, duplicates included, keeps its own centre).
    Output rows == input rows, same order.  Returns int32 (KV, N).
    """
    ndim = indices.shape[1] - 1
    ksize = ntuple(ksize, ndim)
    dilation = ntuple(dilation, ndim)
    offs = kernel_offsets(ksize)
    n = indices.shape[0]
    lut = CoordLookup(indices, spatial_shape)
    coords = indices[:, 1:].astype(np.int64)
    batch = indices[:, 0].astype(np.int64)
    half = np.array([k // 2 for k in ksize], dtype=np.int64)
    dil = np.array(dilation, dtype=np.int64)
    pair = np.full((offs.shape[0], n), -1, dtype=np.int32)
    centre = int(np.ravel_multi_index(tuple(half), ksize))
    for k, kap in enumerate(offs):
        if k == centre:
            pair[k] = np.arange(n, dtype=np.int32)
            continue
        pair[k] = lut.find(batch, coords + (kap - half) * dil).astype(np.int32)
    return pair


def sparse_rulebook(indices: np.ndarray, spatial_shape, batch_size: int, ksize, stride, padding, dilation=1):
    """Regular (strided) sparse conv rulebook (App-A.1-A.4).

    p = q*stride - pad + kappa*dil.  Output rows = all in-bounds q with at least one active (p, kappa),
    in ASCENDING linear-index order (the spconv-CUDA order, App-A.3).
    Returns (out_indices (M, ndim+1) int32, out_shape, pair_fwd (KV, M) int32, pair_bwd (KV, N) int32).
    """
    ndim = indices.shape[1] - 1
    ksize, stride = ntuple(ksize, ndim), ntuple(stride, ndim)
    padding, dilation = ntuple(padding, ndim), ntuple(dilation, ndim)
    out_shape = conv_out_shape(spatial_shape, ksize, stride, padding, dilation)
    offs = kernel_offsets(ksize)
    n = indices.shape[0]
    coords = indices[:, 1:].astype(np.int64)
    batch = indices[:, 0].astype(np.int64)
    s = np.array(stride, dtype=np.int64)
    pad = np.array(padding, dtype=np.int64)
    dil = np.array(dilation, dtype=np.int64)
    osh = np.array(out_shape, dtype=np.int64)

    cand_lin = np.full((offs.shape[0], n), -1, dtype=np.int64)
    for k, kap in enumerate(offs):
        t = coords + pad - kap * dil
        ok = np.all(t % s == 0, axis=1)
        q = t // s
        ok &= np.all((q >= 0) & (q < osh), axis=1)
        lin = batch.copy()
        for a in range(ndim):
            lin = lin * osh[a] + q[:, a]
        cand_lin[k, ok] = lin[ok]
    out_lin = np.unique(cand_lin[cand_lin >= 0])  # sorted ascending
    m = out_lin.shape[0]
    out_indices = np.zeros((m, ndim + 1), dtype=np.int32)
    rem = out_lin.copy()
    for a in range(ndim - 1, -1, -1):
        out_indices[:, a + 1] = rem % osh[a]
        rem //= osh[a]
    out_indices[:, 0] = rem
    assert m == 0 or rem.max() < batch_size

    pair_bwd = np.full((offs.shape[0], n), -1, dtype=np.int32)
    pair_fwd = np.full((offs.shape[0], m), -1, dtype=np.int32)
    rows = np.arange(n, dtype=np.int64)
    for k in range(offs.shape[0]):
        ok = cand_lin[k] >= 0
        o = np.searchsorted(out_lin, cand_lin[k, ok])
        pair_bwd[k, ok] = o
        # duplicates in the input: highest row wins (App-A.5); rows are visited in ascending order
        np.maximum.at(pair_fwd[k], o, rows[ok].astype(np.int32))
    return out_indices, out_shape, pair_fwd, pair_bwd


def weight_per_offset(weight: torch.Tensor) -> torch.Tensor:
    """(Cout, *k, Cin) -> (KV, Cin, Cout)  (App-A.2: cross-correlation, no flip)."""
    cout, cin = weight.shape[0], weight.shape[-1]
    return weight.reshape(cout, -1, cin).permute(1, 2, 0)


def conv_forward(features: torch.Tensor, weight: torch.Tensor, pair_fwd: np.ndarray) -> torch.Tensor:
    """out[o, :] = sum_k  x[pair_fwd[k, o], :] @ W_k   (gather -> mm -> scatter-add; differentiable)."""
    wk = weight_per_offset(weight)
    kv, m = pair_fwd.shape
    out = features.new_zeros((m, weight.shape[0]))
    for k in range(kv):
        col = pair_fwd[k]
        o = np.nonzero(col >= 0)[0]
        if o.size == 0:
            continue
        i = torch.from_numpy(col[o].astype(np.int64))
        out = out.index_add(0, torch.from_numpy(o.astype(np.int64)), features.index_select(0, i) @ wk[k])
    return out


def conv_backward(features: torch.Tensor, weight: torch.Tensor, pair_fwd: np.ndarray, grad_out: torch.Tensor):
    """Explicit backward (App a15): dX[i] += dY[o] @ W_k^T ; dW_k = X[in_k]^T @ dY[out_k]."""
    wk = weight_per_offset(weight)
    kv, m = pair_fwd.shape
    dx = torch.zeros_like(features)
    dwk = torch.zeros_like(wk)
    for k in range(kv):
        col = pair_fwd[k]
        o = np.nonzero(col >= 0)[0]
        if o.size == 0:
            continue
        i = torch.from_numpy(col[o].astype(np.int64))
        ot = torch.from_numpy(o.astype(np.int64))
        gy = grad_out.index_select(0, ot)
        dx.index_add_(0, i, gy @ wk[k].t())
        dwk[k] = features.index_select(0, i).t() @ gy
    dweight = dwk.permute(2, 0, 1).reshape(weight.shape)
    return dx, dweight


def to_dense(features: torch.Tensor, indices: np.ndarray, spatial_shape, batch_size: int) -> torch.Tensor:
    """App-A.6: zeros(B, *spatial, C) <- rows (last write wins on duplicates) -> (B, C, *spatial)."""
    c = features.shape[1]
    shape = tuple(int(s) for s in spatial_shape)
    lin = linear_index(indices, shape)
    dense = features.new_zeros((batch_size * int(np.prod(shape)), c))
    # last write wins: keep, per cell, the highest row
    order = np.lexsort((np.arange(lin.shape[0]), lin))
    ls = lin[order]
    last = np.ones(ls.shape[0], dtype=bool)
    if ls.shape[0] > 1:
        last[:-1] = ls[1:] != ls[:-1]
    rows = torch.from_numpy(order[last].astype(np.int64))
    dense = dense.index_copy(0, torch.from_numpy(ls[last]), features.index_select(0, rows))
    dense = dense.reshape((batch_size,) + shape + (c,))
    perm = (0, len(shape) + 1) + tuple(range(1, len(shape) + 1))
    return dense.permute(*perm).contiguous()


def active_pairs(pair_fwd: np.ndarray) -> int:
    """P_l of SURVEY §8d: number of active (in, out) pairs."""
    return int((pair_fwd >= 0).sum())
