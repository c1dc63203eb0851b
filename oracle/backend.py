"""OracleBackend: exposes the CPU oracle through the product's backend interface (virconv_amd/backend_hip.py).

TEST INFRASTRUCTURE ONLY.  Injected by tests with ``virconv_amd.ops.use_backend(OracleBackend())`` to (a) run the
reference's own composition layer on CPU and produce golden fixtures, and (b) exercise host-side logic (facade,
rulebook caching, autograd wiring, DDP) without a GPU.  The product never instantiates this class.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import geometry, pooling_ref, sparse_ref


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


class OracleBackend:
    name = "oracle"

    # ------------------------------------------------------------------ rulebooks
    def subm_rulebook(self, indices, spatial_shape, ksize, dilation, want_rep: bool):
        idx = _np(indices)
        pair = sparse_ref.subm_rulebook(idx, spatial_shape, ksize, dilation)
        rep = None
        if want_rep:
            lut = sparse_ref.CoordLookup(idx, spatial_shape)
            rep = torch.from_numpy(lut.find(idx[:, 0].astype(np.int64), idx[:, 1:].astype(np.int64)).astype(np.int32))
        return torch.from_numpy(pair), rep

    def sparse_rulebook(self, indices, spatial_shape, batch_size, ksize, stride, padding, dilation):
        oi, oshape, pf, pb = sparse_ref.sparse_rulebook(_np(indices), spatial_shape, batch_size, ksize, stride, padding,
                                                        dilation)
        return torch.from_numpy(oi), oshape, torch.from_numpy(pf), torch.from_numpy(pb)

    def sparse_rulebook_chain(self, indices, spatial_shape, batch_size, geoms):
        """The contract of HipBackend.sparse_rulebook_chain, level by level (so that the geometry plan's "every strided rulebook is
        ready" path runs in the CPU tests too): [(out_indices, out_shape, pair_fwd, pair_bwd, in_indices), ...]."""
        res, cur, shape = [], indices, tuple(int(v) for v in spatial_shape)
        for ksize, stride, padding, dilation in geoms:
            oi, oshape, pf, pb = self.sparse_rulebook(cur, shape, batch_size, ksize, stride, padding, dilation)
            res.append((oi, oshape, pf, pb, cur))
            cur, shape = oi, tuple(int(v) for v in oshape)
        return res

    # ------------------------------------------------------------------ convolution
    def row_order(self, tbl, rep=None, centre=-1, window=1024):
        return None  # scheduling hint of the HIP path only; results never depend on it

    @staticmethod
    def _operands(operand, cin, cout, *tensors):
        """Emulation of the product's reduced-precision mode: MFMA operands rounded (RNE) to fp16 / bf16, exact products,
        wide accumulation -- only where both channel counts are >= 16 (include/virconv_hip.h, vc_operand)."""
        ts = [t.detach() for t in tensors]
        if operand == "f32" or cin < 16 or cout < 16:
            return ts
        dt = {"f16": torch.float16, "bf16": torch.bfloat16}[operand]
        return [t.to(dt).to(t.dtype) for t in ts]

    def conv_forward(self, x, weight, pair_fwd, order=None, operand="f32", sorted_rows=False):
        x, weight = self._operands(operand, weight.shape[-1], weight.shape[0], x, weight)
        return sparse_ref.conv_forward(x, weight, _np(pair_fwd))

    def conv_epilogue_supported(self, n_in, cin, cout, kv, operand="f32"):
        return False  # fused epilogues are a product optimisation; the oracle always takes the plain path

    def conv_backward_input(self, dy, weight, tbl, n_in, mirror, centre=-1, rep=None, order=None, operand="f32",
                            group_ws=None, sorted_rows=False, grp_plan=None, grp=None):
        """dX of the three conv flavours, in the product's calling convention (include/virconv_hip.h).

        SubM (mirror=True, tbl = pair_fwd): the EXACT transpose of the forward gather,
            dx[j] = sum_{k,o : pair_fwd[k,o] = j} dy[o] @ W_k^T,
        computed by scatter-add straight from pair_fwd -- independent of the product's mirrored-table / group-sum
        formulation, so it checks that formulation (duplicate coordinates included; `centre`/`rep` are ignored).
        Strided conv (tbl = pair_bwd) and inverse conv (tbl = pair_fwd of the forward conv): gather form
            dx[i] = sum_k dy[tbl[k,i]] @ W_k^T.
        """
        t = _np(tbl)
        dy, w = self._operands(operand, weight.shape[-1], weight.shape[0], dy, weight)
        if mirror:
            dx, _ = sparse_ref.conv_backward(dy.new_zeros((n_in, w.shape[-1])), w, t, dy)
            return dx
        wk = sparse_ref.weight_per_offset(w)  # (KV, Cin, Cout)
        dx = dy.new_zeros((n_in, w.shape[-1]))
        for k in range(t.shape[0]):
            rows = np.nonzero(t[k] >= 0)[0]
            if rows.size == 0:
                continue
            g = dy.index_select(0, torch.from_numpy(t[k, rows].astype(np.int64)))
            dx.index_add_(0, torch.from_numpy(rows.astype(np.int64)), g @ wk[k].t())
        return dx

    def conv_backward_weight(self, x, dy, pair_fwd, weight_shape, stream=None, keep_alive=None, operand="f32", rep=None, centre=-1,
                             dy_grp=None):
        w0 = x.new_zeros(tuple(weight_shape))
        x, dy = self._operands(operand, int(weight_shape[-1]), int(weight_shape[0]), x, dy)
        _, dw = sparse_ref.conv_backward(x, w0, _np(pair_fwd), dy)
        return dw

    # ------------------------------------------------------------------ projection / discard / dense
    def project_uv(self, indices, calib, trans, batch_size, stride, want_depth=False):
        cal = _np(calib)
        calibs = [{"Tr_velo2cam": c[0:12].reshape(3, 4), "R0": c[12:21].reshape(3, 3), "P2": c[21:33].reshape(3, 4)}
                  for c in cal]
        tp = None if trans is None else _np(trans)
        uv, depth = geometry.index2uv(_np(indices), batch_size, calibs, stride, tp)
        return torch.from_numpy(uv), (torch.from_numpy(depth) if want_depth else None)

    def gather_rows(self, features, indices, keep):
        k = keep.long()
        return (None if features is None else features.detach()[k]), (None if indices is None else indices[k])

    def scatter_rows(self, grad_out, keep, n_in):
        g = grad_out.new_zeros((n_in, grad_out.shape[1]))
        g[keep.long()] = grad_out
        return g

    def to_dense(self, features, indices, spatial_shape, batch_size, pad=(0, 0)):
        d = sparse_ref.to_dense(features.detach(), _np(indices), spatial_shape, batch_size)
        if pad[0] or pad[1]:    # zero border around the last two axes (what nn.ZeroPad2d adds in front of the first BEV conv)
            d = torch.nn.functional.pad(d, (pad[1], pad[1], pad[0], pad[0]))
        return d

    def from_dense(self, dense, indices, spatial_shape, batch_size, pad=(0, 0)):
        idx = indices.long()
        nd = idx.shape[1] - 1
        off = [0] * (nd - 2) + [int(pad[0]), int(pad[1])]
        sl = (idx[:, 0], slice(None)) + tuple(idx[:, a + 1] + off[a] for a in range(nd))
        return dense[sl].contiguous()

    # ------------------------------------------------------------------ RoI grid pooling
    def voxel_index_build(self, indices, batch_size, spatial_shape):
        return torch.from_numpy(pooling_ref.voxel2pinds(_np(indices), int(batch_size), spatial_shape))  # dense volume

    def voxel_query(self, ws, n, batch_size, spatial_shape, xyz, new_xyz, new_coords, max_range, radius, nsample):
        idx, empty = pooling_ref.voxel_query(max_range, radius, nsample, _np(xyz), _np(new_xyz), _np(new_coords), _np(ws))
        return torch.from_numpy(idx), torch.from_numpy(empty)

    def group_points(self, features, features_batch_cnt, idx, idx_batch_cnt):
        return torch.from_numpy(pooling_ref.group_points(_np(features), _np(features_batch_cnt), _np(idx),
                                                         _np(idx_batch_cnt)))

    def group_points_grad(self, grad_out, idx, idx_batch_cnt, features_batch_cnt, n):
        return torch.from_numpy(pooling_ref.group_points_grad(_np(grad_out), _np(idx), _np(idx_batch_cnt),
                                                              _np(features_batch_cnt), int(n)))

    # ------------------------------------------------------------------ voxelise + MeanVFE
    def voxelize_mean(self, points, pc_range, voxel_size, max_points, max_voxels, vfe_max_last):
        vox, coords, num = geometry.voxelize(_np(points), voxel_size, pc_range, max_points, max_voxels)
        feats = geometry.mean_vfe(vox, num, "max" if vfe_max_last else None)
        return torch.from_numpy(feats), torch.from_numpy(coords), torch.from_numpy(num)

    def voxelize(self, points, pc_range, voxel_size, max_points, max_voxels):
        vox, coords, num = geometry.voxelize(_np(points), voxel_size, pc_range, max_points, max_voxels)
        return torch.from_numpy(vox), torch.from_numpy(coords), torch.from_numpy(num)

    # ------------------------------------------------------------------ input point discard + fused front-end
    def input_discard(self, points, bin_num, rate, max_dis=60.0, perms=None, seed=0, sync=True):
        """oracle/geometry.input_point_discard with per-bin injected permutations ({bin: perm}); bins without one draw from
        a numpy generator seeded with `seed`."""
        pts = _np(points).astype(np.float32)
        out = geometry.input_point_discard_binned(pts, bin_num, rate, max_dis, perms, np.random.default_rng(seed))
        return torch.from_numpy(out), torch.tensor([out.shape[0]], dtype=torch.int32)

    def frontend_voxelize_mean(self, lidar, virtual, bin_num, rate, pc_range, voxel_size, max_points, max_voxels,
                               vfe_max_last, max_dis=60.0, perms=None, seed=0, intensity_div=0.0, sync=True):
        virt, _ = self.input_discard(virtual, bin_num, rate, max_dis, perms, seed)
        pts = np.concatenate([_np(lidar).astype(np.float32), _np(virt)])
        if intensity_div:
            pts[:, 3] /= np.float32(intensity_div)
        f, c, n = self.voxelize_mean(torch.from_numpy(pts), pc_range, voxel_size, max_points, max_voxels, vfe_max_last)
        if sync:
            return f, c, n
        return f, c, n, torch.tensor([f.shape[0], pts.shape[0]], dtype=torch.int32)

    # ------------------------------------------------------------------ BatchNorm(+ReLU)
    def bn_forward(self, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, out=None, out_col0=0,
                   num_batches_tracked=None):
        x = x.detach()
        if training and num_batches_tracked is not None:
            num_batches_tracked.add_(1)
        n = x.shape[0]
        if training:
            xd = x.double()
            mean = xd.mean(0)
            var = (xd * xd).mean(0) - mean * mean
            var = var.clamp_min(0)
            if running_mean is not None:
                running_mean.mul_(1 - momentum).add_(momentum * mean.float())
                unb = var * n / max(n - 1, 1)
                running_var.mul_(1 - momentum).add_(momentum * unb.float())
            mean, var = mean.float(), var.float()
        else:
            mean, var = running_mean, running_var
        y = (x - mean) / torch.sqrt(var + eps) * gamma.detach() + beta.detach()
        if relu:
            y = torch.relu(y)
        return y, mean, var

    def bn_backward(self, x, dy, dy_col0, mean, var, gamma, beta, eps, relu, absmax_ws=None):
        n, c = x.shape
        dy = dy[:, dy_col0:dy_col0 + c]
        istd = 1.0 / torch.sqrt(var + eps)
        xh = (x - mean) * istd
        if relu:
            dy = torch.where(xh * gamma + beta > 0, dy, torch.zeros_like(dy))
        dbeta = dy.double().sum(0).float()
        dgamma = (dy * xh).double().sum(0).float()
        dx = gamma * istd * (dy - dbeta / n - xh * dgamma / n)
        return dx, dgamma, dbeta
