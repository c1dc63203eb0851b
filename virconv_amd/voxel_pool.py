"""RoI grid pooling over the backbone outputs -- host-side mirror of the reference interface.

Same names, argument meaning and return values as the reference's modules, on top of the HIP operators of
include/virconv_hip.h (vc_voxel_index_build / vc_voxel_query / vc_group_points[_grad]):

  generate_voxel2pinds(sparse_tensor)    pcdet/utils/spconv_utils.py:12-21        -> VoxelIndex (not a dense volume)
  get_voxel_centers(...)                 pcdet/utils/common_utils.py:65-81
  voxel_query / VoxelQuery               pointnet2_stack/voxel_query_utils.py:10-46
  VoxelQueryAndGrouping                  pointnet2_stack/voxel_query_utils.py:49-100
  grouping_operation / GroupingOperation pointnet2_stack/pointnet2_utils.py:48-105
  NeighborVoxelSAModuleMSG               pointnet2_stack/voxel_pool_modules.py:8-130  (same submodules => same state_dict)

MI355X-first difference: the reference scatters a dense (B, Z, Y, X) int32 volume per tensor per step (12 MB per frame at
x_conv3) and every query thread walks up to 729 of its cells; here the index is an occupancy bitmap + the coordinate hash
(0.4 MB per frame), one wave serves one query, a lane tests a whole x-line of the neighbourhood with one bit-field read,
and only occupied cells cost a hash probe.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch
from torch import nn

from .ops import get_backend


@dataclass
class VoxelIndex:
    """What generate_voxel2pinds returns here: an opaque coordinate -> row index of one sparse tensor."""
    ws: torch.Tensor                 # backend workspace (HIP: bitmap + hash bytes; oracle: the dense volume)
    n: int
    batch_size: int
    spatial_shape: Tuple[int, int, int]

    @property
    def shape(self):                 # the reference reads `B, Z, Y, X = point_indices.shape`
        return (self.batch_size,) + tuple(self.spatial_shape)


def generate_voxel2pinds(sparse_tensor) -> VoxelIndex:
    shape = tuple(int(s) for s in sparse_tensor.spatial_shape)
    idx = sparse_tensor.indices.int().contiguous()
    ws = get_backend().voxel_index_build(idx, int(sparse_tensor.batch_size), shape)
    return VoxelIndex(ws, idx.shape[0], int(sparse_tensor.batch_size), shape)


def get_voxel_centers(voxel_coords: torch.Tensor, downsample_times, voxel_size, point_cloud_range) -> torch.Tensor:
    """(N, 3) [z, y, x] -> (N, 3) [x, y, z] metric centres of the (strided) voxels."""
    assert voxel_coords.shape[1] == 3
    centres = voxel_coords[:, [2, 1, 0]].float()
    vs = torch.tensor(voxel_size, device=centres.device).float() * downsample_times
    lo = torch.tensor(point_cloud_range[0:3], device=centres.device).float()
    return (centres + 0.5) * vs + lo


class VoxelQuery(torch.autograd.Function):
    """idx (M, nsample) int32 rows of `xyz` (global) and the empty-ball mask; not differentiable."""

    @staticmethod
    def forward(ctx, max_range: Sequence[int], radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor,
                new_coords: torch.Tensor, point_indices: VoxelIndex):
        assert new_xyz.is_contiguous() and xyz.is_contiguous() and new_coords.is_contiguous()
        vi = point_indices
        idx, empty = get_backend().voxel_query(vi.ws, vi.n, vi.batch_size, vi.spatial_shape, xyz, new_xyz, new_coords,
                                               max_range, radius, nsample)
        ctx.mark_non_differentiable(idx, empty)
        return idx, empty

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None, None


voxel_query = VoxelQuery.apply


class GroupingOperation(torch.autograd.Function):
    """features (N, C), batch-local idx (M, nsample) -> (M, C, nsample); backward = scatter-add."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, features_batch_cnt: torch.Tensor, idx: torch.Tensor,
                idx_batch_cnt: torch.Tensor):
        assert features.is_contiguous() and features_batch_cnt.is_contiguous()
        assert idx.is_contiguous() and idx_batch_cnt.is_contiguous()
        assert features.shape[0] == int(features_batch_cnt.sum()), \
            f"features: {tuple(features.shape)}, features_batch_cnt: {features_batch_cnt}"
        assert idx.shape[0] == int(idx_batch_cnt.sum()), f"idx: {tuple(idx.shape)}, idx_batch_cnt: {idx_batch_cnt}"
        out = get_backend().group_points(features.detach(), features_batch_cnt, idx, idx_batch_cnt)
        ctx.for_backwards = (features.shape[0], idx, features_batch_cnt, idx_batch_cnt)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        n, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        gf = get_backend().group_points_grad(grad_out.contiguous(), idx, idx_batch_cnt, features_batch_cnt, n)
        return gf, None, None, None


grouping_operation = GroupingOperation.apply


class VoxelQueryAndGrouping(nn.Module):
    def __init__(self, max_range: Sequence[int], radius: float, nsample: int):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords: torch.Tensor, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor, new_xyz: torch.Tensor,
                new_xyz_batch_cnt: torch.Tensor, features: torch.Tensor, voxel2point_indices: VoxelIndex):
        """-> grouped_features (M, C, nsample), grouped_xyz (M, 3, nsample), empty_ball_mask (M,)."""
        assert xyz.shape[0] == int(xyz_batch_cnt.sum()), f"xyz: {tuple(xyz.shape)}, xyz_batch_cnt: {xyz_batch_cnt}"
        assert new_coords.shape[0] == int(new_xyz_batch_cnt.sum()), \
            f"new_coords: {tuple(new_coords.shape)}, new_xyz_batch_cnt: {new_xyz_batch_cnt}"
        idx, empty = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords, voxel2point_indices)
        # global rows -> batch-local rows (what the stacked grouping op expects).  The reference does this with a
        # view(batch_size, -1, nsample), i.e. it assumes equally many queries per sample; the offsets below are the same
        # numbers without that assumption.
        starts = torch.cumsum(xyz_batch_cnt.to(torch.int64), 0) - xyz_batch_cnt.to(torch.int64)
        per_query = torch.repeat_interleave(starts, new_xyz_batch_cnt.to(torch.int64)).to(idx.dtype)
        idx = idx - per_query[:, None]
        idx[empty] = 0
        idx = idx.contiguous()
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty


class NeighborVoxelSAModuleMSG(nn.Module):
    """Multi-scale neighbour-voxel set abstraction.  Per scale k:
    f = BN(Conv1d(features)); g = group(f); p = BN(Conv2d(group(xyz) - new_xyz)); out_k = MLP_out(pool(relu(g + p)))."""

    def __init__(self, *, query_ranges: List[List[int]], radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 use_xyz: bool = True, pool_method: str = "max_pool"):
        super().__init__()
        assert len(query_ranges) == len(nsamples) == len(mlps)
        self.groupers, self.mlps_in = nn.ModuleList(), nn.ModuleList()
        self.mlps_pos, self.mlps_out = nn.ModuleList(), nn.ModuleList()
        for rng, radius, nsample, spec in zip(query_ranges, radii, nsamples, mlps):
            self.groupers.append(VoxelQueryAndGrouping(rng, radius, nsample))
            self.mlps_in.append(nn.Sequential(nn.Conv1d(spec[0], spec[1], kernel_size=1, bias=False), nn.BatchNorm1d(spec[1])))
            self.mlps_pos.append(nn.Sequential(nn.Conv2d(3, spec[1], kernel_size=1, bias=False), nn.BatchNorm2d(spec[1])))
            self.mlps_out.append(nn.Sequential(nn.Conv1d(spec[1], spec[2], kernel_size=1, bias=False),
                                               nn.BatchNorm1d(spec[2]), nn.ReLU()))
        self.relu = nn.ReLU()
        if pool_method not in ("max_pool", "avg_pool"):
            raise NotImplementedError(pool_method)
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords, features, voxel2point_indices):
        """new_coords arrive as [b, x, y, z] (ted_head.py:529-532) and are queried as [b, z, y, x].
        -> (M, sum_k mlps[k][-1])"""
        coords = new_coords[:, [0, 3, 2, 1]].contiguous()
        outs = []
        for grouper, mlp_in, mlp_pos, mlp_out in zip(self.groupers, self.mlps_in, self.mlps_pos, self.mlps_out):
            f = mlp_in(features.t().unsqueeze(0)).squeeze(0).t().contiguous()            # (N, C)
            g, gxyz, empty = grouper(coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, f, voxel2point_indices)
            keep = (~empty).to(g.dtype)[:, None, None]
            g = g * keep                                                                # empty balls contribute zeros
            rel = (gxyz - new_xyz.unsqueeze(-1)) * keep                                 # (M, 3, nsample)
            pos = mlp_pos(rel.permute(1, 0, 2).unsqueeze(0))                            # (1, C, M, nsample)
            x = self.relu(g.permute(1, 0, 2).unsqueeze(0) + pos)
            x = x.max(dim=3)[0] if self.pool_method == "max_pool" else x.mean(dim=3)    # (1, C, M)
            outs.append(mlp_out(x).squeeze(0).t())                                      # (M, C_out)
        return torch.cat(outs, dim=1)
