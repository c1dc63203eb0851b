"""The first BEV conv on the sparse rows ("HeightCompression + first BaseBEVBackbone conv fused").

Reference: ``HeightCompression.forward`` folds the height axis of ``encoded_spconv_tensor.dense()`` into the channels
(pcdet/models/backbones_2d/map_to_bev/height_compression.py:27-31: (B, 64, 4, 200, 176) -> (B, 256, 200, 176)) and the first block of
``BaseBEVBackbone`` runs ``ZeroPad2d(1) + Conv2d(256 -> 64, k3, bias=False) + BatchNorm2d(eps 1e-3, momentum 0.01) + ReLU`` over it
(pcdet/models/backbones_2d/base_bev_backbone.py:31-38) -- a dense 41.5 GFLOP conv over a map in which ~70 % of the cells hold no
voxel.  Seen from the sparse tensor it is a sparse conv from the (b, z, y, x) rows onto the (b, y, x) cells with kernel (4, 3, 3):
4.7 GFLOP on the synthetic KITTI frames.  ``SparseBEVStem`` runs it that way (vc_bev_pairs -> the gather-GEMM in two passes of 18
kernel offsets -> vc_nhwc_to_nchw with BatchNorm + ReLU folded in).  Round 5: also when a gradient is needed (training through the
stem, tools/train_utils/train_utils.py:47): the conv is one autograd node over the sparse rows -- dX = a forward-form gather-GEMM of dY
over the transposed table (vc_bev_pairs_backward), dW = the weight-gradient kernel over the forward table -- and BatchNorm2d + ReLU are
the BEV backbone's own modules applied to the NHWC rows viewed as a channels-last (B, C, H, W) tensor (statistics over the dense
extent, empty cells included, exactly as the dense recipe).  Parameters stay where they are (the BaseBEVBackbone's own Conv2d /
BatchNorm2d: same state_dict keys).
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops

MAX_PASS_OFFSETS = 32      # the gather-GEMM walks a 32-bit mask of kernel offsets per block


class _StemConvFunction(torch.autograd.Function):
    """y (cells, Cout) = the first BEV conv over the sparse rows; cells = every (b, y, x) of the map in dense order (NHWC rows).
    Backward: d features and d weight (the Conv2d parameter, (Cout, C * D, ky, kx)) through the sparse kernels."""

    @staticmethod
    def forward(ctx, feats, weight, indices, spatial_shape, batch_size, passes):
        be = ops.get_backend()
        y, pair = be.bev_stem_conv(feats, indices, spatial_shape, batch_size, passes, weight.shape[0], want_nhwc=True, want_pairs=True,
                                   ksize=(weight.shape[2], weight.shape[3]))
        ctx.save_for_backward(feats, weight)
        ctx.weight_version = weight._version
        ctx.geom = (indices, tuple(int(v) for v in spatial_shape), int(batch_size), passes, pair)
        return y

    @staticmethod
    def backward(ctx, gy):
        feats, weight = ctx.saved_tensors
        indices, shape, bs, passes, pair = ctx.geom
        be = ops.get_backend()
        if weight._version != ctx.weight_version:     # `passes` were packed from the weight as it was at forward time
            raise RuntimeError("SparseBEVStem: the conv weight was modified in place between forward and backward")
        dx, dws = be.bev_stem_conv_backward(feats, indices, shape, bs, passes, weight.shape[0], pair, gy.contiguous(),
                                            ctx.needs_input_grad[0], ctx.needs_input_grad[1], ksize=(weight.shape[2], weight.shape[3]))
        dw = None
        if dws is not None:
            cout, cd, ky, kx = weight.shape
            depth = shape[0]
            full = torch.cat(dws, 1)                                           # (Cout, D * ky * kx, C), offsets (z, ky, kx) row-major
            dw = full.view(cout, depth, ky, kx, cd // depth).permute(0, 4, 1, 2, 3).reshape(cout, cd, ky, kx)   # channel = c * D + z
        return dx, dw, None, None, None, None


def pack_stem_weight(weight: torch.Tensor, depth: int):
    """Conv2d weight (Cout, C * D, ky, kx) over the height-compressed map (channel = c * D + z, height_compression.py:30) -> passes
    [(C, kv_pass, Cout), ...] of the sparse form: offset index (z, ky, kx) row-major, z split so that a pass has <= 32 offsets."""
    cout, cd, ky, kx = weight.shape
    assert cd % depth == 0
    c = cd // depth
    w = weight.detach().view(cout, c, depth, ky, kx).permute(1, 2, 3, 4, 0).contiguous()      # (C, D, ky, kx, Cout)
    z_per = max(1, MAX_PASS_OFFSETS // (ky * kx))
    return [w[:, z0:z0 + z_per].reshape(c, -1, cout).contiguous() for z0 in range(0, depth, z_per)]


class SparseBEVStem(nn.Module):
    """``stem(encoded_spconv_tensor) -> (B, Cout, H, W)``: conv + BatchNorm2d + ReLU of the first BEV block, given that block
    (``nn.Sequential(ZeroPad2d(1), Conv2d, BatchNorm2d, ReLU, ...)``).  Holds no parameters of its own."""

    def __init__(self, first_block: nn.Sequential):
        super().__init__()
        pad, conv, bn, relu = first_block[0], first_block[1], first_block[2], first_block[3]
        assert isinstance(conv, nn.Conv2d) and conv.bias is None and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
        assert conv.kernel_size[0] == conv.kernel_size[1] and conv.kernel_size[0] % 2 == 1
        assert isinstance(bn, nn.BatchNorm2d) and isinstance(relu, nn.ReLU)
        k = conv.kernel_size[0]
        assert (isinstance(pad, nn.ZeroPad2d) and tuple(pad.padding) == (k // 2,) * 4 and conv.padding == (0, 0)) or \
               (isinstance(pad, nn.Identity) and conv.padding in ((0, 0), (k // 2, k // 2)))
        object.__setattr__(self, "_block", first_block)       # not a submodule: the parameters belong to the BEV backbone
        self._packed = None

    def _passes(self, depth: int):
        conv = self._block[1]
        key = (conv.weight.data_ptr(), conv.weight._version, depth, conv.weight.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, pack_stem_weight(conv.weight, depth))
        return self._packed[1]

    def sparse_path_usable(self, t) -> bool:
        conv, bn = self._block[1], self._block[2]
        be = ops.get_backend()
        return bool(hasattr(be, "bev_stem_conv") and t.features.is_cuda and t.features.shape[1] in (16, 32, 64)
                    and conv.out_channels in (16, 32, 64) and len(t.spatial_shape) == 3 and t.features.dtype == torch.float32
                    and conv.in_channels == t.features.shape[1] * t.spatial_shape[0] and not (bn.training and bn.track_running_stats is False))

    def forward(self, t):
        conv, bn = self._block[1], self._block[2]
        if not self.sparse_path_usable(t):
            # the reference recipe on the dense map, border written by the dense pass (HeightCompression(BEV_PAD) form)
            k = conv.kernel_size[0]
            d = t.dense(pad=(k // 2, k // 2)) if conv.padding == (0, 0) else t.dense()
            x = d.view(d.shape[0], -1, d.shape[-2], d.shape[-1])
            return self._block[3](bn(conv(x)))
        be = ops.get_backend()
        depth = int(t.spatial_shape[0])
        passes = self._passes(depth)
        if torch.is_grad_enabled() and (t.features.requires_grad or conv.weight.requires_grad or
                                        (bn.weight is not None and bn.weight.requires_grad)):
            # training through the stem: the conv as one autograd node over the sparse rows, BatchNorm2d + ReLU as the BEV backbone's
            # own modules on the NHWC rows seen as a channels-last (B, C, H, W) tensor (statistics over every cell of the map)
            y = _StemConvFunction.apply(t.features, conv.weight, t.indices, t.spatial_shape, t.batch_size, passes)
            # (B, C, H, W) in channels-last strides: what the NHWC rows are, and what MIOpen's BatchNorm / the next convs take as is.  A
            # consumer that calls .view() on the result needs .contiguous() first (the dense recipe's output is NCHW-contiguous).
            y4 = y.view(t.batch_size, int(t.spatial_shape[1]), int(t.spatial_shape[2]), conv.out_channels).permute(0, 3, 1, 2)
            return self._block[3](bn(y4))
        if bn.training:   # batch statistics over the WHOLE map (zeros of the empty cells included): the NHWC rows hold every cell
            y = be.bev_stem_conv(t.features, t.indices, t.spatial_shape, t.batch_size, passes, conv.out_channels, want_nhwc=True,
                                 ksize=conv.kernel_size)
            with torch.no_grad():
                var, mean = torch.var_mean(y, dim=0, unbiased=False)
                if bn.track_running_stats:
                    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    n = y.shape[0]
                    bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                    bn.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
                    bn.num_batches_tracked += 1
            scale = bn.weight.detach() * torch.rsqrt(var + bn.eps)
            shift = bn.bias.detach() - mean * scale
            dense = torch.empty((t.batch_size, conv.out_channels, int(t.spatial_shape[1]), int(t.spatial_shape[2])),
                                dtype=torch.float32, device=y.device)
            from ._lib import check
            check(be.lib.vc_nhwc_to_nchw(y.data_ptr(), t.batch_size, dense.shape[2] * dense.shape[3], conv.out_channels,
                                         scale.contiguous().data_ptr(), shift.contiguous().data_ptr(), 1, dense.data_ptr(), be.stream()),
                  "vc_nhwc_to_nchw")
            return dense
        scale = (bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
        shift = (bn.bias.detach() - bn.running_mean * scale).contiguous()
        return be.bev_stem_conv(t.features.detach(), t.indices, t.spatial_shape, t.batch_size, passes, conv.out_channels, scale, shift, True,
                                ksize=conv.kernel_size)


class _StemSkippingBlock(nn.Sequential):
    """The first block of a BaseBEVBackbone whose first four modules (pad, conv, BatchNorm, ReLU) already ran inside
    HeightCompression (SparseBEVStem): same children, same state_dict keys, forward starts at the fifth module."""

    def forward(self, x):
        for m in list(self)[4:]:
            x = m(x)
        return x
