"""Sparse convolution modules with spconv's names and constructor signatures (SURVEY §8b operator API).

``SparseConvolution`` is the base the reference discovers with ``isinstance(child, spconv.conv.SparseConvolution)``
(pcdet/utils/spconv_utils.py:49); ``.weight`` has the spconv-2.x canonical layout (Cout, *kernel, Cin) so released
VirConv checkpoints load unchanged (detector3d_template.py:358-370).  Same ``indice_key`` => the rulebook cached in
``x.indice_dict`` is reused.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import ops
from .core import SparseConvTensor
from .modules import SparseModule


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 algo=None, fp32_accum=None, name=None):
        super().__init__()
        assert groups == 1, "groups != 1 is not supported"
        assert not transposed, "transposed sparse conv is not used by the reference and not supported"
        self.ndim = ndim
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size = list(ops.ntuple(kernel_size, ndim))
        self.stride = list(ops.ntuple(stride, ndim))
        self.padding = list(ops.ntuple(padding, ndim))
        self.dilation = list(ops.ntuple(dilation, ndim))
        self.conv1x1 = all(k == 1 for k in self.kernel_size) and all(s == 1 for s in self.stride)
        self.subm, self.inverse, self.transposed = subm, inverse, transposed
        self.groups, self.output_padding = groups, output_padding
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(self.out_channels, *self.kernel_size, self.in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(self.out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        kv = 1
        for k in self.kernel_size:
            kv *= k
        bound = 1.0 / math.sqrt(kv * self.in_channels)  # kaiming_uniform(a=sqrt(5)) on fan_in = KV*Cin
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, subm={self.subm}, inverse={self.inverse}, indice_key={self.indice_key}")

    def _rulebook(self, x: SparseConvTensor) -> ops.Rulebook:
        rb = x.find_indice_pair(self.indice_key)
        if self.inverse:
            assert rb is not None and rb.kind == "sparse", "inverse conv needs the rulebook of its SparseConv (indice_key)"
            return rb
        if rb is not None:
            if self.subm:
                ok = rb.kind == "subm" and rb.n_in == x.indices.shape[0] and list(rb.ksize) == self.kernel_size
                assert ok, f"indice_key {self.indice_key!r} cached for a different tensor/kernel"
            return rb
        if self.subm:
            # 2-D image-space tensors carry duplicate pixels (spconv_backbone.py:217-222): SURVEY App-A.5 rule
            rb = ops.build_subm_rulebook(x.indices, x.spatial_shape, self.kernel_size, self.dilation,
                                         allow_duplicates=(self.ndim == 2))
        else:
            rb = ops.build_sparse_rulebook(x.indices, x.spatial_shape, x.batch_size, self.kernel_size, self.stride,
                                           self.padding, self.dilation)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        return rb

    @property
    def fusable_with_bn(self) -> bool:
        # the fused conv+BN(+ReLU) operators exist for the channel counts the kernels are instantiated for; anything else takes
        # the generic (channel-tiled) ops.sparse_conv + ops.bn_relu
        return self.bias is None and self.in_channels in ops._CONV_CHANNELS and self.out_channels in ops._CONV_CHANNELS

    def forward(self, x: SparseConvTensor, fuse_bn=None, fuse_relu: bool = False) -> SparseConvTensor:
        """`fuse_bn` (set by SparseSequential): the BatchNorm1d that follows this conv, folded into the same operator --
        training mode: one autograd node, statistics from the conv epilogue; eval mode without grad: one kernel launch."""
        assert isinstance(x, SparseConvTensor)
        assert x.features.shape[1] == self.in_channels, "channel size mismatch"
        assert len(x.spatial_shape) == self.ndim
        rb = self._rulebook(x)
        if fuse_bn is not None and not fuse_bn.training:
            feats = ops.conv_bn_relu_eval(x.features, self.weight, rb, self.inverse, fuse_bn, fuse_relu)
            if feats is None:  # not applicable: plain conv, then the modules themselves
                feats = ops.sparse_conv(x.features, self.weight, rb, self.inverse)
                feats = ops.bn_relu(feats, fuse_bn, fuse_relu)
        elif fuse_bn is not None:
            feats = ops.conv_bn_relu(x.features, self.weight, rb, self.inverse, fuse_bn, fuse_relu)
        else:
            feats = ops.sparse_conv(x.features, self.weight, rb, self.inverse)
        if self.bias is not None:
            feats = feats + self.bias
        if self.inverse:
            out = SparseConvTensor(feats, rb.in_indices, rb.in_shape, x.batch_size, x.grid, x.voxel_num, x.indice_dict)
        elif self.subm:
            out = SparseConvTensor(feats, x.indices, x.spatial_shape, x.batch_size, x.grid, x.voxel_num, x.indice_dict)
        else:
            out = SparseConvTensor(feats, rb.out_indices, rb.out_shape, x.batch_size, x.grid, x.voxel_num, x.indice_dict)
        out.benchmark, out.benchmark_record = x.benchmark, x.benchmark_record
        return out


def _make(ndim, subm=False, inverse=False):
    class _Conv(SparseConvolution):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                     indice_key=None, algo=None, fp32_accum=None, name=None):
            super().__init__(ndim, in_channels, out_channels, kernel_size, 1 if (subm or inverse) else stride,
                             padding, dilation, groups, bias, subm=subm, inverse=inverse, indice_key=indice_key,
                             algo=algo, fp32_accum=fp32_accum, name=name)
    return _Conv


class SparseConv2d(_make(2)):
    pass


class SparseConv3d(_make(3)):
    pass


class SubMConv2d(_make(2, subm=True)):
    pass


class SubMConv3d(_make(3, subm=True)):
    pass


class SparseInverseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None, name=None):
        super().__init__(2, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)
