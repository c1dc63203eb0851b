"""SparseModule / SparseSequential (used at spconv_backbone.py:101-105,125-129,248-291,561-567)."""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import nn

from .. import ops
from .core import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base: modules that take and return a SparseConvTensor."""


def is_spconv_module(m: nn.Module) -> bool:
    return isinstance(m, SparseModule)


class SparseSequential(SparseModule):
    """Runs modules in order.  SparseModules see the SparseConvTensor; plain nn.Modules (BatchNorm1d, ReLU) are applied
    to ``.features`` (N, C) and re-wrapped.  ``BatchNorm1d`` [+ ``ReLU``] pairs run as ONE fused HIP pass
    (vc_bn_stats + vc_bn_apply_relu) unless ``fuse_bn_relu`` is False; parameters stay in the stock nn.BatchNorm1d so
    state_dict keys are unchanged."""

    fuse_bn_relu = True

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        mods = tuple(self._modules.values())     # (three entries in this code base: cheaper than walking an iterator)
        if not (-len(mods) <= idx < len(mods)):
            raise IndexError("index {} is out of range".format(idx))
        return mods[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.values())
        k = 0
        while k < len(mods):
            module = mods[k]
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                nxt = mods[k + 1] if k + 1 < len(mods) else None
                eval_fusable = (nxt is not None and not nxt.training and not torch.is_grad_enabled())
                if (self.fuse_bn_relu and getattr(module, "fusable_with_bn", False) and type(nxt) is nn.BatchNorm1d
                        and (nxt.training or eval_fusable) and nxt.affine and nxt.track_running_stats
                        and nxt.momentum is not None and input.features.is_cuda and input.features.shape[0] != 0
                        and nxt.num_features % 4 == 0):
                    relu = k + 2 < len(mods) and type(mods[k + 2]) is nn.ReLU
                    input = module(input, fuse_bn=nxt, fuse_relu=relu)   # conv + BN (+ ReLU) as one autograd node
                    k += 3 if relu else 2
                    continue
                input = module(input)
                k += 1
                continue
            if isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    feats = input.features
                    fused = (self.fuse_bn_relu and type(module) is nn.BatchNorm1d and feats.is_cuda
                             and module.affine and module.momentum is not None)
                    if fused:
                        relu = k + 1 < len(mods) and type(mods[k + 1]) is nn.ReLU
                        input = input.replace_feature(ops.bn_relu(feats, module, relu))
                        k += 2 if relu else 1
                        continue
                    input = input.replace_feature(module(feats))
            else:
                input = module(input)
            k += 1
        return input
