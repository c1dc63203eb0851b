"""Checkpoint compatibility for the sparse-conv weights.

Reference behaviour restated (not copied):
  * pcdet/utils/spconv_utils.py:41-55   find_all_spconv_keys: state_dict keys of every SparseConvolution weight.
  * pcdet/models/detectors/detector3d_template.py:350-381   _load_state_dict: when a sparse-conv weight on disk has a
    different shape from the live one, re-lay it out (spconv 1.x stores (*k, Cin, Cout); spconv 2.x stores
    (Cout, *k, Cin)), keep only keys whose shape then matches, and load strictly or by merging into the live state.

This package's canonical layout is spconv 2.x's (Cout, *k, Cin) (include/virconv_hip.h, vc_conv_forward), so released
VirConv checkpoints (trained with spconv 2.1) load unchanged and spconv-1.x checkpoints are permuted on load.
"""
from __future__ import annotations

from typing import Dict, Set, Tuple

import torch
from torch import nn

from .spconv.conv import SparseConvolution


def find_all_spconv_keys(model: nn.Module, prefix: str = "") -> Set[str]:
    """Names (state_dict keys) of the weights of all sparse convolutions below `model`."""
    keys = set()
    for name, mod in model.named_modules(prefix=prefix):
        if isinstance(mod, SparseConvolution):
            keys.add(f"{name}.weight" if name else "weight")
    return keys


def adapt_spconv_weight(val: torch.Tensor, want_shape: torch.Size, source_layout: str = "auto") -> torch.Tensor:
    """Return `val` re-laid out to `want_shape` = (Cout, *k, Cin) if it is a known foreign layout, else `val` itself.

    On-disk layouts of spconv conv weights (detector3d_template.py:358-370):
      (Cout, *k, Cin)   spconv 2.x implicit-GEMM builds -- ours; released VirConv checkpoints
      (*k, Cin, Cout)   spconv 1.x              -> move the last axis to the front         ('spconv1')
      (*k, Cout, Cin)   spconv 2.x "native"     -> move the second-to-last axis to the front ('native')
    The two foreign layouts have the same shape when Cin == Cout; `source_layout='auto'` then assumes spconv 1.x (the case
    the reference's loader is written for), pass 'native' to override.  (The reference's transpose(-1, -2) branch serves a
    LIVE model in native layout; our live layout is fixed.)
    """
    assert source_layout in ("auto", "spconv1", "native")
    if tuple(val.shape) == tuple(want_shape) or val.dim() != len(want_shape) or val.dim() < 3:
        return val
    nd = val.dim()
    cands = []
    if source_layout in ("auto", "spconv1"):
        cands.append(val.permute(nd - 1, *range(nd - 1)))                    # (*k,Cin,Cout) -> (Cout,*k,Cin)
    if source_layout in ("auto", "native"):
        cands.append(val.permute(nd - 2, *range(nd - 2), nd - 1))            # (*k,Cout,Cin) -> (Cout,*k,Cin)
    for v in cands:
        if tuple(v.shape) == tuple(want_shape):
            return v.contiguous()
    return val


def load_state_dict_adapted(model: nn.Module, state_disk: Dict[str, torch.Tensor], *, strict: bool = True,
                            source_layout: str = "auto") -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """Load `state_disk` into `model` the way Detector3DTemplate._load_state_dict does.

    Returns (live state_dict after the update, the subset that was actually taken from disk).  Keys missing from the model
    or whose shape still differs after adaptation are skipped; with strict=True the remaining set must cover the model
    (torch raises otherwise), with strict=False it is merged into the live state.
    """
    live = model.state_dict()
    conv_keys = find_all_spconv_keys(model)
    taken: Dict[str, torch.Tensor] = {}
    for key, val in state_disk.items():
        if key not in live:
            continue
        if key in conv_keys and live[key].shape != val.shape:
            val = adapt_spconv_weight(val, live[key].shape, source_layout)
        if live[key].shape == val.shape:
            taken[key] = val
    if strict:
        model.load_state_dict(taken)
    else:
        live.update(taken)
        model.load_state_dict(live)
    return live, taken
