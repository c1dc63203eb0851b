"""Gradient clip + Adam step with true weight decay for a model with flat parameters, as two launches (vc_clip_adamw).

Reference: tools/train_utils/train_utils.py:50-51 runs ``clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP)`` and then
``optimizer.step()``; the optimizer (tools/train_utils/optimization/__init__.py:19-32, ``adam_onecycle``) is torch Adam with betas
(0.9, 0.99) inside fastai's ``OptimWrapper`` with ``true_wd=True, bn_wd=True``, whose step (fastai_optim.py:132-149) multiplies every
parameter by ``1 - wd * lr`` and then runs Adam without weight decay -- the arithmetic of ``torch.optim.AdamW``.

``ClipAdamW`` is a ``torch.optim.Optimizer``: ``param_groups[0]["lr"]`` / ``["betas"]`` are read at every step, so a one-cycle scheduler
that rewrites them per iteration (fastai's ``OneCycle``, ``torch.optim.lr_scheduler.OneCycleLR``) drives it unchanged, and its state
(``step``, ``exp_avg``, ``exp_avg_sq`` per parameter) has the keys and shapes of ``torch.optim.AdamW``'s, so optimizer checkpoints move
between the two.  It wants what ``feature_pass.flatten_parameters`` returns: a few contiguous fp32 tensors on the GPU (at most 16).
There is no CPU path: without the HIP library the constructor raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops

from ._lib import ADAM_MAX_TENSORS as MAX_TENSORS, AdamTensor


def supports(params) -> bool:
    """Can ClipAdamW take this parameter list?  (a few contiguous fp32 CUDA tensors: the flat parameters of a flattened model)"""
    params = list(params)
    return (0 < len(params) <= MAX_TENSORS and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params)
            and len({p.device for p in params}) == 1)


class ClipAdamW(torch.optim.Optimizer):
    """``clip_grad_norm_(params, max_norm)`` + ``AdamW.step()`` in two kernel launches.  One parameter group.

    ``step()`` returns the total gradient norm (0-dim tensor on the device, what ``clip_grad_norm_`` returns) without a host read.
    ``max_norm`` None or <= 0: no clipping.  ``scale_grads=True`` also leaves the clipped gradients in ``.grad`` as
    ``clip_grad_norm_`` does (nothing on the training path reads them, so the default skips that write)."""

    clips = True   # a training loop that sees this attribute skips its own clip_grad_norm_ (bench.train_step)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01, max_norm=10.0, scale_grads=False):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, max_norm=max_norm, scale_grads=scale_grads)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("ClipAdamW: one parameter group (the clip is over all parameters at once)")
        ps = self.param_groups[0]["params"]
        if not supports(ps):
            raise ValueError(f"ClipAdamW: wants 1..{MAX_TENSORS} contiguous float32 CUDA tensors on one device "
                             f"(feature_pass.flatten_parameters(model)), got {len(ps)} parameter(s)")
        self._lib = ops.get_backend().lib    # raises without libvirconv_hip.so
        self._device = ps[0].device
        if self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        self._ws = None
        self._norm = None

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)          # as torch.optim.AdamW keeps it (capturable=False)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        live = [p for p in g["params"] if p.grad is not None]
        arr = (AdamTensor * MAX_TENSORS)()
        steps = set()
        for i, p in enumerate(live):
            grad = p.grad
            if grad.dtype != torch.float32 or not grad.is_contiguous() or grad.is_sparse or grad.device != p.device:
                raise ValueError("ClipAdamW: gradients must be dense contiguous float32 tensors on the parameter's device")
            st = self._state_of(p)
            if not torch.is_tensor(st["step"]):            # a checkpoint of an old torch: a Python number
                st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
            elif st["step"].is_cuda:                       # a checkpoint of torch.optim.AdamW(fused=True): the counter lives on the device there;
                st["step"] = st["step"].cpu()              # here the host needs it (one read, once)
            st["step"] += 1
            steps.add(int(st["step"]))
            arr[i] = AdamTensor(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
        if self._norm is None:
            self._norm = torch.zeros((), dtype=torch.float32, device=self._device)
        if not live:
            self._norm.zero_()
            return self._norm if loss is None else loss
        if len(steps) != 1:
            raise ValueError(f"ClipAdamW: the parameters are at different steps {sorted(steps)} (one bias correction per call)")
        n_ws = self._lib.vc_clip_adamw_workspace_bytes(len(live))
        if self._ws is None or self._ws.numel() < n_ws:
            self._ws = torch.empty((n_ws,), dtype=torch.uint8, device=self._device)
        max_norm = g["max_norm"]
        b1, b2 = g["betas"]
        _lib.check(self._lib.vc_clip_adamw(C.cast(arr, C.c_void_p), len(live), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                           float(g["weight_decay"]), steps.pop(), float(max_norm) if max_norm else 0.0,
                                           int(bool(g["scale_grads"])), self._norm.data_ptr(), self._ws.data_ptr(), n_ws,
                                           torch._C._cuda_getCurrentRawStream(self._device.index)), "vc_clip_adamw")
        return self._norm if loss is None else loss

    @property
    def total_norm(self):
        """The gradient norm of the last step (device tensor; None before the first step)."""
        return self._norm
