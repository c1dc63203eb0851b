"""Functional operator layer: rulebooks + autograd Functions over the active backend.

The only backend shipped is :class:`virconv_amd.backend_hip.HipBackend` (the C ABI of libvirconv_hip.so).  The backend
is an object so that tests can inject the CPU oracle (``oracle/backend.py``) to exercise host-side logic without a GPU;
nothing in this package ever selects a CPU implementation by itself.
"""
from __future__ import annotations

import os

import contextlib
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

_BACKEND = None


def get_backend():
    global _BACKEND
    if _BACKEND is None:
        from .backend_hip import HipBackend  # raises loudly if libvirconv_hip.so is missing
        _BACKEND = HipBackend()
    return _BACKEND


def set_backend(backend) -> None:
    """Test hook (tests/ only): replace the operator backend."""
    global _BACKEND
    _BACKEND = backend


@contextlib.contextmanager
def use_backend(backend):
    global _BACKEND
    prev = _BACKEND
    _BACKEND = backend
    try:
        yield backend
    finally:
        _BACKEND = prev


def ntuple(v, n: int) -> Tuple[int, ...]:
    if isinstance(v, (list, tuple, np.ndarray)):
        assert len(v) == n, f"expected {n} values, got {v}"
        return tuple(int(x) for x in v)
    return (int(v),) * n


@dataclass
class Rulebook:
    """Cached index structure of one sparse conv (what spconv keeps in ``indice_dict[indice_key]``)."""
    kind: str                       # 'subm' | 'sparse'
    pair_fwd: torch.Tensor          # (KV, n_out) int32: input row feeding output row through offset k, or -1
    pair_bwd: Optional[torch.Tensor]  # (KV, n_in) int32 (sparse only; subm uses the mirrored pair_fwd)
    rep: Optional[torch.Tensor]     # (n,) int32 representative row per row (duplicate-coordinate subm), else None
    n_in: int
    n_out: int
    in_indices: torch.Tensor
    out_indices: torch.Tensor
    in_shape: Tuple[int, ...]
    out_shape: Tuple[int, ...]
    ksize: Tuple[int, ...]
    stride: Tuple[int, ...]
    padding: Tuple[int, ...]
    dilation: Tuple[int, ...]
    # optional scheduling hints (vc_row_order): permutations of the columns of pair_fwd / of the backward-input table
    order_fwd: Optional[torch.Tensor] = None
    order_bwd: Optional[torch.Tensor] = None
    # SubM rulebook over rows that are in ascending coordinate order (the output of a strided conv): the gather-GEMM may
    # stage its gathers through LDS row windows (VC_CONV_SORTED_ROWS)
    sorted_rows: bool = False
    # duplicate-pixel tables: (2, n) int32 [rows sorted stably by representative | the sorted representatives] -- fixes the
    # order of the additions of the backward's group sum (vc_group_sum_sorted)
    grp_plan: Optional[torch.Tensor] = None

    @property
    def kv(self) -> int:
        n = 1
        for k in self.ksize:
            n *= int(k)
        return n

    @property
    def centre(self) -> int:          # row-major index of the centre tap (what np.ravel_multi_index gives)
        c = 0
        for k in self.ksize:
            c = c * int(k) + int(k) // 2
        return c


class PlanRulebook(Rulebook):
    """A Rulebook whose tensors are views of a geometry-plan arena, CREATED ON FIRST ACCESS (round 6: host time).  A native plan has ~13
    tables of up to eight views each; the native feature pass needs their addresses only (`ptr`), a Python consumer (tests, the
    node-by-node path, a plan observer) gets ordinary tensors the first time it touches a field.  `views`: name -> tensor | None |
    (arena int32 tensor, offset in words, rows, cols or 0 for a 1-D view)."""
    _TENSORS = ("pair_fwd", "pair_bwd", "rep", "in_indices", "out_indices", "order_fwd", "order_bwd", "grp_plan")

    def __init__(self, kind, n_in, n_out, in_shape, out_shape, ksize, stride, padding, dilation, views):   # noqa: D107 (no dataclass init)
        d = self.__dict__
        d["kind"], d["n_in"], d["n_out"] = kind, int(n_in), int(n_out)
        d["in_shape"], d["out_shape"], d["ksize"], d["stride"], d["padding"], d["dilation"] = in_shape, out_shape, ksize, stride, padding, dilation
        d["sorted_rows"] = False
        for name in self._TENSORS:
            d["_v_" + name] = views.get(name)

    def ptr(self, name: str):
        """Device address of a tensor field (None when absent) without creating the view."""
        v = self.__dict__["_v_" + name]
        if v is None:
            return None
        if type(v) is tuple:
            return v[0].data_ptr() + 4 * v[1]
        return v.data_ptr()

    def has(self, name: str) -> bool:
        return self.__dict__["_v_" + name] is not None


def _plan_view_property(name):
    key = "_v_" + name

    def get(self):
        v = self.__dict__[key]
        if type(v) is tuple:
            a, off, rows, cols = v
            v = torch.as_strided(a, (rows, cols), (cols, 1), off) if cols else torch.as_strided(a, (rows,), (1,), off)
            self.__dict__[key] = v
        return v

    def put(self, value):
        self.__dict__[key] = value

    return property(get, put)


for _n in PlanRulebook._TENSORS:
    setattr(PlanRulebook, _n, _plan_view_property(_n))


def rulebook_ptr(rb: Rulebook, name: str):
    """Address of `rb.<name>` (None when absent); for a PlanRulebook without materialising the view."""
    if isinstance(rb, PlanRulebook):
        return rb.ptr(name)
    t = getattr(rb, name)
    return None if t is None else t.data_ptr()


def build_subm_rulebook(indices: torch.Tensor, spatial_shape, ksize, dilation=1, allow_duplicates: bool = False) -> Rulebook:
    ndim = indices.shape[1] - 1
    ks, dl = ntuple(ksize, ndim), ntuple(dilation, ndim)
    shape = tuple(int(s) for s in spatial_shape)
    pair, rep = get_backend().subm_rulebook(indices, shape, ks, dl, want_rep=allow_duplicates)
    n = indices.shape[0]
    rb = Rulebook("subm", pair, None, rep if allow_duplicates else None, n, n, indices, indices, shape, shape, ks,
                  (1,) * ndim, tuple(k // 2 for k in ks), dl)
    # strided-conv outputs are emitted in ascending (b, z, y, x) order and tagged below; a SubM conv on them reads a sorted table
    rb.sorted_rows = bool(WINDOW_GATHER and getattr(indices, "_vc_sorted", False) and not allow_duplicates)
    if rb.rep is not None and hasattr(get_backend(), "group_plan") and indices.is_cuda and torch.is_grad_enabled():
        rb.grp_plan = get_backend().group_plan(rb.rep)
    if rb.rep is not None and REP_FIRST_ORDER and ROW_ORDER != "all" and hasattr(get_backend(), "rep_order") and indices.is_cuda:
        # duplicate-pixel table: the backward-input walks representatives first (homogeneous tiles, see vc_rep_order)
        rb.order_bwd = get_backend().rep_order(rb.rep)
    if ROW_ORDER == "all" and rb.kv <= 32:
        be = get_backend()
        rb.order_fwd = be.row_order(pair, window=ROW_ORDER_WINDOW)
        # mirrored table => same active sets; duplicate-pixel rows take the centre tap only in the backward
        rb.order_bwd = (be.row_order(pair, rb.rep, rb.centre, window=ROW_ORDER_WINDOW) if rb.rep is not None
                        else rb.order_fwd)
    return rb


def begin_sparse_rulebook(indices: torch.Tensor, spatial_shape, batch_size: int, ksize, stride, padding, dilation=1):
    """Start a strided-conv rulebook (output-cell marking + the count's trip to the host) and return a handle for
    `finish_sparse_rulebook`: the geometry plan issues its other kernels in between, so the count read does not stall it."""
    ndim = indices.shape[1] - 1
    ks, st = ntuple(ksize, ndim), ntuple(stride, ndim)
    pd, dl = ntuple(padding, ndim), ntuple(dilation, ndim)
    shape = tuple(int(s) for s in spatial_shape)
    be = get_backend()
    args = (indices, shape, int(batch_size), ks, st, pd, dl)
    handle = be.sparse_rulebook_begin(*args) if hasattr(be, "sparse_rulebook_begin") else None
    return {"args": args, "handle": handle}


def _sparse_rulebook_from(indices, shape, out_idx, out_shape, pf, pb, ks, st, pd, dl) -> Rulebook:
    be = get_backend()
    out_idx._vc_sorted = True  # ascending linear order by construction (bitmap rank); see build_subm_rulebook
    rb = Rulebook("sparse", pf, pb, None, indices.shape[0], out_idx.shape[0], indices, out_idx, shape,
                  tuple(int(s) for s in out_shape), ks, st, pd, dl)
    # the row order only serves the backward-input conv: not built when no gradient will flow (inference)
    if ROW_ORDER in ("bwd", "strided", "all") and 8 < rb.kv <= 32 and torch.is_grad_enabled():
        rb.order_bwd = be.row_order(pb, window=ROW_ORDER_WINDOW)
    if ROW_ORDER in ("strided", "all") and 8 < rb.kv <= 32:
        # forward table of a strided conv: 4.9 active offsets per row but 17.7 per 16-row tile in natural order (72 % of the
        # issued MFMAs are padding); mask-sorted rows bring the tile union down (-25 % measured)
        rb.order_fwd = be.row_order(pf, window=ROW_ORDER_WINDOW)
    return rb


# One host read for a whole chain of strided convs (HipBackend.sparse_rulebook_chain) instead of one per conv; "0" = off
CHAIN_RULEBOOKS = os.environ.get("VIRCONV_CHAIN_RULEBOOKS", "1") != "0"


def build_sparse_rulebook_chain(indices: torch.Tensor, spatial_shape, batch_size: int, convs):
    """Rulebooks of strided convs applied one after the other on an unchanged active set (`convs`: objects with kernel_size /
    stride / padding / dilation), or None when the backend has no chained path.  -> [{"ready": Rulebook}, ...]: handles that
    `finish_sparse_rulebook` accepts in place of a begun rulebook."""
    be = get_backend()
    if not (CHAIN_RULEBOOKS and hasattr(be, "sparse_rulebook_chain") and len(convs) >= 2):
        return None
    ndim = indices.shape[1] - 1
    geoms = [(ntuple(c.kernel_size, ndim), ntuple(c.stride, ndim), ntuple(c.padding, ndim), ntuple(c.dilation, ndim)) for c in convs]
    shape = tuple(int(s) for s in spatial_shape)
    out = []
    for (ks, st, pd, dl), (out_idx, out_shape, pf, pb, src) in zip(geoms, be.sparse_rulebook_chain(indices, shape, int(batch_size), geoms)):
        out.append({"ready": _sparse_rulebook_from(src, shape, out_idx, out_shape, pf, pb, ks, st, pd, dl)})
        shape = tuple(int(s) for s in out_shape)
    return out


def finish_sparse_rulebook(pending) -> Rulebook:
    if "ready" in pending:
        return pending["ready"]
    indices, shape, batch_size, ks, st, pd, dl = pending["args"]
    be = get_backend()
    if pending["handle"] is not None:
        out_idx, out_shape, pf, pb = be.sparse_rulebook_finish(pending["handle"])
    else:
        out_idx, out_shape, pf, pb = be.sparse_rulebook(indices, shape, batch_size, ks, st, pd, dl)
    return _sparse_rulebook_from(indices, shape, out_idx, out_shape, pf, pb, ks, st, pd, dl)


def build_sparse_rulebook(indices: torch.Tensor, spatial_shape, batch_size: int, ksize, stride, padding, dilation=1) -> Rulebook:
    return finish_sparse_rulebook(begin_sparse_rulebook(indices, spatial_shape, batch_size, ksize, stride, padding, dilation))


class SparseConvFunction(torch.autograd.Function):
    """y = conv(features, weight | rulebook); replaces spconv's SparseConvFunction / SubMConvFunction /
    SparseImplicitGemmFunction."""

    @staticmethod
    def forward(ctx, features, weight, rb: Rulebook, inverse: bool):
        be = get_backend()
        ctx.rb, ctx.inverse = rb, inverse
        ctx.save_for_backward(features, weight)
        if inverse:
            return be.conv_forward(features, weight, rb.pair_bwd, order=rb.order_bwd, operand=MFMA_OPERAND)
        return be.conv_forward(features, weight, rb.pair_fwd, order=rb.order_fwd, operand=MFMA_OPERAND,
                               sorted_rows=rb.sorted_rows)

    @staticmethod
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        dx, dw = _conv_backward(ctx.rb, ctx.inverse, features, weight, grad_out.contiguous(), ctx.needs_input_grad[0],
                                ctx.needs_input_grad[1])
        return dx, dw, None, None


def _conv_backward(rb: "Rulebook", inverse: bool, features, weight, grad_out, need_dx: bool, need_dw: bool,
                   group_ws=None):
    """dX and dW of one sparse conv.  dW and dX are independent and both latency-bound: dW runs on a side stream
    underneath dX and is joined before returning (autograd / DDP hooks only ever see completed gradients)."""
    be = get_backend()
    dx = dw = None
    tbl_w = rb.pair_bwd if inverse else rb.pair_fwd
    side = _side_stream(grad_out.device) if (OVERLAP_WEIGHT_GRAD and need_dx and need_dw and grad_out.is_cuda) else None
    # duplicate-pixel table with a group plan: ONE group sum feeds the backward-input conv and the weight gradient (whose
    # non-centre offsets then walk the representatives only) -- the same launches as vc_post_act_block_backward
    dup_kw, grp = {}, None
    cout = weight.shape[0]
    if (need_dx and not inverse and rb.kind == "subm" and rb.rep is not None and rb.grp_plan is not None and grad_out.is_cuda
            and (cout & (cout - 1)) == 0 and hasattr(be, "group_sum_sorted")):
        grp = be.group_sum_sorted(grad_out, rb.grp_plan)
        dup_kw = dict(rep=rb.rep, centre=rb.centre, dy_grp=grp)
    if side is not None:
        # buffers are allocated on the main stream and kept referenced until the join below, which orders every later
        # reuse after the side stream's work (no record_stream bookkeeping needed)
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        alive = []  # scratch + operands of the side-stream launch stay referenced until the join
        dw = be.conv_backward_weight(features, grad_out, tbl_w, tuple(weight.shape), stream=side.cuda_stream,
                                     keep_alive=alive, operand=MFMA_OPERAND, **dup_kw)
    if need_dx:
        if inverse:
            dx = be.conv_backward_input(grad_out, weight, rb.pair_fwd, rb.n_out, mirror=False, order=rb.order_fwd,
                                        operand=MFMA_OPERAND)
        elif rb.kind == "subm":
            dx = be.conv_backward_input(grad_out, weight, rb.pair_fwd, rb.n_in, mirror=True, centre=rb.centre, rep=rb.rep,
                                        order=rb.order_bwd, operand=MFMA_OPERAND, group_ws=group_ws,
                                        sorted_rows=rb.sorted_rows, grp_plan=rb.grp_plan, grp=grp)
        else:
            dx = be.conv_backward_input(grad_out, weight, rb.pair_bwd, rb.n_in, mirror=False, order=rb.order_bwd,
                                        operand=MFMA_OPERAND)
    if side is not None:
        main.wait_stream(side)
        del alive
    elif need_dw:
        dw = be.conv_backward_weight(features, grad_out, tbl_w, tuple(weight.shape), operand=MFMA_OPERAND, **dup_kw)
    return dx, dw


def _as_wide_rows(g: torch.Tensor):
    """(tensor, first column) such that tensor[:, col : col + C] == g, WITHOUT copying when g is a column slice of a wider
    row-major matrix (what the backward of the NRConv concat hands to its two producers).  The BN backward kernels take a row
    stride and a column offset, so the `.contiguous()` copy of every such slice can be skipped."""
    if g.is_contiguous():
        return g, 0
    n, c = g.shape
    rs = g.stride(0)
    if (g.dim() == 2 and g.stride(1) == 1 and rs > c and rs % 4 == 0 and c % 4 == 0 and g.is_cuda):
        col0 = g.storage_offset() % rs
        start = g.storage_offset() - col0
        if col0 % 4 == 0 and col0 + c <= rs and g.untyped_storage().nbytes() >= (start + n * rs) * g.element_size():
            return torch.as_strided(g, (n, rs), (rs, 1), start), col0
    return g.contiguous(), 0


class ConvBNReLUFunction(torch.autograd.Function):
    """conv -> training-mode BatchNorm1d -> (ReLU) as ONE autograd node; on the HIP backend each direction is ONE C-ABI call
    (vc_post_act_block_forward / _backward: the unit of spconv_backbone.py:86-131), same kernels, same numerics."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, rb, inverse, running_mean, running_var, nbt, momentum, eps, relu):
        be = get_backend()
        tbl, order = (rb.pair_bwd, rb.order_bwd) if inverse else (rb.pair_fwd, rb.order_fwd)
        srt = rb.sorted_rows and not inverse
        ctx.unit_call = (FUSED_UNIT_CALLS and hasattr(be, "post_act_block_forward") and x.is_cuda and tbl.shape[1] > 0
                         and weight.shape[0] in (4, 8, 16, 32, 64, 128) and gamma is not None and beta is not None
                         and running_mean is not None and not OVERLAP_WEIGHT_GRAD)
        if ctx.unit_call:
            y, y_raw, mean, var = be.post_act_block_forward(x, weight, tbl, order, MFMA_OPERAND, srt, gamma, beta,
                                                            running_mean, running_var, nbt, momentum, eps, relu)
        elif FUSE_BN_STATS and MFMA_OPERAND == "f32" and be.conv_epilogue_supported(x.shape[0], weight.shape[-1],
                                                                                    weight.shape[0], rb.kv):
            # the conv epilogue emits the per-block (sum, sum of squares): the statistics need no pass over y_raw
            y_raw, partial = be.conv_forward_stats(x, weight, tbl, order=order, sorted_rows=srt)
            y, mean, var = be.bn_forward(y_raw, gamma, beta, running_mean, running_var, True, momentum, eps, relu,
                                         num_batches_tracked=nbt, partial=partial)
        else:
            y_raw = be.conv_forward(x, weight, tbl, order=order, operand=MFMA_OPERAND, sorted_rows=srt)
            y, mean, var = be.bn_forward(y_raw, gamma, beta, running_mean, running_var, True, momentum, eps, relu,
                                         num_batches_tracked=nbt)
        ctx.rb, ctx.inverse, ctx.cfg = rb, inverse, (float(eps), bool(relu))
        ctx.save_for_backward(x, weight, y_raw, mean, var, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        be = get_backend()
        x, weight, y_raw, mean, var, gamma, beta = ctx.saved_tensors
        eps, relu = ctx.cfg
        wide, col0 = _as_wide_rows(grad_out)
        rb, gws = ctx.rb, None
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if ctx.unit_call:
            inverse = ctx.inverse
            if inverse:        # inverse conv: forward walked pair_bwd, its transpose walks pair_fwd
                tbl_w, tbl_dx, n_dx, mirror, order_dx, rep, centre = rb.pair_bwd, rb.pair_fwd, rb.n_out, False, rb.order_fwd, None, -1
            elif rb.kind == "subm":
                tbl_w, tbl_dx, n_dx, mirror, order_dx, rep, centre = rb.pair_fwd, rb.pair_fwd, rb.n_in, True, rb.order_bwd, rb.rep, rb.centre
            else:
                tbl_w, tbl_dx, n_dx, mirror, order_dx, rep, centre = rb.pair_fwd, rb.pair_bwd, rb.n_in, False, rb.order_bwd, None, -1
            dx, dw, dgamma, dbeta = be.post_act_block_backward(
                x, weight, y_raw, wide, col0, mean, var, gamma, beta, eps, relu, tbl_w, tbl_dx, n_dx, mirror, centre, rep,
                rb.grp_plan if rep is not None else None, order_dx, MFMA_OPERAND, rb.sorted_rows and rb.kind == "subm" and not inverse, need_dx, need_dw)
            return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None, None
        d_raw, dgamma, dbeta = be.bn_backward(y_raw, wide, col0, mean, var, gamma, beta, eps, relu)
        dx, dw = _conv_backward(rb, ctx.inverse, x, weight, d_raw, need_dx, need_dw, group_ws=gws)
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None, None


def conv_bn_relu(x: torch.Tensor, weight: torch.Tensor, rb: "Rulebook", inverse: bool, bn: torch.nn.BatchNorm1d,
                 relu: bool) -> torch.Tensor:
    """Fused unit for the training path; callers fall back to sparse_conv + bn_relu otherwise."""
    nbt = bn.num_batches_tracked if bn.track_running_stats else None
    return ConvBNReLUFunction.apply(x, weight, bn.weight, bn.bias, rb, inverse, bn.running_mean, bn.running_var, nbt,
                                    bn.momentum, bn.eps, relu)


def conv_bn_relu_eval(x: torch.Tensor, weight: torch.Tensor, rb: "Rulebook", inverse: bool, bn: torch.nn.BatchNorm1d,
                      relu: bool) -> Optional[torch.Tensor]:
    """Inference: conv + BatchNorm1d(running statistics) (+ReLU) as ONE kernel launch, or None when not applicable (the
    caller then runs the two modules one after the other)."""
    be = get_backend()
    if not (FUSE_BN_EVAL and MFMA_OPERAND == "f32" and x.is_cuda and x.shape[0] != 0 and bn.track_running_stats
            and be.conv_epilogue_supported(x.shape[0], weight.shape[-1], weight.shape[0], rb.kv)):
        return None
    tbl, order = (rb.pair_bwd, rb.order_bwd) if inverse else (rb.pair_fwd, rb.order_fwd)
    return be.conv_forward_affine(x.detach(), weight.detach(), tbl, order, bn.running_mean, bn.running_var,
                                  None if bn.weight is None else bn.weight.detach(),
                                  None if bn.bias is None else bn.bias.detach(), bn.eps, relu,
                                  sorted_rows=rb.sorted_rows and not inverse)


# dW on a side stream under the backward-input conv.  Was worth 0.5 ms when the kernels left the chip half empty; since the
# kernel work of this round it only adds two stream joins per layer on the host: 6.61-6.63 ms without vs 6.85-7.15 ms with
# (three interleaved A/B pairs).  Off by default.
OVERLAP_WEIGHT_GRAD = os.environ.get("VIRCONV_OVERLAP_DW", "0") != "0"
# training: BN partial sums in the conv epilogue.  Measured A/B on the bench step: 7.21-7.24 ms with, 7.07 ms without -- the
# saved read-back pass (0.14 ms) is paid back by the epilogue's two extra barriers and a 10x longer finalize; off by default.
FUSE_BN_STATS = os.environ.get("VIRCONV_FUSE_BN_STATS", "0") != "0"
FUSE_BN_EVAL = os.environ.get("VIRCONV_FUSE_BN_EVAL", "1") != "0"     # inference: BN(+ReLU) folded into the conv store
# training: a conv+BN+ReLU unit is ONE C-ABI call per direction (vc_post_act_block_forward / _backward) instead of 4-9 calls
FUSED_UNIT_CALLS = os.environ.get("VIRCONV_FUSED_UNIT_CALLS", "1") != "0"
# vc_row_order permutations computed with the rulebooks (tile-homogeneity hint for the gather-GEMM; results identical).
#   "bwd"  (default) strided convs' backward-input tables only: their active sets are parity classes, sorting cuts the
#          issued work 2.3x (measured: s3.down bwd 197 -> 85 us) and a 1024-row window is enough
#   "strided" the strided convs' forward AND backward-input tables.  Round 3 made the sort of a many-mask table 20x cheaper
#          (1.4-2.8 ms -> 65-100 us: the mask-set overflow exits at once) and measured it: stand-alone the strided forwards gain
#          (s2/s3/s4.down 70/171/163 -> 51/140/147 us), inside the step the three extra sorts on the plan stream cost more GPU
#          time than the 65 us they save on the main stream: 5.66 vs 5.49 ms per step, same box (profiles/r03_bench_lines.txt)
#   "all"  every table (measured: strided forward -25 %, SubM +-0 -- the sort costs more than it saves there)
#   "none" natural order everywhere
# MFMA operand type of the conv kernels: "f32" (exact, default, the parity path) | "f16" | "bf16" (BASELINE configs[4]:
# "fp16 MFMA contraction"; tensors stay fp32, operands are rounded in registers, accumulation is fp32)
MFMA_OPERAND = os.environ.get("VIRCONV_MFMA_OPERAND", "f32")
# LDS row-window gather-GEMM for SubM convs on coordinate-sorted rows (VC_CONV_SORTED_ROWS hint).  Built, parity-tested and
# MEASURED SLOWER than the direct gathers on MI355X (s3.d3_conv1 64->32 forward 243-273 us vs 210 us: the per-wave windows cost
# LDS, LDS costs resident waves, and these kernels' throughput follows their occupancy -- profiles/r02_kbench_variants.txt,
# r02_pmc_conv_variants.md).  Off by default; "1" turns it on (tools/kbench.py measures both).
WINDOW_GATHER = os.environ.get("VIRCONV_WINDOW_GATHER", "0") != "0"
ROW_ORDER = os.environ.get("VIRCONV_ROW_ORDER", "bwd")
# duplicate-pixel (2-D) SubM tables: backward-input in representative-first row order (vc_rep_order).  Measured: no gain (2-D
# backward-input 316 vs 303 us per pass, train step 6.21 vs 6.17 ms) -- those launches are dominated by the group-sum machinery
# and the gathers, not by padded MFMAs.  Off by default.
REP_FIRST_ORDER = os.environ.get("VIRCONV_REP_FIRST_ORDER", "0") != "0"
ROW_ORDER_WINDOW = int(os.environ.get("VIRCONV_ROW_ORDER_WINDOW", "2048"))
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


_CONV_CHANNELS = (4, 8, 16, 32, 64)        # channel counts the gather-GEMM / weight-gradient kernels are instantiated for
_BN_CHANNELS = (4, 8, 16, 32, 64, 128)


def _channel_pieces(c: int, sizes):
    """[(first, last, padded)] covering range(c) with pieces of at most max(sizes) channels, each padded up to a supported size."""
    out, a, top = [], 0, max(sizes)
    while a < c:
        b = min(c, a + top)
        out.append((a, b, min(s for s in sizes if s >= b - a)))
        a = b
    return out


def sparse_conv(features: torch.Tensor, weight: torch.Tensor, rb: Rulebook, inverse: bool = False) -> torch.Tensor:
    """Sparse conv for ANY channel counts.  The kernels serve 4/8/16/32/64 channels on either side (every layer of the reference's
    models); other counts (e.g. NRConvBlock(conv_depth=True): +4 input channels, spconv_backbone.py:172-173) are tiled into
    supported pieces with zero padding -- same kernels, differentiable through the pad / slice / add ops around them."""
    cout, cin = weight.shape[0], weight.shape[-1]
    if (cin in _CONV_CHANNELS and cout in _CONV_CHANNELS) or not features.is_cuda:
        return SparseConvFunction.apply(features, weight, rb, inverse)
    F = torch.nn.functional
    outs = []
    for o0, o1, op in _channel_pieces(cout, _CONV_CHANNELS):
        acc = None
        for i0, i1, ip in _channel_pieces(cin, _CONV_CHANNELS):
            x = F.pad(features[:, i0:i1], (0, ip - (i1 - i0)))
            w = F.pad(weight[o0:o1, ..., i0:i1], (0, ip - (i1 - i0)) + (0, 0) * (weight.dim() - 2) + (0, op - (o1 - o0)))
            y = SparseConvFunction.apply(x.contiguous(), w.contiguous(), rb, inverse)[:, :o1 - o0]
            acc = y if acc is None else acc + y
        outs.append(acc)
    return outs[0] if len(outs) == 1 else torch.cat(outs, 1)


class GatherRowsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, keep):
        ctx.n_in = features.shape[0]
        ctx.save_for_backward(keep)
        fo, _ = get_backend().gather_rows(features, None, keep)
        return fo

    @staticmethod
    def backward(ctx, grad_out):
        (keep,) = ctx.saved_tensors
        return get_backend().scatter_rows(grad_out.contiguous(), keep, ctx.n_in), None


def discard_rows(features: torch.Tensor, indices: torch.Tensor, keep: torch.Tensor):
    """Device half of layer_voxel_discard (spconv_backbone.py:134-147): (features[keep], indices[keep])."""
    keep = keep.to(device=features.device, dtype=torch.int64)
    f = GatherRowsFunction.apply(features, keep)
    # indices carry no gradient; gather them with the same kernel through a 4-channel float view is not possible
    # (int payload), so use the index half of vc_gather_rows on a detached call
    _, idx = get_backend().gather_rows(features.detach(), indices, keep)
    return f, idx


class ToDenseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, spatial_shape, batch_size, pad=(0, 0)):
        ctx.meta = (tuple(spatial_shape), int(batch_size), (int(pad[0]), int(pad[1])))
        ctx.save_for_backward(indices)
        if pad[0] or pad[1]:
            return get_backend().to_dense(features, indices, spatial_shape, batch_size, pad=ctx.meta[2])
        return get_backend().to_dense(features, indices, spatial_shape, batch_size)

    @staticmethod
    def backward(ctx, grad_dense):
        (indices,) = ctx.saved_tensors
        shape, bs, pad = ctx.meta
        kw = {"pad": pad} if (pad[0] or pad[1]) else {}
        if grad_dense.dim() >= 3 and grad_dense.shape[0] > 1 and grad_dense.stride(0) == 0:
            # a gradient that is the same for every sample (broadcast along the batch axis): gather from ONE sample's planes
            # instead of materialising batch_size copies of them
            idx0 = indices.clone()
            idx0[:, 0] = 0
            return get_backend().from_dense(grad_dense[:1].contiguous(), idx0, shape, 1, **kw), None, None, None, None
        return get_backend().from_dense(grad_dense.contiguous(), indices, shape, bs, **kw), None, None, None, None


def to_dense(features, indices, spatial_shape, batch_size, pad=(0, 0)):
    """(B, C, *spatial) scatter; `pad` = (pad_h, pad_w) adds a zero border around the last two axes in the same pass."""
    return ToDenseFunction.apply(features, indices, tuple(int(s) for s in spatial_shape), batch_size, (int(pad[0]), int(pad[1])))


class WeightedSumFunction(torch.autograd.Function):
    """sum(x * g) with g broadcast over the leading axis of x, as one pass over x (vc_weighted_sum) instead of an elementwise product
    plus a reduction.  The gradient for x is gout * g: a stride-0 view over the leading axis for a dense (B, ...) map (to_dense's
    backward gathers it from one sample's planes), materialised rows for an (N, C) feature matrix (what the feature pass takes)."""

    @staticmethod
    def forward(ctx, x, g):
        ctx.save_for_backward(g)
        ctx.xshape = tuple(x.shape)
        return get_backend().weighted_sum(x, g)

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        shp = ctx.xshape
        be = get_backend()
        if len(shp) == 2:
            return be.weighted_sum_backward(gout.reshape(1), g, shp[0]), None
        row = be.weighted_sum_backward(gout.reshape(1), g, 1)
        return row.view((1,) + shp[1:]).expand(shp), None


def weighted_sum(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """sum(x * g), g broadcast over x's leading axis.  One fused, deterministic pass on the HIP backend; plain tensor ops otherwise
    (the CPU oracle)."""
    be = get_backend()
    e = x.numel() // max(x.shape[0], 1)
    if hasattr(be, "weighted_sum") and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and e % 4 == 0 and x.shape[0] >= 1 \
            and not g.requires_grad and (e <= 1024 and (e & (e - 1)) == 0 or x.shape[0] <= 65535):
        return WeightedSumFunction.apply(x, g)
    return (x * g).sum()


class BNReLUFunction(torch.autograd.Function):
    """Training-mode BatchNorm1d(+ReLU) over the active rows."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, nbt=None):
        be = get_backend()
        y, mean, var = be.bn_forward(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu,
                                     num_batches_tracked=nbt)
        ctx.cfg = (training, float(eps), bool(relu))
        ctx.save_for_backward(x, mean, var, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        be = get_backend()
        x, mean, var, gamma, beta = ctx.saved_tensors
        training, eps, relu = ctx.cfg
        if not training:
            # frozen statistics (fine-tuning with BatchNorm in eval mode, input-gradient probes): mean / var are constants, so
            # dx = dy * gamma / sqrt(var + eps) behind the ReLU mask; rare path, plain tensor ops
            istd = torch.rsqrt(var + eps)
            xh = (x - mean) * istd
            g = grad_out
            if relu:
                g = torch.where(xh * gamma + beta > 0, g, torch.zeros_like(g))
            dgamma = (g * xh).sum(0) if ctx.needs_input_grad[1] else None
            dbeta = g.sum(0) if ctx.needs_input_grad[2] else None
            return g * (gamma * istd), dgamma, dbeta, None, None, None, None, None, None, None
        dx, dgamma, dbeta = be.bn_backward(x, grad_out.contiguous(), 0, mean, var, gamma, beta, eps, relu)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def bn_relu(x: torch.Tensor, bn: torch.nn.BatchNorm1d, relu: bool) -> torch.Tensor:
    """Fused BatchNorm1d(+ReLU) using the parameters/buffers of a stock nn.BatchNorm1d module."""
    training = bn.training or not bn.track_running_stats
    momentum = 0.0 if bn.momentum is None else bn.momentum
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    # num_batches_tracked is incremented inside the stats-finalize kernel (no extra launch)
    nbt = bn.num_batches_tracked if (training and bn.track_running_stats) else None
    c = x.shape[1]
    if c in _BN_CHANNELS or not x.is_cuda or bn.weight is None:
        return BNReLUFunction.apply(x, bn.weight, bn.bias, rm, rv, training, momentum, bn.eps, relu, nbt)
    # any other channel count: BatchNorm is per channel, so it is applied piecewise on zero-padded channel slices (a padded
    # channel has mean 0 / variance 0 and is cut off again); running statistics are updated on padded copies and written back
    F = torch.nn.functional
    outs = []
    for k, (a, b, p_) in enumerate(_channel_pieces(c, _BN_CHANNELS)):
        pad = (0, p_ - (b - a))
        rm_p = F.pad(rm[a:b], pad) if rm is not None else None
        rv_p = F.pad(rv[a:b], pad, value=1.0) if rv is not None else None
        y = BNReLUFunction.apply(F.pad(x[:, a:b], pad).contiguous(), F.pad(bn.weight[a:b], pad), F.pad(bn.bias[a:b], pad), rm_p,
                                 rv_p, training, momentum, bn.eps, relu, nbt if k == 0 else None)
        if training and rm is not None:
            with torch.no_grad():
                rm[a:b].copy_(rm_p[:b - a])
                rv[a:b].copy_(rv_p[:b - a])
        outs.append(y[:, :b - a])
    return outs[0] if len(outs) == 1 else torch.cat(outs, 1)


def project_uv(indices, calib, trans, batch_size, stride):
    uv, _ = get_backend().project_uv(indices, calib, trans, batch_size, stride)
    return uv


def calib_tensor(calibs, device) -> torch.Tensor:
    """Pack per-sample calibration (objects with .V2C/.R0/.P2 or dicts) into the (B, 33) float32 block of the ABI."""
    rows = []
    for c in calibs:
        if isinstance(c, dict):
            v2c, r0, p2 = c["Tr_velo2cam"], c["R0"], c["P2"]
        else:
            v2c, r0, p2 = c.V2C, c.R0, c.P2
        rows.append(np.concatenate([np.asarray(v2c, np.float32).reshape(12), np.asarray(r0, np.float32).reshape(9),
                                    np.asarray(p2, np.float32).reshape(12)]))
    return torch.from_numpy(np.stack(rows)).to(device)
