"""The feature pass of a backbone as ONE native call per direction (vc_pass_forward / vc_pass_backward).

`VirConvL8x.forward` (pcdet/models/backbones_3d/spconv_backbone.py:609-699) is, once its geometry plan exists, a fixed chain of
post_act_block units (:86-131), NRConvBlock channel concats (:207-229) and layer discards (:134-147).  Issued node by node from
Python that chain costs ~25 us of host time per launch (autograd node, ctypes marshalling, allocator round trips) against ~4 us
for the launch itself, and a train step is ~370 launches: the host, not the GPU, bounds the step.  Here the chain is described
ONCE per model as a small program over numbered buffers; per step only the row counts, rulebook pointers and keep indices are
filled in, and the library runs the whole forward (and, from ONE autograd node, the whole reverse sweep) out of one arena per
direction.  Same kernels, same order, bit-identical results (tests/test_ops_gpu.py::test_native_pass_*).

Parameters stay ordinary autograd inputs of the node (their gradients are returned as views of one flat buffer), so
`DistributedDataParallel` hooks, `torch.autograd.grad`, frozen parameters and gradient accumulation behave as with the
node-by-node path.  Anything the program does not cover (eval mode with gradients, non-standard units, empty tensors, the CPU
oracle backend used by host-logic tests) falls back to that path: `usable()` decides, per call.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional

import torch

from . import _lib, ops

NATIVE_PASS = os.environ.get("VIRCONV_NATIVE_PASS", "1") != "0"
NATIVE_PASS_EVAL = os.environ.get("VIRCONV_NATIVE_PASS_EVAL", "0") != "0"
_CHANNELS = (4, 8, 16, 32, 64)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class _Program:
    """Static part of a pass: ops, buffer widths, which module feeds which unit, which plan entry feeds which table."""

    def __init__(self):
        self.ops: List[tuple] = []        # (kind, src, dst, dst_col0, unit, table, keep, relu)
        self.cols: List[int] = []         # per buffer
        self.rows_of: List[tuple] = []    # per buffer: ("in",) | ("table_out", t) | ("keep", k)
        self.units: List[tuple] = []      # (conv module, bn module)
        self.n_tables = 0
        self.n_keeps = 0
        self.outputs: List[int] = []      # buffer ids handed back to the caller

    def buf(self, cols: int, rows_of: tuple) -> int:
        self.cols.append(int(cols))
        self.rows_of.append(rows_of)
        return len(self.cols) - 1

    def unit(self, seq) -> int:
        self.units.append((seq[0], seq[1]))
        return len(self.units) - 1

    def table(self) -> int:
        self.n_tables += 1
        return self.n_tables - 1

    def freeze(self):
        n = len(self.ops)
        self.c_ops = (_lib.PassOp * n)()
        for i, o in enumerate(self.ops):
            (self.c_ops[i].kind, self.c_ops[i].src, self.c_ops[i].dst, self.c_ops[i].dst_col0, self.c_ops[i].unit,
             self.c_ops[i].table, self.c_ops[i].keep, self.c_ops[i].relu) = o
        # flat gradient layout: per unit dweight | dgamma | dbeta
        self.grad_sizes, self.grad_shapes = [], []
        for conv, bn in self.units:
            self.grad_sizes += [conv.weight.numel(), bn.weight.numel(), bn.bias.numel()]
            self.grad_shapes += [tuple(conv.weight.shape), None, None]
        self.grad_offsets = [0]
        for s in self.grad_sizes:
            self.grad_offsets.append(self.grad_offsets[-1] + ((s + 63) // 64) * 64)  # 256-byte aligned slots
        self.grad_total = self.grad_offsets[-1]
        self.unit_table = {o[4]: o[5] for o in self.ops if o[0] == _lib.PASS_UNIT}
        return self


def unit_is_plain(seq) -> bool:
    """conv (no bias) -> BatchNorm1d -> ReLU with the channel counts the kernels serve: the only shape post_act_block builds."""
    from .backbone import NRConvBlock
    if not NRConvBlock._unit_is_plain(seq):
        return False
    w = seq[0].weight
    return w.shape[0] in _CHANNELS and w.shape[-1] in _CHANNELS and not getattr(seq[0], "inverse", False)


def build_virconv_l_program(model, discard_active: bool, training: bool) -> _Program:
    """VirConvL8x: four NRConvBlocks (+ the layer discard after the first three in training) and conv_out.  Training: the
    second 2-D unit writes its half of the channel concat in place.  Eval: every unit is ONE launch (BatchNorm folded into the
    conv store, which writes dense rows), so both halves are copied into the concat."""
    P = _Program()
    cur = P.buf(model.vir_conv1.d3_conv1[0].weight.shape[-1], ("in",))
    P.plan_tables = []   # per table: (stage index | "conv_out", key kind)
    P.plan_keeps = []    # per keep: stage index
    for bi, blk in enumerate([model.vir_conv1, model.vir_conv2, model.vir_conv3, model.vir_conv4]):
        if blk.stride > 1:
            t = P.table()
            P.plan_tables.append((bi, "down"))
            u = P.unit(blk.down_layer)
            d = P.buf(blk.down_layer[0].weight.shape[0], ("table_out", t))
            P.ops.append((_lib.PASS_UNIT, cur, d, 0, u, t, 0, 1))
            cur = d
        t3, t2 = P.table(), P.table()
        P.plan_tables += [(bi, "3d"), (bi, "2d")]
        c = blk.d3_conv1[0].weight.shape[0]
        f1 = P.buf(c, ("table_out", t3))
        f3 = P.buf(c, ("table_out", t3))
        g1 = P.buf(c, ("table_out", t3))
        cat = P.buf(2 * c, ("table_out", t3))
        P.ops.append((_lib.PASS_UNIT, cur, f1, 0, P.unit(blk.d3_conv1), t3, 0, 1))
        P.ops.append((_lib.PASS_UNIT, f1, f3, 0, P.unit(blk.d3_conv2), t3, 0, 1))
        # the first 2-D unit reads f3 BEFORE f3 is copied into the concat: in the reverse sweep the concat slice then arrives first
        # and the unit's backward-input conv -- the last contributor to f3's gradient -- adds it in its epilogue
        P.ops.append((_lib.PASS_UNIT, f3, g1, 0, P.unit(blk.d2_conv1), t2, 0, 1))
        P.ops.append((_lib.PASS_COPY, f3, cat, 0, 0, 0, 0, 0))
        if training:
            P.ops.append((_lib.PASS_UNIT, g1, cat, c, P.unit(blk.d2_conv2), t2, 0, 1))      # concat written in place
        else:
            g2 = P.buf(c, ("table_out", t3))
            P.ops.append((_lib.PASS_UNIT, g1, g2, 0, P.unit(blk.d2_conv2), t2, 0, 1))
            P.ops.append((_lib.PASS_COPY, g2, cat, c, 0, 0, 0, 0))
        cur = cat
        if discard_active and bi < 3:
            k = P.n_keeps
            P.n_keeps += 1
            P.plan_keeps.append(bi)
            kept = P.buf(2 * c, ("keep", k))
            P.ops.append((_lib.PASS_GATHER, cat, kept, 0, 0, 0, k, 0))
            cur = kept
        P.outputs.append(cur)
    t = P.table()
    P.plan_tables.append(("conv_out", None))
    out = P.buf(model.conv_out[0].weight.shape[0], ("table_out", t))
    P.ops.append((_lib.PASS_UNIT, cur, out, 0, P.unit(model.conv_out), t, 0, 1))
    P.outputs.append(out)
    return P.freeze()


def _plan_table(model, plan, entry):
    where, kind = entry
    if where == "conv_out":
        return plan["conv_out"][model.conv_out[0].indice_key]
    blk = [model.vir_conv1, model.vir_conv2, model.vir_conv3, model.vir_conv4][where]
    kd, k3, k2 = blk._keys()
    st = plan["stages"][where]
    return st["rb3d"][kd] if kind == "down" else (st["rb3d"][k3] if kind == "3d" else st["rb2d"][k2])


def usable(model, feats: torch.Tensor, plan) -> bool:
    be = ops.get_backend()
    if not (NATIVE_PASS and ops.FUSED_UNIT_CALLS and not ops.OVERLAP_WEIGHT_GRAD and getattr(be, "native_pass", False)
            and feats.is_cuda and feats.dtype == torch.float32 and feats.shape[0] > 0):
        return False
    seqs = [model.conv_out]
    for blk in (model.vir_conv1, model.vir_conv2, model.vir_conv3, model.vir_conv4):
        seqs += ([blk.down_layer] if blk.stride > 1 else []) + [blk.d3_conv1, blk.d3_conv2, blk.d2_conv1, blk.d2_conv2]
        if blk.conv_depth:
            return False
    if not all(unit_is_plain(s) and s[1].training == model.training for s in seqs):
        return False
    if not model.training:
        # eval: the node-by-node path is already ONE launch per unit (BatchNorm folded into the conv store) and at bs 1 the step is
        # bound by the geometry plan's count reads, not by the feature pass: measured 1.25 ms/frame (nodes) vs 1.31 ms (native
        # pass, whose burst of launches competes with the next frame's plan kernels).  The native eval pass stays available
        # (VIRCONV_NATIVE_PASS_EVAL=1, bit-equal, tested) but is not the default.
        if not NATIVE_PASS_EVAL:
            return False
        if torch.is_grad_enabled() and (feats.requires_grad or any(p.requires_grad for p in model.parameters())):
            return False  # running statistics + gradients: the node-by-node path (eval-mode BatchNorm backward)
    return True


_PROGRAMS = weakref.WeakKeyDictionary()   # model -> {(discard_active, training): _Program}; kept off the module (deepcopy / pickle safe)


class _Call:
    """Everything one forward call filled in (kept alive for the backward of the same call)."""
    __slots__ = ("prog", "c_prog", "c_bufs", "c_units", "c_tables", "c_keeps", "arena", "offsets", "keep_alive", "group_bytes")


def _fill(model, P: _Program, feats, plan, training: bool) -> Optional[_Call]:
    call = _Call()
    call.prog = P
    tables = [_plan_table(model, plan, e) for e in P.plan_tables]
    keeps = [plan["stages"][bi]["keep"] for bi in P.plan_keeps]
    if any(rb.n_out <= 0 or rb.n_in <= 0 for rb in tables) or any(k is None or k.shape[0] == 0 for k in keeps):
        return None
    c_tables = (_lib.PassTable * len(tables))()
    group_bytes = 0
    for i, rb in enumerate(tables):
        t = c_tables[i]
        subm = rb.kind == "subm"
        kv = rb.pair_fwd.shape[0]
        t.pair_fwd, t.pair_bwd, t.rep = rb.pair_fwd.data_ptr(), _ptr(rb.pair_bwd), _ptr(rb.rep)
        t.order_fwd, t.order_bwd = _ptr(rb.order_fwd), _ptr(rb.order_bwd)
        t.n_in, t.n_out, t.kv, t.subm = rb.n_in, rb.n_out, kv, 1 if subm else 0
        t.centre = kv // 2 if subm else -1     # odd kernel sizes: the centre tap is the middle offset
        t.sorted_rows = 1 if rb.sorted_rows else 0
    c_keeps = (C.c_void_p * max(len(keeps), 1))()
    for i, k in enumerate(keeps):
        assert k.dtype == torch.int64 and k.is_contiguous()
        c_keeps[i] = k.data_ptr()
    c_bufs = (_lib.PassBuf * len(P.cols))()
    for i, (cols, ro) in enumerate(zip(P.cols, P.rows_of)):
        b = c_bufs[i]
        b.cols = cols
        if ro[0] == "in":
            b.rows, b.external, b.ptr = feats.shape[0], 1, feats.data_ptr()
        elif ro[0] == "table_out":
            b.rows = tables[ro[1]].n_out
        else:
            b.rows = keeps[ro[1]].shape[0]
    c_units = (_lib.PassUnit * len(P.units))()
    for i, (conv, bn) in enumerate(P.units):
        u = c_units[i]
        w = conv.weight
        u.weight, u.gamma, u.beta = w.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr()
        u.running_mean, u.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        u.num_batches_tracked = _ptr(bn.num_batches_tracked)
        u.cin, u.cout, u.momentum, u.eps = w.shape[-1], w.shape[0], bn.momentum, bn.eps
        rb = tables[P.unit_table[i]]
        if rb.rep is not None:   # duplicate-pixel table: its backward needs the persistent group-sum accumulator
            group_bytes = max(group_bytes, rb.n_out * w.shape[0] * 8 + 64)
    c_prog = _lib.PassProgram()
    c_prog.ops, c_prog.n_ops = P.c_ops, len(P.ops)
    c_prog.bufs, c_prog.n_bufs = c_bufs, len(P.cols)
    c_prog.units, c_prog.n_units = c_units, len(P.units)
    c_prog.tables, c_prog.n_tables = c_tables, len(tables)
    c_prog.keeps, c_prog.n_keeps = c_keeps, len(keeps)
    c_prog.training, c_prog.operand_type = 1 if training else 0, _lib.OPERAND_TYPES[ops.MFMA_OPERAND]
    call.c_prog, call.c_bufs, call.c_units, call.c_tables, call.c_keeps = c_prog, c_bufs, c_units, c_tables, c_keeps
    call.keep_alive = (tables, keeps, feats)
    call.group_bytes = group_bytes
    return call


class PassFunction(torch.autograd.Function):
    """forward(feats, call, *params) -> the program's output buffers; ONE autograd node for the whole backbone."""

    @staticmethod
    def forward(ctx, feats, call: _Call, *params):
        be = ops.get_backend()
        lib = be.lib
        P = call.prog
        prog = C.byref(call.c_prog)
        nbytes = lib.vc_pass_forward_arena_bytes(prog)
        if nbytes == 0:
            _lib.check(_lib.VC_EINVAL, "vc_pass_forward_arena_bytes")
        arena = torch.empty((nbytes,), dtype=torch.uint8, device=feats.device)
        offsets = (C.c_int64 * len(P.cols))()
        _lib.check(lib.vc_pass_forward(prog, arena.data_ptr(), nbytes, offsets, be.stream()), "vc_pass_forward")
        call.arena, call.offsets = arena, offsets
        ctx.call = call
        outs = []
        for b in P.outputs:
            rows, cols = call.c_bufs[b].rows, P.cols[b]
            outs.append(arena[offsets[b]: offsets[b] + rows * cols * 4].view(torch.float32).view(rows, cols))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        be = ops.get_backend()
        lib = be.lib
        call = ctx.call
        P = call.prog
        dev = call.arena.device
        ext = (C.c_void_p * len(P.cols))()
        hold = []
        for b, g in zip(P.outputs, gouts):
            if g is not None:
                g = g.contiguous()
                assert g.dtype == torch.float32 and g.shape == (call.c_bufs[b].rows, P.cols[b])
                hold.append(g)
                ext[b] = g.data_ptr()
        flat = torch.empty((P.grad_total,), dtype=torch.float32, device=dev)
        base = flat.data_ptr()
        need = ctx.needs_input_grad
        for i in range(len(P.units)):
            u = call.c_units[i]
            o = P.grad_offsets
            u.dweight = base + 4 * o[3 * i] if need[2 + 3 * i] else None
            u.dgamma = base + 4 * o[3 * i + 1] if need[2 + 3 * i + 1] else None
            u.dbeta = base + 4 * o[3 * i + 2] if need[2 + 3 * i + 2] else None
        want_in = bool(need[0])
        feats = call.keep_alive[2]
        gin = torch.empty_like(feats) if want_in else None
        prog = C.byref(call.c_prog)
        nbytes = lib.vc_pass_backward_arena_bytes(prog, ext, 1 if want_in else 0)
        if nbytes == 0:
            _lib.check(_lib.VC_EINVAL, "vc_pass_backward_arena_bytes")
        arena = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        gacc = be._group_acc(call.group_bytes, dev) if call.group_bytes else None
        side = be._side_stream(dev) if be.pass_overlap_dw() else None
        _lib.check(lib.vc_pass_backward(prog, call.arena.data_ptr(), call.arena.numel(), ext, _ptr(gin), _ptr(gacc),
                                        gacc.numel() if gacc is not None else 0, arena.data_ptr(), nbytes, side, be.stream()),
                   "vc_pass_backward")
        grads = []
        for i in range(len(P.grad_sizes)):
            if need[2 + i]:
                g = flat[P.grad_offsets[i]: P.grad_offsets[i] + P.grad_sizes[i]]
                grads.append(g.view(P.grad_shapes[i]) if P.grad_shapes[i] is not None else g)
            else:
                grads.append(None)
        ctx.call = None
        return (gin, None, *grads)


def run(model, feats: torch.Tensor, plan):
    """-> [x_conv1, x_conv2, x_conv3, x_conv4, out] feature matrices, or None when this call cannot take the native pass."""
    discard = model._discard_active()
    cache = _PROGRAMS.setdefault(model, {})
    key = (discard, bool(model.training))
    P = cache.get(key)
    if P is None:
        P = cache[key] = build_virconv_l_program(model, discard, bool(model.training))
    call = _fill(model, P, feats, plan, model.training)
    if call is None:
        return None
    params = []
    for conv, bn in P.units:
        params += [conv.weight, bn.weight, bn.bias]
    return PassFunction.apply(feats, call, *params)
