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
NATIVE_PASS_EVAL = os.environ.get("VIRCONV_NATIVE_PASS_EVAL", "1") != "0"
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
        self.table_getters: List = []     # per table: ctx -> Rulebook   (ctx = whatever the model hands to run_program)
        self.keep_getters: List = []      # per keep:  ctx -> int64 row indices
        self.outputs: List[int] = []      # buffer ids handed back to the caller

    def buf(self, cols: int, rows_of: tuple) -> int:
        self.cols.append(int(cols))
        self.rows_of.append(rows_of)
        return len(self.cols) - 1

    def unit(self, seq) -> int:
        self.units.append((seq[0], seq[1]))
        return len(self.units) - 1

    def table(self, getter) -> int:
        self.table_getters.append(getter)
        return len(self.table_getters) - 1

    def keep(self, getter) -> int:
        self.keep_getters.append(getter)
        return len(self.keep_getters) - 1

    def add_unit(self, seq, src: int, table: int, dst: Optional[int] = None, dst_col0: int = 0) -> int:
        """Append one post_act_block; -> its output buffer (a new dense one unless `dst` is given)."""
        if dst is None:
            dst = self.buf(seq[0].weight.shape[0], ("table_out", table))
        self.ops.append((_lib.PASS_UNIT, src, dst, dst_col0, self.unit(seq), table, 0, 1))
        return dst

    def add_gather(self, src: int, keep: int) -> int:
        dst = self.buf(self.cols[src], ("keep", keep))
        self.ops.append((_lib.PASS_GATHER, src, dst, 0, 0, 0, keep, 0))
        return dst

    def add_nrconv_chain(self, cur: int, blocks, discard_flags, training: bool, stage_of) -> int:
        """The chain of NRConvBlocks (spconv_backbone.py:150-229) + the layer discard after a block whose flag is set; every
        block's (post-discard) output is appended to `outputs`.  `stage_of(ctx, bi)` -> the block's plan dict."""
        for bi, (blk, flag) in enumerate(zip(blocks, discard_flags)):
            kd, k3, k2 = blk._keys()
            if blk.stride > 1:
                cur = self.add_unit(blk.down_layer, cur, self.table(lambda c, bi=bi, kd=kd: stage_of(c, bi)["rb3d"][kd]))
            t3 = self.table(lambda c, bi=bi, k3=k3: stage_of(c, bi)["rb3d"][k3])
            t2 = self.table(lambda c, bi=bi, k2=k2: stage_of(c, bi)["rb2d"][k2])
            ch = blk.d3_conv1[0].weight.shape[0]
            f3 = self.add_unit(blk.d3_conv2, self.add_unit(blk.d3_conv1, cur, t3), t3)
            cat = self.buf(2 * ch, ("table_out", t3))
            # the first 2-D unit reads f3 BEFORE f3 is copied into the concat: in the reverse sweep the concat slice then arrives
            # first and the unit's backward-input conv -- the last contributor to f3's gradient -- adds it in its epilogue
            g1 = self.add_unit(blk.d2_conv1, f3, t2, self.buf(ch, ("table_out", t3)))
            self.ops.append((_lib.PASS_COPY, f3, cat, 0, 0, 0, 0, 0))
            if training:
                self.add_unit(blk.d2_conv2, g1, t2, cat, ch)                      # concat written in place
            else:   # eval: one launch per unit (BatchNorm folded into the conv store, dense rows), both halves are copied
                g2 = self.add_unit(blk.d2_conv2, g1, t2, self.buf(ch, ("table_out", t3)))
                self.ops.append((_lib.PASS_COPY, g2, cat, ch, 0, 0, 0, 0))
            cur = cat
            if flag:
                cur = self.add_gather(cat, self.keep(lambda c, bi=bi: stage_of(c, bi)["keep"]))
            self.outputs.append(cur)
        return cur

    def flat(self):
        """The flat parameter of this program's units (flatten_parameters) while it is still what the units' parameters alias, else None."""
        fp = getattr(self, "_flat", None)
        if fp is None:
            return None
        base, offs = fp.data_ptr(), self.grad_offsets
        cache = self.__dict__.get("_unit_cache")
        if cache is not None and all(h is not None for h in cache[1]):
            # the addresses _unit_structs read for THIS call (it runs first and re-reads every tensor's address on every call)
            for i, (_, ptrs) in enumerate(cache[1]):
                if ptrs[0] != base + 4 * offs[3 * i] or ptrs[1] != base + 4 * offs[3 * i + 1] or ptrs[2] != base + 4 * offs[3 * i + 2]:
                    return None
            return fp
        for i, (conv, bn) in enumerate(self.units):      # someone re-assigned a parameter (load_state_dict(assign=True), .to(), ...)
            if (conv.weight.data_ptr() != base + 4 * offs[3 * i] or bn.weight.data_ptr() != base + 4 * offs[3 * i + 1]
                    or bn.bias.data_ptr() != base + 4 * offs[3 * i + 2]):
                return None
        return fp

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
    conv store, which writes dense rows), so both halves are copied into the concat.  ctx = the geometry plan."""
    P = _Program()
    cur = P.buf(model.vir_conv1.d3_conv1[0].weight.shape[-1], ("in",))
    blocks = [model.vir_conv1, model.vir_conv2, model.vir_conv3, model.vir_conv4]
    cur = P.add_nrconv_chain(cur, blocks, [discard_active and bi < 3 for bi in range(4)], training,
                             lambda plan, bi: plan["stages"][bi])
    key = model.conv_out[0].indice_key
    P.outputs.append(P.add_unit(model.conv_out, cur, P.table(lambda plan: plan["conv_out"][key])))
    return P.freeze()


def build_virconv8x_lidar_program(model) -> _Program:
    """VirConv8x LiDAR stream (spconv_backbone.py:362-407): conv_input, conv1..conv4 (a strided unit + two SubM units sharing
    one rulebook per stage), conv_out.  Outputs x_conv1..4 and the encoded tensor.  ctx = {indice_key: Rulebook}."""
    P = _Program()
    cur = P.buf(model.conv_input[0].weight.shape[-1], ("in",))
    tables = {}

    def tbl(seq):
        key = seq[0].indice_key
        if key not in tables:
            tables[key] = P.table(lambda rbs, key=key: rbs[key])
        return tables[key]

    cur = P.add_unit(model.conv_input, cur, tbl(model.conv_input))
    for stage in (model.conv1, model.conv2, model.conv3, model.conv4):
        for unit in stage:
            cur = P.add_unit(unit, cur, tbl(unit))
        P.outputs.append(cur)
    P.outputs.append(P.add_unit(model.conv_out, cur, tbl(model.conv_out)))
    return P.freeze()


def build_virconv8x_mm_program(model, discard_active: bool, training: bool = True) -> _Program:
    """VirConv8x virtual-point stream (spconv_backbone.py:444-535): the input discard (:488-489) and four NRConvBlocks with the
    layer discard after the first three.  ctx = plan["mm"][rid]."""
    P = _Program()
    cur = P.buf(model.vir_conv1.d3_conv1[0].weight.shape[-1], ("in",))
    if discard_active:
        cur = P.add_gather(cur, P.keep(lambda pm: pm["keep0"]))
    blocks = [model.vir_conv1, model.vir_conv2, model.vir_conv3, model.vir_conv4]
    P.add_nrconv_chain(cur, blocks, [discard_active and bi < 3 for bi in range(4)], training, lambda pm, bi: pm["stages"][bi])
    return P.freeze()


_UNITS_OK = weakref.WeakKeyDictionary()   # model -> {key: (fingerprint, verdict)}


def _units_fingerprint(seqs):
    """What the verdict of `_units_ok` depends on, read without a function call per module: the identity of each unit's three modules
    (a swapped BatchNorm -- convert_sync_batchnorm -- changes it) and the BatchNorm flags that can be flipped in place."""
    fp = []
    for s in seqs:
        m = s._modules
        conv, bn = m.get("0"), m.get("1")
        d = bn.__dict__ if bn is not None else {}
        w = conv._parameters.get("weight") if conv is not None else None
        fp.append((id(conv), id(w), id(bn), id(m.get("2")), len(m), d.get("training"), d.get("momentum"), d.get("affine"),
                   d.get("track_running_stats")))
    return tuple(fp)


def _units_ok(seqs, training: bool, model=None, key=None) -> bool:
    """Every unit is the plain conv -> BatchNorm1d -> ReLU triple the kernels serve, in the given mode.  With `model`: the verdict is
    kept per (model, key) and re-derived only when the fingerprint of the units changes (34 checks of ~4 us each per VirConv8x
    step otherwise -- host time of configurations that are host-bound)."""
    if model is None:
        return all(unit_is_plain(s) and s[1].training == training for s in seqs)
    fp = _units_fingerprint(seqs)
    slot = _UNITS_OK.setdefault(model, {})
    hit = slot.get((key, training))
    if hit is not None and hit[0] == fp:
        return hit[1]
    verdict = all(unit_is_plain(s) and s[1].training == training for s in seqs)
    slot[(key, training)] = (fp, verdict)
    return verdict


def _backend_ok(feats: torch.Tensor) -> bool:
    be = ops.get_backend()
    return bool(NATIVE_PASS and ops.FUSED_UNIT_CALLS and not ops.OVERLAP_WEIGHT_GRAD and getattr(be, "native_pass", False)
                and feats.is_cuda and feats.dtype == torch.float32 and feats.shape[0] > 0)


def _nrconv_seqs(blocks):
    seqs = []
    for blk in blocks:
        if blk.conv_depth:
            return None
        seqs += ([blk.down_layer] if blk.stride > 1 else []) + [blk.d3_conv1, blk.d3_conv2, blk.d2_conv1, blk.d2_conv2]
    return seqs


def usable(model, feats: torch.Tensor, plan) -> bool:
    """VirConvL8x: may this call take the native pass?"""
    if not _backend_ok(feats):
        return False
    seqs = _nrconv_seqs([model.vir_conv1, model.vir_conv2, model.vir_conv3, model.vir_conv4])
    if seqs is None or not _units_ok(seqs + [model.conv_out], model.training, model, "L"):
        return False
    if not model.training:
        # eval: the node-by-node path is already ONE launch per unit (BatchNorm folded into the conv store); while the geometry plan
        # still read four counts per frame the native pass lost to it (1.31 vs 1.25 ms/frame at bs 1: its burst of launches
        # competed with the next frame's plan kernels).  With ONE count read per frame (ops.build_sparse_rulebook_chain) it wins:
        # 1.22 vs 1.30 ms/frame at bs 1, 1.77 vs 1.92 ms per 4 frames.  Bit-equal to the node path (tested);
        # VIRCONV_NATIVE_PASS_EVAL=0 selects the node path.
        if not NATIVE_PASS_EVAL:
            return False
        if torch.is_grad_enabled() and (feats.requires_grad or any(p.requires_grad for p in model.parameters())):
            return False  # running statistics + gradients: the node-by-node path (eval-mode BatchNorm backward)
    return True


def usable_8x(model, feats: torch.Tensor, stream: str) -> bool:
    """VirConv8x: the LiDAR stream ("lidar") or the virtual-point stream ("mm"); training, and (round 6) eval without gradients -- the
    test-time path over the x-concatenated tensor (spconv_backbone.py:409-442)."""
    if not _backend_ok(feats):
        return False
    if stream == "lidar":
        seqs = [model.conv_input, model.conv_out] + [u for st in (model.conv1, model.conv2, model.conv3, model.conv4) for u in st]
    else:
        seqs = _nrconv_seqs([model.vir_conv1, model.vir_conv2, model.vir_conv3, model.vir_conv4])
    if seqs is None or not _units_ok(seqs, model.training, model, stream):
        return False
    if not model.training:
        if not NATIVE_PASS_EVAL:
            return False
        if torch.is_grad_enabled() and (feats.requires_grad or any(p.requires_grad for p in model.parameters())):
            return False  # running statistics + gradients: the node-by-node path
    return True


_PROGRAMS = weakref.WeakKeyDictionary()   # model -> {(discard_active, training): _Program}; kept off the module (deepcopy / pickle safe)


class _Call:
    """Everything one forward call filled in (kept alive for the backward of the same call)."""
    __slots__ = ("prog", "c_prog", "c_bufs", "c_units", "c_tables", "c_keeps", "arena", "offsets", "keep_alive", "flat")


def _unit_structs(P: _Program):
    """vc_pass_unit array of the program's units for one call.  The array is kept on the program and an entry is rewritten only when one of
    its six tensors is another object or lives at another address than last time (flatten_parameters, .to(), load_state_dict(assign=True),
    `p.data = ...`) or a BatchNorm scalar changed: the tensors are read through the modules' own dicts, not through nn.Module.__getattr__
    -- 20-34 units x 11 attribute walks per forward were 0.1-0.2 ms of host time.  Every call gets its own copy (its backward reads it)."""
    n = len(P.units)
    cache = P.__dict__.get("_unit_cache")
    if cache is None:
        cache = P._unit_cache = ((_lib.PassUnit * n)(), [None] * n)
    arr, held = cache
    for i, (conv, bn) in enumerate(P.units):
        bp, bb, bd = bn._parameters, bn._buffers, bn.__dict__
        w, g, b = conv._parameters["weight"], bp["weight"], bp["bias"]
        rm, rv, nbt = bb["running_mean"], bb["running_var"], bb["num_batches_tracked"]
        ptrs = (w.data_ptr(), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr() if nbt is not None else 0,
                bd["momentum"], bd["eps"])
        h = held[i]
        if h is not None and h[0] is w and h[1] == ptrs:
            continue
        u = arr[i]
        u.weight, u.gamma, u.beta, u.running_mean, u.running_var = ptrs[:5]
        u.num_batches_tracked = ptrs[5] or None
        u.cin, u.cout, u.momentum, u.eps = w.shape[-1], w.shape[0], ptrs[6], ptrs[7]
        held[i] = (w, ptrs)
    return (_lib.PassUnit * n).from_buffer_copy(arr)


def _fill(P: _Program, feats, ctx, training: bool) -> Optional[_Call]:
    call = _Call()
    call.prog = P
    tables = [g(ctx) for g in P.table_getters]
    keeps = [g(ctx) for g in P.keep_getters]
    if any(rb.n_out <= 0 or rb.n_in <= 0 for rb in tables) or any(k is None or k.shape[0] == 0 for k in keeps):
        return None
    if training:
        # a duplicate-pixel table built under no_grad (a cached indice_dict entry, checkpoint-style recompute) has no group plan
        # yet; vc_pass_backward needs it: build it now and keep it on the rulebook
        be = ops.get_backend()
        for rb in tables:
            if ops.rulebook_ptr(rb, "rep") is not None and ops.rulebook_ptr(rb, "grp_plan") is None:
                rb.grp_plan = be.group_plan(rb.rep)
    c_tables = (_lib.PassTable * len(tables))()
    for i, rb in enumerate(tables):
        t = c_tables[i]
        subm = rb.kind == "subm"
        kv = rb.kv
        rp = ops.rulebook_ptr      # (a plan's rulebooks hand out addresses without creating tensor views)
        t.pair_fwd, t.pair_bwd, t.rep = rp(rb, "pair_fwd"), rp(rb, "pair_bwd"), rp(rb, "rep")
        t.order_fwd, t.order_bwd = rp(rb, "order_fwd"), rp(rb, "order_bwd")
        t.n_in, t.n_out, t.kv, t.subm = rb.n_in, rb.n_out, kv, 1 if subm else 0
        t.centre = kv // 2 if subm else -1     # odd kernel sizes: the centre tap is the middle offset
        t.sorted_rows = 1 if rb.sorted_rows else 0
        t.grp_plan = rp(rb, "grp_plan")
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
    c_units = _unit_structs(P)
    c_prog = _lib.PassProgram()
    c_prog.ops, c_prog.n_ops = P.c_ops, len(P.ops)
    c_prog.bufs, c_prog.n_bufs = c_bufs, len(P.cols)
    c_prog.units, c_prog.n_units = c_units, len(P.units)
    c_prog.tables, c_prog.n_tables = c_tables, len(tables)
    c_prog.keeps, c_prog.n_keeps = c_keeps, len(keeps)
    c_prog.training, c_prog.operand_type = 1 if training else 0, _lib.OPERAND_TYPES[ops.MFMA_OPERAND]
    # snapshot of the split-products switch: the forward and the backward arena layout of this call both follow it
    split = C.c_int64(0)
    be_ = ops.get_backend()
    c_prog.pack_all = 1 if (be_.lib.vc_debug_get(b"f32_split", C.byref(split)) == 0 and split.value != 0) else 0
    call.c_prog, call.c_bufs, call.c_units, call.c_tables, call.c_keeps = c_prog, c_bufs, c_units, c_tables, c_keeps
    call.keep_alive = (tables, keeps, feats)
    return call


class _NoGradCtx:
    """Stand-in for the autograd context when gradients are disabled (PassFunction.forward called directly)."""
    call = None
    flat_mode = False

    def save_for_backward(self, *tensors):
        pass

    def __setattr__(self, name, value):     # (the forward stores the call on its context: nothing to keep here)
        pass


_NO_CTX = _NoGradCtx()


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
        # the backward re-reads the input features and every parameter through the raw pointers bound in `call`: saving them makes
        # autograd's version counters catch an in-place edit between forward and backward
        ctx.flat_mode = call.flat is not None
        ctx.save_for_backward(feats, *params)
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
        if call is None:
            raise RuntimeError("virconv_amd feature pass: backward called twice on the same forward (its activation arena is "
                               "released after the first backward; retain_graph is not supported by the native pass -- set "
                               "VIRCONV_NATIVE_PASS=0 for the node-by-node path)")
        _ = ctx.saved_tensors   # raises if the input features or a parameter were modified in place since the forward
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
        need = ctx.needs_input_grad
        if ctx.flat_mode:
            # ONE parameter tensor (flatten_parameters): its gradient IS the flat buffer -- zeroed, because the alignment gaps between
            # the slots are part of the tensor the optimizer and the clip see
            flat = torch.zeros((P.grad_total,), dtype=torch.float32, device=dev)
            need = (need[0], None) + (need[2],) * (3 * len(P.units))
        else:
            flat = torch.empty((P.grad_total,), dtype=torch.float32, device=dev)
        base = flat.data_ptr()
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
        side = be._side_stream(dev) if be.pass_overlap_dw() else None
        _lib.check(lib.vc_pass_backward(prog, call.arena.data_ptr(), call.arena.numel(), ext, _ptr(gin), arena.data_ptr(),
                                        nbytes, side, be.stream()), "vc_pass_backward")
        # behind this point the stream carries no conv kernel of this pass any more (the weight-gradient stream is joined): the next
        # forward's pixel projection may start here instead of at the next forward's entry (backbone.note_pass_end, LOG.md A.17)
        from . import backbone
        backbone.note_pass_end(dev)
        if ctx.flat_mode:
            ctx.call = None
            return (gin, None, flat if need[2] else None)
        grads = []
        for i in range(len(P.grad_sizes)):
            if need[2 + i]:
                g = flat[P.grad_offsets[i]: P.grad_offsets[i] + P.grad_sizes[i]]
                grads.append(g.view(P.grad_shapes[i]) if P.grad_shapes[i] is not None else g)
            else:
                grads.append(None)
        ctx.call = None
        return (gin, None, *grads)


def _run_program(P: _Program, feats: torch.Tensor, ctx, training: bool):
    # the native calls bind raw pointers with dense row strides: a strided view (a column slice from a custom VFE, say) must be
    # packed first, and a non-contiguous parameter sends the call to the node-by-node path (which copies through _need())
    if not feats.is_contiguous():
        feats = feats.contiguous()
    params = []
    for conv, bn in P.units:     # (through the modules' own dicts: nn.Module.__getattr__ costs 1 us a piece, 60-100 of them per forward)
        bp = bn._parameters
        params += [conv._parameters["weight"], bp["weight"], bp["bias"]]
    if not all(p.is_contiguous() for p in params):
        return None
    call = _fill(P, feats, ctx, training)
    if call is None:
        return None
    call.flat = P.flat() if training else None
    if not torch.is_grad_enabled():
        # nothing to record: the body itself, without autograd.Function.apply walking 60 parameter arguments (inference: 0.05 ms per frame)
        return PassFunction.forward(_NO_CTX, feats, call)
    if call.flat is not None:
        return PassFunction.apply(feats, call, call.flat)   # ONE parameter input, ONE gradient (flatten_parameters)
    return PassFunction.apply(feats, call, *params)


def _program(model, key, build):
    cache = _PROGRAMS.setdefault(model, {})
    P = cache.get(key)
    if P is None:
        P = cache[key] = build()
    return P


def run(model, feats: torch.Tensor, plan):
    """VirConvL8x -> [x_conv1, x_conv2, x_conv3, x_conv4, out] feature matrices, or None when this call cannot take the native
    pass (an empty tensor somewhere)."""
    discard, training = model._discard_active(), bool(model.training)
    P = _program(model, ("L", discard, training), lambda: build_virconv_l_program(model, discard, training))
    return _run_program(P, feats, plan, training)


def run_8x_lidar(model, feats: torch.Tensor, rbs):
    """VirConv8x LiDAR stream -> [x_conv1..4, out] feature matrices or None."""
    return _run_program(_program(model, ("8x-lidar",), lambda: build_virconv8x_lidar_program(model)), feats, rbs, bool(model.training))


def run_8x_mm(model, feats: torch.Tensor, pm):
    """VirConv8x virtual-point stream -> [m1..m4] feature matrices (each after its layer discard) or None."""
    discard, training = model._discard_active(), bool(model.training)
    key = ("8x-mm", discard) if training else ("8x-mm", discard, False)
    P = _program(model, key, lambda: build_virconv8x_mm_program(model, discard, training))
    return _run_program(P, feats, pm, training)


# ------------------------------------------------------------------------------------------------ flat parameters (host time)
# A train step's host time is not its launches (all C-ABI calls together: 1.0 ms of 3.3) but Python / autograd around them: 60 parameter
# gradients handed back one by one (AccumulateGrad x 60), clip_grad_norm_ and the optimizer walking 60 tensors (0.5 ms; tools/hostsplit.py,
# profiles/r05_hostsplit.txt).  flatten_parameters re-homes the parameters of a backbone's native-pass units into ONE fp32 buffer per
# pass, laid out exactly like the pass's gradient buffer; the modules' own parameters become views of it (state_dict, checkpoint loading
# through copy_, .weight accesses are unchanged), the pass takes the flat tensor as its single parameter input and returns ONE gradient.
# The optimizer and the clip then run the stock torch functions on one or two tensors (what DistributedDataParallel's
# gradient_as_bucket_view / FSDP's flat parameters do for the same reason).  The norm of one flat tensor rounds differently from the norm
# of 60 per-tensor norms (last bit of the clip coefficient); everything else is element-wise and bit-identical (tests/test_flat_params_gpu.py).
def _training_programs(model):
    from .backbone import VirConv8x, VirConvL8x
    discard = model._discard_active() if hasattr(model, "_discard_active") else False
    if isinstance(model, VirConvL8x):
        return [_program(model, ("L", d, True), lambda d=d: build_virconv_l_program(model, d, True)) for d in (discard, not discard)]
    if isinstance(model, VirConv8x):
        progs = [_program(model, ("8x-lidar",), lambda: build_virconv8x_lidar_program(model))]
        if getattr(model, "mm", False):
            progs += [_program(model, ("8x-mm", d), lambda d=d: build_virconv8x_mm_program(model, d)) for d in (discard, not discard)]
        return progs
    return []


def flatten_parameters(model):
    """-> the list of tensors to hand to the optimizer / clip_grad_norm_ / the gradient all-reduce instead of model.parameters():
    one flat nn.Parameter per native pass (+ whatever parameter no pass covers).  Idempotent; the model must be on the GPU, in
    training mode, with every unit parameter trainable (else nothing is changed and list(model.parameters()) comes back)."""
    existing = getattr(model, "_vc_flat_params", None)
    if existing is not None:
        return list(existing)
    params = list(model.parameters())
    progs = _training_programs(model) if (params and params[0].is_cuda and model.training) else []
    if not progs or not all(p.requires_grad and p.dtype == torch.float32 for p in params):
        return params
    groups = {}                                       # unit set -> programs that share it (discard on / off: the same units in the same order)
    for P in progs:
        groups.setdefault(tuple(id(m) for u in P.units for m in u), []).append(P)
    covered, flats = set(), []
    for key, plist in groups.items():
        if any(k in covered for k in key):
            return params                             # two passes over one module: not a layout this scheme serves
        P = plist[0]
        assert all(Q.grad_offsets == P.grad_offsets for Q in plist)
        flat = torch.zeros((P.grad_total,), dtype=torch.float32, device=params[0].device)
        offs = P.grad_offsets
        with torch.no_grad():
            for i, (conv, bn) in enumerate(P.units):
                for t, o in ((conv.weight, offs[3 * i]), (bn.weight, offs[3 * i + 1]), (bn.bias, offs[3 * i + 2])):
                    view = flat[o: o + t.numel()].view(t.shape)
                    view.copy_(t)
                    t.data = view                     # the module's parameter now aliases the flat buffer
                    covered.add(id(t))
        fp = torch.nn.Parameter(flat)
        for Q in plist:
            Q._flat = fp
        flats.append(fp)
        covered.update(key)
    rest = [p for p in params if id(p) not in covered]
    out = flats + rest
    object.__setattr__(model, "_vc_flat_params", tuple(out))     # (not registered: state_dict and named_parameters are unchanged)
    return out


def trainable_parameters(model):
    """What a training loop should optimise: the flat parameters if flatten_parameters ran on this model, else model.parameters()."""
    fp = getattr(model, "_vc_flat_params", None)
    return list(fp) if fp is not None else list(model.parameters())
