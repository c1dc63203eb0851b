"""The geometry plan of an NRConvBlock chain as native calls (vc_plan_begin / vc_plan_wait / vc_plan_finish, csrc/plan.hip).

Reference path: the indice generation spconv runs conv by conv inside ``VirConvL8x.forward``
(pcdet/models/backbones_3d/spconv_backbone.py:609-699; NRConvBlock :150-229, layer_voxel_discard :134-147, conv_out :561-567),
one device-to-host sync per strided conv.  Round 3 built the same structures from Python (``backbone._plan_nrconv_chain``: ~90
ctypes calls, ~60 allocations and four pipelined count reads per train step -- 2.2 ms of host time per step and, through the
plan stream's kernel work, 0.84 ms of step time, tools/whatif.py).  Here: three C calls, two arenas, ONE count read, and index
kernels that use what the chain knows (bitmap-rank SubM rulebooks, dense pixel images, residue-class row orders, an LDS radix
sort for the group plans).  The result is the same ``plan`` dictionary of ``ops.Rulebook`` objects the Python path builds: the
feature pass and the node-by-node path consume it unchanged.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import _lib, ops

NATIVE_PLAN = os.environ.get("VIRCONV_NATIVE_PLAN", "1") != "0"
_DESCS = weakref.WeakKeyDictionary()      # model -> static part of its vc_plan_desc
_PINNED = {}                              # device index -> [pinned int32[RING, 64] for the counts, next slot, owners]
_RING = 64                                # plans between their begin and their finish, per device (= the library's event ring)
ARENA_ALLOC = None                        # developer hook (tools/det_check.py): callable(n int32 words) -> tensor, instead of torch.empty


_WARNED = []


def _warn_unfenced():
    if not _WARNED:
        _WARNED.append(1)
        import warnings
        warnings.warn("VIRCONV_PLAN_GUARD=0: the pixel projection of the geometry plan runs unfenced; beside the conv kernels of a feature pass "
                      "it computes wrong pixels in some waves (LOG.md A.17).  Diagnostics only.", RuntimeWarning, stacklevel=3)


def _arena(nbytes: int, dev) -> torch.Tensor:
    nwords = (nbytes + 3) >> 2
    if ARENA_ALLOC is not None:
        return ARENA_ALLOC(nwords)
    return torch.empty((nwords,), dtype=torch.int32, device=dev)


def _conv_geom(dst: _lib.PlanConv, conv) -> None:
    for a in range(3):
        dst.ksize[a], dst.stride[a] = int(conv.kernel_size[a]), int(conv.stride[a])
        dst.padding[a], dst.dilation[a] = int(conv.padding[a]), int(conv.dilation[a])


class ChainBlock:
    """One block of a chain: an optional strided conv (`down`), the 3-D SubM conv whose rulebook the block's SubM convs share
    (`subm`), and optionally the image-space branch (`conv2d` + the index2uv stride)."""
    __slots__ = ("down", "subm", "conv2d", "uv_stride")

    def __init__(self, down, subm, conv2d=None, uv_stride=1):
        self.down, self.subm, self.conv2d, self.uv_stride = down, subm, conv2d, int(uv_stride)


def nrconv_blocks(blocks):
    """[(NRConvBlock, uv stride)] -> [ChainBlock]"""
    return [ChainBlock(blk.down_layer[0] if blk.stride > 1 else None, blk.d3_conv1[0], blk.d2_conv1[0], uv) for blk, uv in blocks]


def _static_desc(chain, tail, sparse_shape, image_shape) -> _lib.PlanDesc:
    d = _lib.PlanDesc()
    d.n_blocks = len(chain)
    for a in range(3):
        d.spatial_shape[a] = int(sparse_shape[a])
    d.image_shape[0], d.image_shape[1] = int(image_shape[0]), int(image_shape[1])
    for b, cb in enumerate(chain):
        B = d.blocks[b]
        B.has_down = 1 if cb.down is not None else 0
        if cb.down is not None:
            _conv_geom(B.down, cb.down)
        for a in range(3):
            B.subm_ksize[a], B.subm_dilation[a] = int(cb.subm.kernel_size[a]), int(cb.subm.dilation[a])
        B.has_2d, B.uv_stride = (1 if cb.conv2d is not None else 0), cb.uv_stride
        if cb.conv2d is not None:
            for a in range(2):
                B.ksize2d[a], B.dilation2d[a] = int(cb.conv2d.kernel_size[a]), int(cb.conv2d.dilation[a])
    d.has_tail = 1 if tail is not None else 0
    if tail is not None:
        _conv_geom(d.tail, tail)
    return d


def usable(coords: torch.Tensor, blocks=None) -> bool:
    be = ops.get_backend()
    from .backbone import FAST_RANDOM_KEEP
    ok = bool(NATIVE_PLAN and getattr(be, "native_plan", False) and coords.is_cuda and coords.shape[0] > 0 and FAST_RANDOM_KEEP
              and ops.ROW_ORDER in ("bwd", "strided") and not ops.WINDOW_GATHER)
    if ok and blocks is not None:
        ok = len(blocks) <= _lib.PLAN_MAX_BLOCKS and all(
            not blk.conv_depth and blk._modules["d3_conv1"]._modules["0"].ndim == 3 and blk._modules["d2_conv1"]._modules["0"].ndim == 2
            for blk, _ in blocks)
    return ok


def _view(arenas, v: _lib.PlanView, external=None):
    if v.arena < 0:
        return external
    a = arenas[v.arena]
    return torch.as_strided(a, (v.rows, v.cols), (v.cols, 1), v.offset >> 2)


def _view1(arenas, v: _lib.PlanView):
    if v.arena < 0:
        return None
    return torch.as_strided(arenas[v.arena], (v.rows,), (1,), v.offset >> 2)


def _keep_view(arenas, v: _lib.PlanView):
    off = v.offset >> 2
    return arenas[v.arena][off: off + 2 * v.rows].view(torch.int64)


def _table(arenas, t: _lib.PlanTableOut, kind, in_shape, conv, in_idx) -> ops.Rulebook:
    """ops.Rulebook over views of the arenas; `in_idx`: the tensor holding the table's input coordinates."""
    geom = conv.__dict__.get("_vc_geom")       # the conv's static geometry as tuples of ints, made once per module (host time per table)
    if geom is None or geom[0] is not conv.kernel_size:
        nd = conv.ndim
        ks_ = tuple(int(k) for k in conv.kernel_size)
        geom = conv.__dict__["_vc_geom"] = (conv.kernel_size, nd, ks_, tuple(int(s) for s in conv.stride), tuple(int(p) for p in conv.padding),
                                            tuple(int(x) for x in conv.dilation), (1,) * nd, tuple(k // 2 for k in ks_))
    _, ndim, ks, g_stride, g_padding, g_dilation, g_one, g_half = geom
    out_shape = tuple(t.out_shape[:ndim])

    def lazy(v: _lib.PlanView, one_d=False):      # (arena, offset in words, rows, cols | 0): the view is made when somebody reads the field
        return None if v.arena < 0 else (arenas[v.arena], v.offset >> 2, v.rows, 0 if one_d else v.cols)

    views = {"pair_fwd": lazy(t.pair_fwd), "pair_bwd": lazy(t.pair_bwd), "rep": lazy(t.rep, True), "in_indices": in_idx,
             "out_indices": in_idx if kind == "subm" else lazy(t.out_indices), "order_fwd": lazy(t.order_fwd, True),
             "order_bwd": lazy(t.order_bwd, True), "grp_plan": lazy(t.grp_plan)}
    sparse = kind == "sparse"
    return ops.PlanRulebook(kind, t.n_in, t.n_out, tuple(in_shape), out_shape, ks, g_stride if sparse else g_one,
                            g_padding if sparse else g_half, g_dilation, views)


class ChainPlan:
    """The native plan of a chain of ChainBlocks (+ `tail`: the strided conv behind it) over the coordinates `idx` (N, 4) int32, in
    two steps.  The constructor enqueues vc_plan_begin on the current stream (coordinates, keeps and row counts of every level;
    no host synchronisation); `finish()` polls the counts (vc_plan_wait: the ONE host synchronisation), enqueues the tables and
    returns the result.  `build_chain` does both back to back; `backbone.VirConvL8x.plan_ahead_begin` enqueues the first half a
    training step early (the counts have long arrived when finish() asks for them).
    `discard_tags[b]`: the batch_dict tag of the layer discard after block b or None; `input_discard_tag`: discard of the chain's
    input (VirConv8x MM stream).  `kind`: cache key of the chain's static description on `model`.
    `deferred`: None -> finish() enqueues everything.  A list -> only what a FORWARD pass reads is enqueued; a closure that enqueues
    the rest (group plans, backward row orders: vc_plan_finish_backward) on the then-current stream is appended, and the caller
    runs it after the forward pass is on its stream (backbone.join_plan).
    `guard` (torch.cuda.Event | None): the stream waits for it in front of its first TABLE kernel (vc_plan_desc.tables_wait_event;
    backbone.PLAN_GUARD).  `defer_tables`: vc_plan_begin builds no table at all -- for callers that begin several plans, or one a
    step early, before they finish any (`finish(guard=...)` then names the event).  A chain with an image-space branch and no guard
    at all is refused unless `unfenced=True` (the caller's device is otherwise idle) or backbone.PLAN_GUARD == 0 (diagnostics)."""

    def __init__(self, model, kind: str, chain, tail, idx: torch.Tensor, batch_size: int, calib, trans_param, discard_tags, rate: float,
                 batch_dict, image_shape, input_discard_tag=None, deferred=None, guard=None, defer_tables=False, debug_buf=None,
                 unfenced=False, sparse_shape=None):
        self.unfenced = bool(unfenced)
        self.sparse_shape = [int(v) for v in (sparse_shape if sparse_shape is not None else model.sparse_shape)]
        be = ops.get_backend()
        lib = be.lib
        dev = idx.device
        cache = _DESCS.setdefault(model, {})
        if sparse_shape is not None:      # a chain over another grid than the model's (VirConv8x: the x-concatenated test-time tensor)
            kind = (kind, tuple(self.sparse_shape))
        if kind not in cache:
            cache[kind] = _static_desc(chain, tail, self.sparse_shape, image_shape)
        d = _lib.PlanDesc.from_buffer_copy(cache[kind])   # private copy: plans of one kind may be in flight together (plan-ahead, rids)
        hold = [idx, calib]
        d.indices, d.n, d.batch_size = idx.data_ptr(), idx.shape[0], int(batch_size)
        d.calib = calib.data_ptr() if calib is not None else None
        if trans_param is not None:
            trans_param = torch.as_tensor(trans_param, dtype=torch.float32, device=dev).reshape(batch_size, 3).contiguous()
            hold.append(trans_param)
            d.trans = trans_param.data_ptr()
        else:
            d.trans = None
        d.discard_rate = float(rate) if (any(t is not None for t in discard_tags) or input_discard_tag is not None) else 0.0
        d.need_grad = 1 if torch.is_grad_enabled() else 0
        d.row_order_fwd = 1 if ops.ROW_ORDER == "strided" else 0
        d.defer_early_tables = 1 if defer_tables else 0
        d.allow_unfenced_projection = 0
        d.tables_wait_event = None
        if guard is not None:
            hold.append(guard)
            d.tables_wait_event = guard.cuda_event
        d.debug_buf, d.debug_bytes = None, 0
        if debug_buf is not None:   # developer diagnostics (tools/det_check.py)
            hold.append(debug_buf)
            d.debug_buf, d.debug_bytes = debug_buf.data_ptr(), debug_buf.numel() * debug_buf.element_size()
        inj = batch_dict.get("layer_discard_keep")

        def keep_source(tag):
            """(seed, injected tensor | None): the injected permutation prefix, or a seed drawn from torch's CPU generator exactly as
            backbone.draw_random_keep does (same sequence of draws => same kept rows as the Python plan)."""
            if inj is not None:
                k = inj[tag].to(device=dev, dtype=torch.int64).contiguous()
                hold.append(k)
                return 0, k
            return int(torch.empty((), dtype=torch.int64).random_().item()) & 0xFFFFFFFFFFFFFFFF, None

        d.input_discard = 0
        if input_discard_tag is not None:
            seed, k = keep_source(input_discard_tag)
            d.input_discard, d.input_keep_seed = 1, seed
            d.input_keep, d.input_keep_rows = (k.data_ptr(), k.shape[0]) if k is not None else (None, 0)
        for b, tag in enumerate(discard_tags):
            B = d.blocks[b]
            B.discard = 1 if tag is not None else 0
            B.keep, B.keep_rows, B.keep_seed = None, 0, 0
            if tag is not None:
                seed, k = keep_source(tag)
                B.keep_seed = seed
                if k is not None:
                    B.keep, B.keep_rows = k.data_ptr(), k.shape[0]
        dref = C.byref(d)
        na = lib.vc_plan_begin_arena_bytes(dref)
        if na == 0:
            _lib.check(_lib.VC_EINVAL, "vc_plan_begin_arena_bytes")
        arena_a = _arena(na, dev)
        ring = _PINNED.get(dev.index)
        if ring is None:   # a ring of count buffers: up to _RING plans between their begin and their finish
            ring = _PINNED[dev.index] = [torch.empty((_RING, 64), dtype=torch.int32).pin_memory(), 0, [None] * _RING]
        slot = ring[1] % _RING
        owner = ring[2][slot]() if ring[2][slot] is not None else None
        if owner is not None and owner.result is None and not owner.counts_read:
            # a plan begun _RING plans ago and never finished (a plan begun ahead for a batch that never came, still referenced
            # somewhere): its count buffer is recycled now, so it can no longer be finished -- it says so if anyone tries
            owner.stale = True
        pinned = ring[0][slot]
        ring[1] += 1
        ring[2][slot] = weakref.ref(self)
        self.result, self.counts_read, self.stale = None, False, False
        state = _lib.PlanState()
        _lib.check(lib.vc_plan_begin(dref, arena_a.data_ptr(), arena_a.numel() * 4, pinned.data_ptr(), C.byref(state), be.stream()),
                   "vc_plan_begin")
        self.model, self.chain, self.tail, self.idx, self.discard_tags, self.input_discard_tag = model, chain, tail, idx, discard_tags, input_discard_tag
        self.image_shape, self.deferred = image_shape, deferred
        self.d, self.state, self.hold, self.arena_a, self.pinned = d, state, hold, arena_a, pinned

    def finish(self, guard=None):
        """-> (per block: {"down", "subm3d", "uv", "subm2d", "out_indices", "out_shape", "keep", "kept_indices"}, tail Rulebook | None,
        input keep | None, kept input indices | None, [arena_a, arena_b])"""
        if self.result is not None:
            return self.result
        if self.stale and not self.counts_read:
            raise RuntimeError(f"virconv_amd geometry plan: this plan was begun more than {_RING} plans ago and never finished; its pinned "
                               "count buffer has been recycled (begin it again)")
        be = ops.get_backend()
        lib = be.lib
        d, state, arena_a, hold = self.d, self.state, self.arena_a, self.hold
        dref, sref = C.byref(d), C.byref(state)
        st = be.stream()
        if guard is not None:
            hold.append(guard)
            d.tables_wait_event = guard.cuda_event
        if not d.tables_wait_event and any(cb.conv2d is not None for cb in self.chain):
            # LOG.md A.17: an image-space branch without the fence has to be asked for (vc_plan_finish refuses it otherwise)
            from . import backbone
            if backbone.PLAN_GUARD == 0 or self.unfenced:
                d.allow_unfenced_projection = 1
                if backbone.PLAN_GUARD == 0:
                    _warn_unfenced()
            else:
                raise RuntimeError("virconv_amd geometry plan: a plan with an image-space branch needs the event its pixel projection waits for "
                                   "(`guard`: recorded on the stream of the feature passes; backbone._PlanScope does it) -- or `unfenced=True` "
                                   "from a caller whose device is otherwise idle (LOG.md A.17)")
        _lib.check(lib.vc_plan_wait(dref, sref), "vc_plan_wait")          # the ONE host synchronisation of the plan
        self.counts_read = True
        nb = lib.vc_plan_finish_arena_bytes(dref, sref)
        if nb == 0:
            _lib.check(_lib.VC_EINVAL, "vc_plan_finish_arena_bytes")
        arena_b = _arena(nb, self.idx.device)
        out = _lib.PlanOut()
        _lib.check(lib.vc_plan_finish(dref, sref, arena_a.data_ptr(), arena_b.data_ptr(), arena_b.numel() * 4, C.byref(out), st),
                   "vc_plan_finish")
        if d.need_grad:
            if self.deferred is None:
                _lib.check(lib.vc_plan_finish_backward(dref, sref, arena_a.data_ptr(), arena_b.data_ptr(), arena_b.numel() * 4, st),
                           "vc_plan_finish_backward")
            else:
                def finish_backward(d=d, state=state, hold=hold, arena_a=arena_a, arena_b=arena_b):
                    _lib.check(lib.vc_plan_finish_backward(C.byref(d), C.byref(state), arena_a.data_ptr(), arena_b.data_ptr(),
                                                           arena_b.numel() * 4, be.stream()), "vc_plan_finish_backward")

                self.deferred.append(finish_backward)
        arenas = (arena_a, arena_b)
        res = []
        in_keep = in_kept = None
        cur_idx, shape = self.idx, list(self.sparse_shape)
        if self.input_discard_tag is not None:
            in_keep, in_kept = _keep_view(arenas, out.input_keep), _view(arenas, out.input_kept_indices)
            cur_idx = in_kept
        for b, (cb, tag) in enumerate(zip(self.chain, self.discard_tags)):
            O = out.blocks[b]
            r = {"down": None, "uv": None, "subm2d": None, "keep": None}
            if cb.down is not None:
                r["down"] = _table(arenas, O.down, "sparse", shape, cb.down, cur_idx)
                cur_idx, shape = r["down"].out_indices, list(r["down"].out_shape)
            r["subm3d"] = _table(arenas, O.subm3d, "subm", shape, cb.subm, cur_idx)
            if cb.conv2d is not None:
                r["uv"] = _view(arenas, O.uv)
                r["subm2d"] = _table(arenas, O.subm2d, "subm", self.image_shape, cb.conv2d, r["uv"])
            r["out_indices"], r["out_shape"] = cur_idx, shape
            if tag is not None:
                r["keep"] = _keep_view(arenas, O.keep)
                r["kept_indices"] = _view(arenas, O.kept_indices)
                cur_idx = r["kept_indices"]
            res.append(r)
        rb_tail = _table(arenas, out.tail, "sparse", shape, self.tail, cur_idx) if self.tail is not None else None
        self.hold = None   # the keeps / calibration were consumed by the launches enqueued above (stream order keeps them valid)
        self.result = (res, rb_tail, in_keep, in_kept, [arena_a, arena_b])
        return self.result


def build_chain(model, kind: str, chain, tail, idx: torch.Tensor, batch_size: int, calib, trans_param, discard_tags, rate: float,
                batch_dict, image_shape, input_discard_tag=None, deferred=None, guard=None, unfenced=False):
    """ChainPlan begun and finished back to back (see there)."""
    return ChainPlan(model, kind, chain, tail, idx, batch_size, calib, trans_param, discard_tags, rate, batch_dict, image_shape,
                     input_discard_tag, deferred, guard, unfenced=unfenced).finish()


def nrconv_stages(blocks, res):
    """ChainPlan result of a chain of NRConvBlocks -> the per-block dictionaries backbone._plan_nrconv_chain returns."""
    stages = []
    for (blk, _), r in zip(blocks, res):
        kd, k3, k2 = blk._keys()
        rbs3 = {}
        if r["down"] is not None:
            rbs3[kd] = r["down"]
        rbs3[k3] = r["subm3d"]
        st_ = {"rb3d": rbs3, "uv": r["uv"], "rb2d": {k2: r["subm2d"]}, "out_indices": r["out_indices"], "out_shape": r["out_shape"],
               "keep": r["keep"]}
        if r["keep"] is not None:
            st_["kept_indices"] = r["kept_indices"]
        stages.append(st_)
    return stages


def nrconv_kind(blocks, tail, input_discard_tag):
    return ("nrconv", len(blocks), tail is not None, input_discard_tag is not None)


def begin(model, blocks, tail, idx: torch.Tensor, batch_size: int, calib, trans_param, discard_tags, rate: float, batch_dict,
          image_shape, input_discard_tag=None, deferred=None) -> ChainPlan:
    """First half of `build` with every table left to the second (`finish_nrconv`): for callers with several plans per forward."""
    return ChainPlan(model, nrconv_kind(blocks, tail, input_discard_tag), nrconv_blocks(blocks), tail, idx, batch_size, calib, trans_param,
                     discard_tags, rate, batch_dict, image_shape, input_discard_tag, deferred, None, True)


def finish_nrconv(cp: ChainPlan, blocks, guard=None):
    res, rb_tail, in_keep, in_kept, arenas = cp.finish(guard)
    return nrconv_stages(blocks, res), rb_tail, in_keep, in_kept, arenas


def build(model, blocks, tail, idx: torch.Tensor, batch_size: int, calib, trans_param, discard_tags, rate: float, batch_dict,
          image_shape, input_discard_tag=None, deferred=None, guard=None, unfenced=False):
    """The plan of a chain of NRConvBlocks `blocks` = [(block, uv stride)] in the form backbone._plan_nrconv_chain returns:
    -> (stages, tail Rulebook | None, input keep | None, kept input indices | None, [arena_a, arena_b])."""
    res, rb_tail, in_keep, in_kept, arenas = build_chain(model, nrconv_kind(blocks, tail, input_discard_tag),
                                                         nrconv_blocks(blocks), tail, idx, batch_size, calib, trans_param,
                                                         discard_tags, rate, batch_dict, image_shape, input_discard_tag, deferred, guard,
                                                         unfenced)
    return nrconv_stages(blocks, res), rb_tail, in_keep, in_kept, arenas
