"""MI355X-native VirConv backbones: drop-in for ``pcdet.models.backbones_3d`` ``VirConvL8x`` / ``VirConv8x``.

Same constructor ``(model_cfg, input_channels, grid_size, **kwargs)``, same ``forward(batch_dict) -> batch_dict`` keys,
same submodule names (hence the same state_dict keys, so released VirConv-*.pth load unchanged) as
pcdet/models/backbones_3d/spconv_backbone.py:150-229 (NRConvBlock), :232-535 (VirConv8x), :538-699 (VirConvL8x).

What is different from running the reference file on the facade (which also works):
  * the voxel->pixel projection ``index2uv`` (python loop over the batch, ~15 small kernels + 3 host syncs per sample,
    spconv_backbone.py:54-83) is ONE HIP kernel over all rows (vc_project_uv), no host sync
  * the two 3-D SubM convs of a block share one rulebook (same coordinates; the reference gives them different
    indice_keys :186,199 and rebuilds), likewise the two 2-D convs
  * BatchNorm1d+ReLU run as a fused two-pass HIP op
  * layer discard (StVD): ``LAYER_DISCARD_MODE`` selects the spconv-2.x silent no-op (the default: what the reference does
    under its required spconv version) or the spconv-1.x in-place behaviour of the paper -- then a real device
    gather with injectable keep indices
"""
from __future__ import annotations

import os
from functools import partial
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import feature_pass, native_plan, ops
from . import spconv

# layer discard: draw the kept rows with vc_random_keep (point-wise pseudo-random permutation) instead of torch.randperm
FAST_RANDOM_KEEP = os.environ.get("VIRCONV_FAST_RANDOM_KEEP", "1") != "0"


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    if hasattr(cfg, "get"):
        try:
            return cfg.get(key, default)
        except TypeError:
            pass
    return getattr(cfg, key, default)


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type="subm",
                   norm_fn=None):
    """spconv_backbone.py:86-107."""
    if conv_type == "subm":
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
        relu = nn.ReLU()
    elif conv_type == "spconv":
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
        relu = nn.ReLU(inplace=True)
    elif conv_type == "inverseconv":
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
        relu = nn.ReLU()
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), relu)


def post_act_block2d(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type="subm",
                     norm_fn=None):
    """spconv_backbone.py:110-131."""
    if conv_type == "subm":
        conv = spconv.SubMConv2d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
        relu = nn.ReLU()
    elif conv_type == "spconv":
        conv = spconv.SparseConv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
        relu = nn.ReLU(inplace=True)
    elif conv_type == "inverseconv":
        conv = spconv.SparseInverseConv2d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
        relu = nn.ReLU()
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), relu)


def draw_random_keep(n: int, n_keep: int, device) -> torch.Tensor:
    """perm[:n_keep] of a random permutation of the n rows.  On the GPU the permutation is evaluated point-wise by
    vc_random_keep, seeded from torch's CPU generator (so torch.manual_seed reproduces it); torch.randperm otherwise."""
    be = ops.get_backend()
    if FAST_RANDOM_KEEP and torch.device(device).type == "cuda" and hasattr(be, "random_keep"):
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        return be.random_keep(n, n_keep, seed, device)
    return torch.randperm(n, device=device)[:n_keep]


def layer_voxel_discard(sp: spconv.SparseConvTensor, rate: float, keep: Optional[torch.Tensor] = None):
    """StVD layer discard (spconv_backbone.py:134-147), spconv-1.x semantics: keep rows ``perm[:int(N*(1-rate))]`` in
    permuted order.  ``keep`` injects the permutation prefix (tests / benchmarks); otherwise a device randperm is drawn
    (no host round trip -- the reference does np.random.permutation + H2D)."""
    if rate == 0:
        return sp
    n = sp.features.shape[0]
    n_keep = int(n * (1 - rate))
    if keep is None:
        keep = draw_random_keep(n, n_keep, sp.features.device)
    else:
        keep = keep.to(sp.features.device)
        assert keep.shape[0] == n_keep, f"injected keep has {keep.shape[0]} rows, expected int({n}*(1-{rate}))={n_keep}"
    f, idx = ops.discard_rows(sp.features, sp.indices, keep)
    return spconv.SparseConvTensor(f, idx, sp.spatial_shape, sp.batch_size)


class NRConvBlock(nn.Module):
    """Noise-resistant conv block (the paper's NRConv / "VirConvBlock"): 3-D convs + image-space 2-D convs, concatenated.
    spconv_backbone.py:150-229."""

    IMAGE_SHAPE = [1600, 600]  # spconv_backbone.py:220

    def __init__(self, input_c=16, output_c=16, stride=1, padding=1, indice_key="vir1", conv_depth=False):
        super().__init__()
        self.stride = stride
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.conv_depth = conv_depth
        if self.stride > 1:
            self.down_layer = post_act_block(input_c, output_c, 3, norm_fn=norm_fn, stride=stride, padding=padding,
                                             indice_key=("sp" + indice_key), conv_type="spconv")
        c1 = output_c if self.stride > 1 else input_c
        if self.conv_depth:
            c1 += 4
        c2 = output_c
        # the two 3-D (2-D) SubM convs see identical coordinates: one shared rulebook each
        self.d3_conv1 = post_act_block(c1, c2 // 2, 3, norm_fn=norm_fn, padding=1, indice_key=("subm3d" + indice_key))
        self.d2_conv1 = post_act_block2d(c2 // 2, c2 // 2, 3, norm_fn=norm_fn, padding=1, indice_key=("subm2d" + indice_key))
        self.d3_conv2 = post_act_block(c2 // 2, c2 // 2, 3, norm_fn=norm_fn, padding=1, indice_key=("subm3d" + indice_key))
        self.d2_conv2 = post_act_block2d(c2 // 2, c2 // 2, 3, norm_fn=norm_fn, padding=1, indice_key=("subm2d" + indice_key))

    # ---- geometry: everything below depends on coordinates only (never on features)
    def _keys(self):
        m = self._modules                      # (through the module dicts: called per block and forward, nn.Module.__getattr__ is 1 us a piece)
        k3 = m["d3_conv1"]._modules["0"].indice_key
        k2 = m["d2_conv1"]._modules["0"].indice_key
        kd = m["down_layer"]._modules["0"].indice_key if self.stride > 1 else None
        return kd, k3, k2

    def begin_down(self, indices, spatial_shape, batch_size):
        """Start the strided-conv rulebook of this block (None for a stride-1 block); `plan(..., pending=...)` finishes it."""
        if self.stride <= 1:
            return None
        conv = self.down_layer[0]
        return ops.begin_sparse_rulebook(indices, list(spatial_shape), batch_size, conv.kernel_size, conv.stride, conv.padding,
                                         conv.dilation)

    def plan(self, indices, spatial_shape, batch_size, calib, stride, trans_param, pending=None, after_down=None):
        """Build every index structure of the block from the input coordinates: the strided-conv rulebook (+ its output
        coordinates), the shared 3-D SubM rulebook, the pixel coordinates and the shared 2-D SubM rulebook.
        `pending`: the block's strided rulebook already begun (begin_down).  `after_down(indices, shape)`: called as soon as the
        block's output coordinates exist -- the chain uses it to begin the NEXT strided rulebook, whose count read then
        overlaps this block's SubM / projection / 2-D rulebook kernels."""
        kd, k3, k2 = self._keys()
        rbs3, shape = {}, list(spatial_shape)
        if self.stride > 1:
            rb = ops.finish_sparse_rulebook(pending if pending is not None else self.begin_down(indices, shape, batch_size))
            rbs3[kd] = rb
            indices, shape = rb.out_indices, list(rb.out_shape)
        extra = after_down(indices, shape) if after_down is not None else None
        rbs3[k3] = ops.build_subm_rulebook(indices, shape, self.d3_conv1[0].kernel_size, self.d3_conv1[0].dilation, False)
        if trans_param is not None:
            trans_param = torch.as_tensor(trans_param, dtype=torch.float32, device=indices.device).reshape(batch_size, 3)
        uv = ops.project_uv(indices, calib, trans_param, batch_size, stride)
        rb2 = ops.build_subm_rulebook(uv, self.IMAGE_SHAPE, self.d2_conv1[0].kernel_size, self.d2_conv1[0].dilation, True)
        return {"rb3d": rbs3, "uv": uv, "rb2d": {k2: rb2}, "out_indices": indices, "out_shape": shape, "after_down": extra}

    @staticmethod
    def _unit_is_plain(seq) -> bool:
        """conv (no bias) -> BatchNorm1d -> ReLU, the only shape post_act_block builds."""
        mods = list(seq._modules.values())
        return (len(mods) == 3 and getattr(seq, "fuse_bn_relu", False) and getattr(mods[0], "fusable_with_bn", False)
                and type(mods[1]) is nn.BatchNorm1d
                and type(mods[2]) is nn.ReLU and mods[1].affine and mods[1].track_running_stats
                and mods[1].momentum is not None and mods[1].num_features % 4 == 0)

    def _unit(self, seq, feats, rb):
        """One conv+BN+ReLU unit on a feature matrix with a ready rulebook -- what SparseSequential.forward does for this
        module triple, minus the per-layer SparseConvTensor bookkeeping (the plan path knows every index structure)."""
        conv, bn = seq[0], seq[1]
        if bn.training:
            return ops.conv_bn_relu(feats, conv.weight, rb, False, bn, True)
        out = ops.conv_bn_relu_eval(feats, conv.weight, rb, False, bn, True) if not torch.is_grad_enabled() else None
        if out is None:
            out = ops.bn_relu(ops.sparse_conv(feats, conv.weight, rb, False), bn, True)
        return out

    def _forward_planned(self, sp_tensor, batch_size, plan):
        kd, k3, k2 = self._keys()
        f = sp_tensor.features
        if self.stride > 1:
            f = self._unit(self.down_layer, f, plan["rb3d"][kd])
        rb3, rb2 = plan["rb3d"][k3], plan["rb2d"][k2]
        f3 = self._unit(self.d3_conv2, self._unit(self.d3_conv1, f, rb3), rb3)
        f2 = self._unit(self.d2_conv2, self._unit(self.d2_conv1, f3, rb2), rb2)
        indice_dict = dict(sp_tensor.indice_dict)
        indice_dict.update(plan["rb3d"])
        return spconv.SparseConvTensor(torch.cat([f3, f2], -1), plan["out_indices"], plan["out_shape"], batch_size,
                                       indice_dict=indice_dict)

    def forward(self, sp_tensor, batch_size, calib, stride, x_trans_train=None, trans_param=None, plan=None):
        if (plan is not None and sp_tensor.features.is_cuda and sp_tensor.features.shape[0] != 0
                and all(self._unit_is_plain(m) for m in ([self.down_layer] if self.stride > 1 else []) +
                        [self.d3_conv1, self.d3_conv2, self.d2_conv1, self.d2_conv2])):
            return self._forward_planned(sp_tensor, batch_size, plan)
        if plan is not None:
            sp_tensor.indice_dict.update(plan["rb3d"])
        if self.stride > 1:
            sp_tensor = self.down_layer(sp_tensor)
        d3_feat1 = self.d3_conv1(sp_tensor)
        d3_feat2 = self.d3_conv2(d3_feat1)

        if plan is not None:
            uv_coords, d2_dict = plan["uv"], dict(plan["rb2d"])
        else:
            if not torch.is_tensor(calib):
                calib = ops.calib_tensor(calib, d3_feat2.indices.device)
            if trans_param is not None:
                trans_param = torch.as_tensor(trans_param, dtype=torch.float32,
                                              device=d3_feat2.indices.device).reshape(batch_size, 3)
            uv_coords, d2_dict = ops.project_uv(d3_feat2.indices, calib, trans_param, batch_size, stride), None
        d2_sp_tensor1 = spconv.SparseConvTensor(d3_feat2.features, uv_coords, self.IMAGE_SHAPE, batch_size,
                                                indice_dict=d2_dict)
        d2_feat1 = self.d2_conv1(d2_sp_tensor1)
        d2_feat2 = self.d2_conv2(d2_feat1)
        return d3_feat2.replace_feature(torch.cat([d3_feat2.features, d2_feat2.features], -1))


def _record_stream(obj, stream):
    """Mark every tensor reachable from a plan as used on `stream` (it was allocated on the plan stream)."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, ops.Rulebook):
        for t in (obj.pair_fwd, obj.pair_bwd, obj.rep, obj.in_indices, obj.out_indices, obj.order_fwd, obj.order_bwd,
                  obj.grp_plan):
            _record_stream(t, stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


_PLAN_STREAMS = {}
# backward-only structures of a native plan are enqueued behind the event the forward pass waits for (VIRCONV_PLAN_DEFER_BACKWARD=0: A/B)
PLAN_DEFER_BACKWARD = os.environ.get("VIRCONV_PLAN_DEFER_BACKWARD", "1") != "0"


def _plan_stream(device):
    key = torch.device(device).index
    if key not in _PLAN_STREAMS:
        # high priority: the plan's short index kernels (and the host waiting on their counts) must not queue behind
        # the main stream's long conv kernels
        _PLAN_STREAMS[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("VIRCONV_PLAN_PRIORITY", "-1")))
    return _PLAN_STREAMS[key]


_FRAME_MARKS = {}


def _bound_run_ahead(device, main):
    """Flow control for forward-only loops.  The plan stream only waits for the inputs, so a host that is faster than the GPU can
    plan frame after frame while the main stream still owes the feature passes of earlier frames -- and the high-priority plan
    kernels of those future frames then starve the main stream: measured, the same inference loop runs at 1.35 ms/frame or at
    2.8-5.2 ms/frame depending on which side of that edge the host happens to sit.  Every forward marks the main stream on
    entry (= "everything up to the previous frame is enqueued before this point") and waits for the PREVIOUS forward's mark:
    at most two frames are ever in flight (plan of frame f over the feature pass of frame f - 1).  In training the mark is
    a step old and long complete.  -> this forward's mark."""
    key = torch.device(device).index
    prev = _FRAME_MARKS.get(key)
    mark = torch.cuda.Event()
    mark.record(main)
    _FRAME_MARKS[key] = mark
    if prev is not None:
        prev.synchronize()
    return mark


# LOG.md A.15 / A.17: the pixel projection of a plan (project_uv: the plan's only floating-point kernel) computes wrong pixels in lanes
# 48-63 of some waves when it runs on the plan stream beside the conv kernels of a feature pass -- forward or backward, training or
# pipelined inference; measured in round 5: 675 structures of 57 out of 64 steps of the round-4 training loop differ from a plan
# built on an idle GPU, 0 with the event below.  Memory, the parameter block, its flag word, the allocator, the third stream, packed
# fp32 instructions and the IEEE division sequence were all ruled out (tools/det_check.py); the exact-fp32-MFMA conv kernels never
# trigger it, the bf16-split ones do.  Whatever the silicon-level cause, the projection no longer shares the chip with them:
#   1 (default): the plan stream waits, in front of the image-space branch of vc_plan_finish (projection + pixel tables of all blocks:
#       three launches), for an event recorded on the main stream at the start of this forward -- behind everything the previous
#       step / frame enqueued.  The coordinate / count chain, the counts' trip to the host and every integer table (strided-conv pair
#       tables, 3-D SubM tables) still overlap the previous step.
#   2: the whole plan waits for that event (serial; A/B only).        0: no event (the round-4 loop; A/B and tools/det_check.py).
PLAN_GUARD = int(os.environ.get("VIRCONV_PLAN_GUARD", "1"))
NATIVE_CAT_PLAN = os.environ.get("VIRCONV_NATIVE_CAT_PLAN", "1") != "0"   # VirConv8x test-time path: chain plan over the x-concatenated tensor


def _guard_wanted(device) -> bool:
    return PLAN_GUARD != 0


# Round 6: WHAT the projection must be kept away from is known (tools/a17_lab.py, profiles/r06_a17_cu_mask.md): waves of the bf16-split
# gather-GEMM / weight-gradient kernels on the same compute unit.  Every other kernel of a train step -- BatchNorm, group sums, the loss,
# clip, AdamW, the exact-fp32-MFMA convs -- ran beside 800 projection launches without one wrong row, and the parameter-level soak
# (tests/test_soak_gpu.py) is green.  So the event need not be the START of the next forward: a native backward pass marks its own end
# (note_pass_end: behind its last conv kernel and the join of the weight-gradient stream) together with the library's count of conv
# launches; if no conv kernel has been enqueued since, the next plan waits for THAT event and its image-space branch runs under the
# clip / optimizer kernels instead of in a bubble in front of the forward pass (VIRCONV_PLAN_GUARD_EARLY=0: the round-5 placement).
PLAN_GUARD_EARLY = os.environ.get("VIRCONV_PLAN_GUARD_EARLY", "1") != "0"
_PASS_END = {}


def _conv_launch_seq() -> int:
    import ctypes
    v = ctypes.c_int64(0)
    be = ops.get_backend()
    if not hasattr(be, "lib") or be.lib.vc_debug_get(b"conv_launch_seq", ctypes.byref(v)) != 0:
        return -1
    return int(v.value)


def note_pass_end(device) -> None:
    """Called by the native feature pass at the end of its backward sweep (current stream = the caller's main stream)."""
    if not PLAN_GUARD_EARLY or PLAN_GUARD == 0:
        return
    ev = torch.cuda.Event()
    st = _current_stream()
    ev.record(st)
    _PASS_END[torch.device(device).index] = (ev, _conv_launch_seq(), st.cuda_stream)


def _early_guard(device, main):
    """The event of the last backward pass if nothing convolutional was enqueued behind it (and it sits on this stream), else None."""
    rec = _PASS_END.pop(torch.device(device).index, None)
    if rec is None:
        return None
    ev, seq, stream = rec
    if seq < 0 or seq != _conv_launch_seq() or stream != main.cuda_stream:
        return None
    return ev


def _current_stream():
    """torch.cuda.current_stream() of the current device without its device-index normalisation (8 of its 10 us; a forward asks 4-7 times)."""
    sid, didx, dtype = torch._C._cuda_getCurrentStream(torch._C._cuda_getDevice())
    return torch.cuda.Stream(stream_id=sid, device_index=didx, device_type=dtype)


class _PlanScope:
    """Run the geometry plan on the high-priority side stream (see VirConvL8x.build_plan) and hand the result to the main
    stream.  CPU tensors: a no-op scope.  `guard`: the event the plan's image-space branch waits for (see PLAN_GUARD), or None."""

    def __init__(self, ref_tensor, batch_dict, ahead=False):
        self.on_gpu = ref_tensor.is_cuda
        self.guard = None
        # closures that enqueue what only backward passes read (native_plan.build_chain): run AFTER the event the forward waits for
        self.deferred = [] if (ref_tensor.is_cuda and PLAN_DEFER_BACKWARD) else None
        if self.on_gpu:
            self.main = _current_stream()
            mark = None
            if not ahead:   # (a plan begun a step ahead is bounded by the training loop itself)
                mark = _bound_run_ahead(ref_tensor.device, self.main)
            self.side = _plan_stream(ref_tensor.device)
            if not ahead and _guard_wanted(ref_tensor.device):   # (a plan begun ahead names the event when it is finished)
                self.guard = (_early_guard(ref_tensor.device, self.main) if PLAN_GUARD_EARLY else None) or mark
            ready = batch_dict.get("inputs_ready_event")
            if ready is not None:
                self.side.wait_event(ready)  # inputs were produced before this event: no need to wait for the stream tail
            else:
                self.side.wait_stream(self.main)
            if self.guard is not None and PLAN_GUARD >= 2:
                self.side.wait_event(self.guard)
            self.ctx = torch.cuda.stream(self.side)
        else:
            import contextlib
            self.ctx = contextlib.nullcontext()

    def __enter__(self):
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)

    def guard_tables(self):
        """Python-composed plans (the fallback of native_plan; they interleave projections and tables): everything enqueued on the plan
        stream from here on waits for the guard."""
        if self.on_gpu and self.guard is not None:
            self.side.wait_event(self.guard)

    def publish(self, plan, defer_wait=False):
        """`defer_wait`: the main stream does NOT wait here; the event goes into plan["_fwd_ready"] and the caller waits for it where it
        first reads the plan."""
        if self.on_gpu:
            if self.deferred:
                # the forward pass waits for the tables IT reads.  The group-plan sorts and backward row orders are ENQUEUED by
                # join_plan -- after the forward pass is on the main stream: their ~50 launches cost the host 0.2 ms that used to
                # stand between the row counts and the first forward kernel -- and run on the plan stream underneath the forward
                fwd_ready = torch.cuda.Event()
                fwd_ready.record(self.side)
                if defer_wait:
                    plan["_fwd_ready"] = fwd_ready
                else:
                    self.main.wait_event(fwd_ready)
                plan["_deferred"] = (self.side, self.deferred)
            elif defer_wait:
                fwd_ready = torch.cuda.Event()
                fwd_ready.record(self.side)
                plan["_fwd_ready"] = fwd_ready
            else:
                self.main.wait_stream(self.side)
            # a native plan lives in two arenas (every structure is a view of one of them): marking those is marking everything
            _record_stream(plan["_arenas"] if (isinstance(plan, dict) and "_arenas" in plan) else plan, self.main)
        return plan


def join_plan(plan):
    """End of a forward pass: enqueue what only backward passes read (native_plan.build_chain's `deferred` closures) on the plan
    stream, and let everything behind this point on the current stream (the loss, the backward pass) wait for it."""
    d = plan.pop("_deferred", None) if isinstance(plan, dict) else None
    if d is not None:
        side, fns = d
        with torch.cuda.stream(side):
            for fn in fns:
                fn()
        _current_stream().wait_stream(side)


def _draw_keep(rate, n, batch_dict, tag, device):
    """Rows kept by layer_voxel_discard: the first floor(n * (1 - rate)) entries of a permutation (injected or drawn)."""
    n_keep = int(n * (1 - rate))
    inj = batch_dict.get("layer_discard_keep")
    if inj is not None:
        keep = inj[tag].to(device=device, dtype=torch.int64)
        assert keep.shape[0] == n_keep, f"injected keep has {keep.shape[0]} rows, expected {n_keep}"
        return keep
    return draw_random_keep(n, n_keep, device)


def _plan_nrconv_chain(blocks, in_idx, shape, batch_size, calib, trans_param, discard_tags, rate, batch_dict, tail=None):
    """Plans of consecutive NRConvBlocks (+ the layer discard after a block when its tag is not None).  `tail`: the strided
    conv that follows the chain (conv_out), or None; its begun rulebook is returned as the fourth value.
    Every strided rulebook needs ONE number on the host (its output row count).  The chain starts the next strided rulebook
    as soon as the current block's output coordinates (and its discard) exist and only then issues the current block's SubM /
    projection / 2-D rulebook kernels, so the count is on the host by the time it is needed."""
    stages = []
    # No layer discard anywhere (inference, or the reference's run-time behaviour under spconv 2.x): the active set only changes at
    # the strided convs, so ALL of their rulebooks are built up front with ONE host read (ops.build_sparse_rulebook_chain)
    # instead of one read per conv; the loop below then finds every strided rulebook ready.
    ready = None
    if all(t is None for t in discard_tags):
        convs = [blk.down_layer[0] for blk, _ in blocks if blk.stride > 1] + ([tail] if tail is not None else [])
        ready = ops.build_sparse_rulebook_chain(in_idx, shape, batch_size, convs) if convs else None
    if ready is not None:
        ready = list(ready)
        take_ready = lambda b: ready.pop(0) if (b is not None and b.stride > 1) else None   # noqa: E731
        pending = take_ready(blocks[0][0]) if blocks else None
    else:
        pending = blocks[0][0].begin_down(in_idx, shape, batch_size) if blocks else None
    for i, ((blk, stride), tag) in enumerate(zip(blocks, discard_tags)):
        nxt = blocks[i + 1][0] if i + 1 < len(blocks) else None

        def after_down(idx, shp, tag=tag, nxt=nxt, last=(i + 1 == len(blocks))):
            keep, kept = None, idx
            if tag is not None:
                keep = _draw_keep(rate, idx.shape[0], batch_dict, tag, idx.device)
                _, kept = ops.get_backend().gather_rows(None, idx, keep)
            if ready is not None:
                begun = take_ready(nxt) if nxt is not None else (ready.pop(0) if (last and tail is not None) else None)
            elif nxt is not None:
                begun = nxt.begin_down(kept, shp, batch_size)
            elif last and tail is not None:
                begun = ops.begin_sparse_rulebook(kept, shp, batch_size, tail.kernel_size, tail.stride, tail.padding, tail.dilation)
            else:
                begun = None
            return keep, kept, begun

        p = blk.plan(in_idx, shape, batch_size, calib, stride, trans_param, pending=pending, after_down=after_down)
        keep, kept, pending = p.pop("after_down")
        in_idx, shape = kept, p["out_shape"]
        p["keep"] = keep
        if keep is not None:
            p["kept_indices"] = kept
        stages.append(p)
    return stages, in_idx, shape, pending


def _run_nrconv_chain(blocks, stages, x, batch_size, calib, trans_param):
    outs = []
    for (blk, stride), p in zip(blocks, stages):
        x = blk(x, batch_size, calib, stride, None, trans_param, plan=p)
        if p["keep"] is not None:
            f = ops.GatherRowsFunction.apply(x.features, p["keep"])
            x = spconv.SparseConvTensor(f, p["kept_indices"], x.spatial_shape, batch_size)
        outs.append(x)
    return outs


class VirConvL8x(nn.Module):
    """VirConv-L backbone: one stream over fused LiDAR+virtual voxels (spconv_backbone.py:538-699)."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.return_num_features_as_dict = _cfg_get(model_cfg, "RETURN_NUM_FEATURES_AS_DICT", False)
        self.out_features = _cfg_get(model_cfg, "OUT_FEATURES", 64)
        self.layer_discard_rate = _cfg_get(model_cfg, "LAYER_DISCARD_RATE", 0.0)
        # default = what the reference does under its required spconv 2.x: layer_voxel_discard rebinds a local and returns
        # None, i.e. NO discard -- the released checkpoints were trained that way.  "spconv1_inplace" is the
        # opt-in paper / spconv-1.x behaviour (a real discard).
        self.layer_discard_mode = _cfg_get(model_cfg, "LAYER_DISCARD_MODE", "spconv2_noop")
        assert self.layer_discard_mode in ("spconv1_inplace", "spconv2_noop")
        self.plan_ahead = bool(_cfg_get(model_cfg, "PLAN_AHEAD", True))
        self._ahead = []        # plans begun for coming batches (plan_ahead_begin), oldest first; at most two
        self._fwd_count = 0     # forward passes so far: a plan begun during step t serves a LATER forward, never step t's own
        num_filters = _cfg_get(model_cfg, "NUM_FILTERS")
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]

        self.vir_conv1 = NRConvBlock(input_channels, num_filters[0], stride=1, indice_key="vir1")
        self.vir_conv2 = NRConvBlock(num_filters[0], num_filters[1], stride=2, indice_key="vir2")
        self.vir_conv3 = NRConvBlock(num_filters[1], num_filters[2], stride=2, indice_key="vir3")
        self.vir_conv4 = NRConvBlock(num_filters[2], num_filters[3], stride=2, padding=(0, 1, 1), indice_key="vir4")

        last_pad = _cfg_get(model_cfg, "last_pad", 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(num_filters[3], self.out_features, (3, 1, 1), stride=(2, 1, 1), padding=last_pad,
                                bias=False, indice_key="spconv_down2"),
            norm_fn(self.out_features),
            nn.ReLU(),
        )
        self.num_point_features = self.out_features
        if self.return_num_features_as_dict:
            self.num_point_features = {"x_conv1": num_filters[0], "x_conv2": num_filters[1], "x_conv3": num_filters[2],
                                       "x_conv4": num_filters[3]}

    def _discard_active(self):
        return self.training and self.layer_discard_mode != "spconv2_noop" and self.layer_discard_rate != 0

    def _discard(self, sp, batch_dict, tag):
        if not self._discard_active():
            return sp
        keep = None
        inj = batch_dict.get("layer_discard_keep")
        if inj is not None:
            keep = inj[tag]
        return layer_voxel_discard(sp, self.layer_discard_rate, keep)

    def build_plan(self, coords, batch_size, calib, trans_param, batch_dict, rid=""):
        """GEOMETRY PLAN.  All rulebooks, pixel coordinates, strided-conv output sets and discard permutations of the
        whole backbone depend on coordinates only, so they are built first -- on a side stream, where the four
        data-dependent size reads (one per strided conv) only wait for a few short index kernels instead of draining
        the feature work queued on the main stream.  The feature pass that follows is free of host syncs."""
        blocks = [(self.vir_conv1, 1), (self.vir_conv2, 2), (self.vir_conv3, 4), (self.vir_conv4, 8)]
        active = self._discard_active()
        tags = [f"x_conv{bi + 1}{rid}" if (bi < 3 and active) else None for bi in range(4)]
        with _PlanScope(coords, batch_dict) as scope:
            idx = coords.int()
            co = self.conv_out[0]
            if native_plan.usable(idx, blocks):
                # the whole plan as three native calls around ONE count read (csrc/plan.hip)
                stages, rb_out, _, _, arenas = native_plan.build(self, blocks, co, idx, batch_size, calib, trans_param, tags,
                                                                 self.layer_discard_rate, batch_dict, NRConvBlock.IMAGE_SHAPE,
                                                                 deferred=scope.deferred, guard=scope.guard)
                plan = {"in_indices": idx, "stages": stages, "conv_out": {co.indice_key: rb_out}, "_arenas": arenas + [idx]}
            else:
                scope.guard_tables()
                stages, in_idx, shape, begun = _plan_nrconv_chain(blocks, idx, list(self.sparse_shape), batch_size, calib,
                                                                  trans_param, tags, self.layer_discard_rate, batch_dict, tail=co)
                rb_out = ops.finish_sparse_rulebook(begun)
                plan = {"in_indices": idx, "stages": stages, "conv_out": {co.indice_key: rb_out}}
        return scope.publish(plan)

    # ---- plan-ahead: the first half of the geometry plan of the NEXT batch under the current step's forward.  The plan depends on the
    # batch's coordinates (and calibration / augmentation parameters / discard seeds) only, not on the weights, so a training loop that
    # holds batch t + 1 while it runs step t -- any prefetching loader does -- can enqueue its coordinate / keep / row-count chain
    # (vc_plan_begin: no host synchronisation) a step early:
    #     model.plan_ahead_begin(next_batch)      # before the forward of step t
    # forward(next_batch) then finds the counts on the host without waiting, builds the tables and runs (matched by the identity and
    # version of its voxel_coords tensor; a mismatch of any kind just discards the early half and plans in place).  Measured neutral on
    # MI355X (the plan costs kernel work, not latency: LOG.md A.10).  The TABLES are deliberately not built under the previous step's
    # backward: with that overlap a few waves of the pixel projection came out wrong (LOG.md A.15, cause not found).
    def _rot_inputs(self, batch_dict):
        rot_num = batch_dict["transform_param"].shape[1] if "transform_param" in batch_dict else 1
        for i in range(rot_num):
            rid = "" if i == 0 else str(i)
            trans_param = batch_dict["aug_param"] if "aug_param" in batch_dict else None
            if "transform_param" in batch_dict:
                trans_param = batch_dict["transform_param"][:, i, :]
            yield rid, trans_param

    def _ahead_mode(self):
        return (self.training, torch.is_grad_enabled(), self._discard_active(), ops.ROW_ORDER)

    def plan_ahead_begin(self, batch_dict) -> bool:
        coords0 = batch_dict["voxel_coords"]
        if not (self.plan_ahead and coords0.is_cuda):
            return False
        calib = batch_dict["calib"]
        if not torch.is_tensor(calib):
            calib = ops.calib_tensor(calib, coords0.device)
        batch_size = batch_dict["batch_size"]
        blocks = [(self.vir_conv1, 1), (self.vir_conv2, 2), (self.vir_conv3, 4), (self.vir_conv4, 8)]
        co = self.conv_out[0]
        active = self._discard_active()
        entries = {}
        scope = _PlanScope(coords0, batch_dict, ahead=True)
        with scope:
            for rid, trans_param in self._rot_inputs(batch_dict):
                coords = batch_dict["voxel_coords" + rid]
                idx = coords.int()
                if not native_plan.usable(idx, blocks):
                    return False
                tags = [f"x_conv{bi + 1}{rid}" if (bi < 3 and active) else None for bi in range(4)]
                cp = native_plan.ChainPlan(self, native_plan.nrconv_kind(blocks, co, None), native_plan.nrconv_blocks(blocks), co, idx,
                                           batch_size, calib, trans_param, tags, self.layer_discard_rate, batch_dict,
                                           NRConvBlock.IMAGE_SHAPE, None, scope.deferred, None, True)
                entries[rid] = (coords, coords._version, idx, cp)
        self._ahead.append({"entries": entries, "scope": scope, "mode": self._ahead_mode(), "blocks": blocks, "fwd_ready": None,
                            "born": self._fwd_count})
        del self._ahead[:-2]      # the current step's and the next one's; anything older was never consumed
        return True

    @staticmethod
    def _finish_ahead(a) -> None:
        """Second half of a plan begun ahead: EVERY table, at the start of the step that uses it, behind the guard event recorded now on
        the main stream (i.e. behind the previous step's backward pass; PLAN_GUARD)."""
        if a["fwd_ready"] is None:
            guard = None
            if _guard_wanted(a["scope"].side.device):
                guard = torch.cuda.Event()
                guard.record(_current_stream())
            with torch.cuda.stream(a["scope"].side):
                for _, _, _, cp in a["entries"].values():
                    cp.finish(guard)
                a["fwd_ready"] = torch.cuda.Event()
                a["fwd_ready"].record()

    def _take_ahead(self, coords, rid):
        mode = self._ahead_mode()
        for a in self._ahead:
            e = a["entries"].get(rid)
            if (e is not None and e[0] is coords and e[1] == coords._version and a["mode"] == mode and a["born"] < self._fwd_count
                    and not any(x[3].stale for x in a["entries"].values())):
                break
        else:
            return None            # no early plan for this batch (or not in this mode): the caller plans in place
        if rid == "":
            _bound_run_ahead(coords.device, _current_stream())   # the flow control every forward has (see _PlanScope)
        self._finish_ahead(a)
        _, _, idx, cp = a["entries"].pop(rid)
        res, rb_out, _, _, arenas = cp.finish()
        plan = {"in_indices": idx, "stages": native_plan.nrconv_stages(a["blocks"], res),
                "conv_out": {self.conv_out[0].indice_key: rb_out}, "_arenas": arenas + [idx]}
        scope = a["scope"]
        main = _current_stream()
        main.wait_event(a["fwd_ready"])
        if scope.deferred:
            plan["_deferred"] = (scope.side, scope.deferred)
            scope.deferred = []
        _record_stream(plan["_arenas"], main)
        if not a["entries"]:
            self._ahead.remove(a)
        return plan

    def forward(self, batch_dict):
        if "transform_param" in batch_dict:
            rot_num = batch_dict["transform_param"].shape[1]
        else:
            rot_num = 1
        batch_size = batch_dict["batch_size"]
        calib = batch_dict["calib"]
        if not torch.is_tensor(calib):
            calib = ops.calib_tensor(calib, batch_dict["voxel_features"].device)

        for i in range(rot_num):
            rid = "" if i == 0 else str(i)
            feats, coords = batch_dict["voxel_features" + rid], batch_dict["voxel_coords" + rid]
            feats[:, 4:7] = 0  # remove the RGB features, in place on the batch tensor (spconv_backbone.py:636)

            if "aug_param" in batch_dict:
                trans_param = batch_dict["aug_param"]
            else:
                trans_param = None
            if "transform_param" in batch_dict:
                trans_param = batch_dict["transform_param"][:, i, :]

            if self.plan_ahead:
                plan = self._take_ahead(coords, rid)
                if plan is None:
                    plan = self.build_plan(coords, batch_size, calib, trans_param, batch_dict, rid)
                if "plan_observer" in batch_dict:      # tests: hand the step's geometry plan out (tests/test_plan_stress_gpu.py)
                    batch_dict["plan_observer"](rid, plan)
                native = feature_pass.run(self, feats, plan) if feature_pass.usable(self, feats, plan) else None
                if native is not None:
                    # the whole chain below as ONE native call per direction (virconv_amd/feature_pass.py): same kernels, same order
                    xs, idict = [], {}
                    for st, f in zip(plan["stages"], native[:4]):
                        idict = dict(idict)
                        idict.update(st["rb3d"])
                        kept = st["keep"] is not None
                        xs.append(spconv.SparseConvTensor(f, st["kept_indices"] if kept else st["out_indices"], st["out_shape"],
                                                          batch_size, indice_dict={} if kept else idict))
                        if kept:
                            idict = {}
                    x1, x2, x3, x4 = xs
                    x4.indice_dict.update(plan["conv_out"])
                    rb_out = plan["conv_out"][self.conv_out[0].indice_key]
                    out = spconv.SparseConvTensor(native[4], rb_out.out_indices, list(rb_out.out_shape), batch_size,
                                                  indice_dict=x4.indice_dict)
                else:
                    x = spconv.SparseConvTensor(feats, plan["in_indices"], self.sparse_shape, batch_size)
                    blocks = [(self.vir_conv1, 1), (self.vir_conv2, 2), (self.vir_conv3, 4), (self.vir_conv4, 8)]
                    outs = _run_nrconv_chain(blocks, plan["stages"], x, batch_size, calib, trans_param)
                    x1, x2, x3, x4 = outs
                    x4.indice_dict.update(plan["conv_out"])
                    out = self.conv_out(x4)
            else:
                x0 = spconv.SparseConvTensor(feats, coords.int(), self.sparse_shape, batch_size)
                x1 = self.vir_conv1(x0, batch_size, calib, 1, None, trans_param)
                x1 = self._discard(x1, batch_dict, f"x_conv1{rid}")
                x2 = self.vir_conv2(x1, batch_size, calib, 2, None, trans_param)
                x2 = self._discard(x2, batch_dict, f"x_conv2{rid}")
                x3 = self.vir_conv3(x2, batch_size, calib, 4, None, trans_param)
                x3 = self._discard(x3, batch_dict, f"x_conv3{rid}")
                x4 = self.vir_conv4(x3, batch_size, calib, 8, None, trans_param)
                out = self.conv_out(x4)

            if self.plan_ahead:
                join_plan(plan)
            batch_dict.update({
                "encoded_spconv_tensor" + rid: out,
                "encoded_spconv_tensor_stride" + rid: 8,
                "multi_scale_3d_features" + rid: {"x_conv1": x1, "x_conv2": x2, "x_conv3": x3, "x_conv4": x4},
                "multi_scale_3d_strides" + rid: {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8},
            })
        self._fwd_count += 1
        return batch_dict


class VirConv8x(nn.Module):
    """VirConv-T / VirConv-S backbone (spconv_backbone.py:232-535): a LiDAR stream (conv_input, conv1..4, conv_out with
    shared rulebooks per stage) and, with ``MM: True``, the virtual-point stream of four NRConvBlocks.

    Training: the LiDAR stream runs once per transformed frame (:362-407).  Eval: the ``rot_num`` frames are
    concatenated along x into ONE sparse tensor of width 4*W (:409-442) so a single set of launches serves all frames,
    then split back with ``decompose_tensor`` (:314-337, strict ``begin < x < end`` -- the x == begin voxels of each slab
    are dropped, reference quirk kept; 64-bit coordinate keys make the 7.3e8-cell tensor safe at any batch size).
    """

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.return_num_features_as_dict = _cfg_get(model_cfg, "RETURN_NUM_FEATURES_AS_DICT", False)
        self.out_features = _cfg_get(model_cfg, "OUT_FEATURES", 64)
        self.layer_discard_rate = _cfg_get(model_cfg, "LAYER_DISCARD_RATE", 0.0)
        # default = what the reference does under its required spconv 2.x: layer_voxel_discard rebinds a local and returns
        # None, i.e. NO discard -- the released checkpoints were trained that way.  "spconv1_inplace" is the
        # opt-in paper / spconv-1.x behaviour (a real discard).
        self.layer_discard_mode = _cfg_get(model_cfg, "LAYER_DISCARD_MODE", "spconv2_noop")
        assert self.layer_discard_mode in ("spconv1_inplace", "spconv2_noop")
        self.mm = bool(_cfg_get(model_cfg, "MM", False))
        self.plan_ahead = bool(_cfg_get(model_cfg, "PLAN_AHEAD", True))
        nf = _cfg_get(model_cfg, "NUM_FILTERS")
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]
        block = post_act_block

        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, nf[0], 3, padding=1, bias=False, indice_key="subm1"),
            norm_fn(nf[0]), nn.ReLU())
        self.conv1 = spconv.SparseSequential(block(nf[0], nf[0], 3, norm_fn=norm_fn, padding=1, indice_key="subm1"))
        self.conv2 = spconv.SparseSequential(
            block(nf[0], nf[1], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
            block(nf[1], nf[1], 3, norm_fn=norm_fn, padding=1, indice_key="subm2"),
            block(nf[1], nf[1], 3, norm_fn=norm_fn, padding=1, indice_key="subm2"))
        self.conv3 = spconv.SparseSequential(
            block(nf[1], nf[2], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3", conv_type="spconv"),
            block(nf[2], nf[2], 3, norm_fn=norm_fn, padding=1, indice_key="subm3"),
            block(nf[2], nf[2], 3, norm_fn=norm_fn, padding=1, indice_key="subm3"))
        self.conv4 = spconv.SparseSequential(
            block(nf[2], nf[3], 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4", conv_type="spconv"),
            block(nf[3], nf[3], 3, norm_fn=norm_fn, padding=1, indice_key="subm4"),
            block(nf[3], nf[3], 3, norm_fn=norm_fn, padding=1, indice_key="subm4"))
        last_pad = _cfg_get(model_cfg, "last_pad", 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(nf[3], self.out_features, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key="spconv_down2"),
            norm_fn(self.out_features), nn.ReLU())
        if self.mm:
            self.vir_conv1 = NRConvBlock(input_channels, nf[0], stride=1, indice_key="vir1")
            self.vir_conv2 = NRConvBlock(nf[0], nf[1], stride=2, indice_key="vir2")
            self.vir_conv3 = NRConvBlock(nf[1], nf[2], stride=2, indice_key="vir3")
            self.vir_conv4 = NRConvBlock(nf[2], nf[3], stride=2, padding=(0, 1, 1), indice_key="vir4")
        self.num_point_features = self.out_features
        if self.return_num_features_as_dict:
            self.num_point_features = {"x_conv1": nf[0], "x_conv2": nf[1], "x_conv3": nf[2], "x_conv4": nf[3]}

    @staticmethod
    def decompose_tensor(tensor, i, batch_size):
        """Slab i of an x-concatenated tensor (spconv_backbone.py:314-337)."""
        w = tensor.spatial_shape[2]
        begin, end = i * (w // 4), (i + 1) * (w // 4)
        x = tensor.indices[:, 3]
        keep = torch.nonzero((begin < x) & (x < end)).squeeze(1)
        if tensor.features.is_cuda:
            feats, idx = ops.discard_rows(tensor.features, tensor.indices, keep)
        else:
            feats, idx = tensor.features[keep], tensor.indices[keep]
        idx = idx.clone()
        idx[:, 3] -= begin
        shape = [tensor.spatial_shape[0], tensor.spatial_shape[1], tensor.spatial_shape[2] // 4]
        return spconv.SparseConvTensor(feats, idx.int(), shape, batch_size)

    def _lidar_stream(self, sp):
        x = self.conv_input(sp)
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        return x1, x2, x3, x4, self.conv_out(x4)

    def _discard_active(self):
        return self.training and self.layer_discard_mode != "spconv2_noop" and self.layer_discard_rate != 0

    def _discard(self, sp, batch_dict, tag):
        if not self._discard_active():
            return sp
        inj = batch_dict.get("layer_discard_keep")
        return layer_voxel_discard(sp, self.layer_discard_rate, None if inj is None else inj[tag])

    # ---- geometry plan (same idea as VirConvL8x.build_plan): every rulebook of both streams, the eval-time slab splits and
    # the discard permutations are functions of the coordinates only and are built first, on the plan stream
    def _lidar_chain(self):
        first, co = self.conv_input[0], self.conv_out[0]
        stages = (self.conv2, self.conv3, self.conv4)
        chain = [native_plan.ChainBlock(None, first)] + [native_plan.ChainBlock(seq[0][0], seq[1][0]) for seq in stages]
        return first, co, stages, chain

    def _begin_lidar(self, idx, batch_size, batch_dict, deferred, sparse_shape=None):
        """First half of the LiDAR stream's native chain plan (native_plan.ChainPlan): coordinates, row counts, the first 3-D table.
        `sparse_shape`: the grid when it is not the model's own (the x-concatenated test-time tensor, spconv_backbone.py:414-432)."""
        _, co, _, chain = self._lidar_chain()
        return native_plan.ChainPlan(self, "8x-lidar", chain, co, idx, batch_size, None, None, [None] * 4, 0.0, batch_dict,
                                     NRConvBlock.IMAGE_SHAPE, None, deferred, None, False, sparse_shape=sparse_shape)

    def _finish_lidar(self, cp, guard, arenas):
        first, co, stages, _ = self._lidar_chain()
        res, rb_tail, _, _, ar = cp.finish(guard)
        arenas.extend(ar)
        rbs, coords = {first.indice_key: res[0]["subm3d"]}, {}
        for name, seq, r in zip(("x2", "x3", "x4"), stages, res[1:]):
            rbs[seq[0][0].indice_key], rbs[seq[1][0].indice_key] = r["down"], r["subm3d"]
            coords[name] = (r["out_indices"], list(r["out_shape"]))
        rbs[co.indice_key] = rb_tail
        coords["out"] = (rb_tail.out_indices, list(rb_tail.out_shape))
        return rbs, coords

    def _plan_lidar(self, idx, shape, batch_size):
        """{indice_key: Rulebook} of the LiDAR stream + the coordinates of x2, x3, x4 and the output, operator by operator (the eval
        path over the x-concatenated tensor, and the fallback of the native chain plan)."""
        first, co = self.conv_input[0], self.conv_out[0]
        rbs = {}
        rbs[first.indice_key] = ops.build_subm_rulebook(idx, shape, first.kernel_size, first.dilation, False)
        cur, cur_shape, coords = idx, list(shape), {}
        for name, seq in (("x2", self.conv2), ("x3", self.conv3), ("x4", self.conv4)):
            down, subm = seq[0][0], seq[1][0]
            rb = ops.build_sparse_rulebook(cur, cur_shape, batch_size, down.kernel_size, down.stride, down.padding, down.dilation)
            rbs[down.indice_key] = rb
            cur, cur_shape = rb.out_indices, list(rb.out_shape)
            rbs[subm.indice_key] = ops.build_subm_rulebook(cur, cur_shape, subm.kernel_size, subm.dilation, False)
            coords[name] = (cur, cur_shape)
        rb = ops.build_sparse_rulebook(cur, cur_shape, batch_size, co.kernel_size, co.stride, co.padding, co.dilation)
        rbs[co.indice_key] = rb
        coords["out"] = (rb.out_indices, list(rb.out_shape))
        return rbs, coords

    @staticmethod
    def _plan_split(indices, shape, i):
        """Geometry half of decompose_tensor: kept rows and shifted coordinates of slab i."""
        w = shape[2]
        begin, end = i * (w // 4), (i + 1) * (w // 4)
        x = indices[:, 3]
        keep = torch.nonzero((begin < x) & (x < end)).squeeze(1)
        idx = indices[keep].clone()
        idx[:, 3] -= begin
        return keep, idx.int(), [shape[0], shape[1], shape[2] // 4]

    @staticmethod
    def _plan_splits(co, rids):
        """_plan_split for every slab and every tensor with ONE host read: rows sorted stably by slab (dropped rows last), the slab
        sizes of the three tensors fetched together.  Same kept rows in the same order as the `nonzero` of _plan_split (the rows of a
        strided conv's output are in ascending coordinate order and the sort is stable).  -> {rid: {"x3" | "x4" | "out": split}}"""
        n_rot = len(rids)
        metas = []
        for k in ("x3", "x4", "out"):
            indices, shape = co[k]
            qw = shape[2] // 4
            x = indices[:, 3].long()
            slab = torch.div(x, qw, rounding_mode="floor")
            key = torch.where((x % qw != 0) & (slab < n_rot), slab, torch.full_like(slab, n_rot))   # strict begin < x < end
            metas.append((k, indices, shape, qw, torch.argsort(key, stable=True), torch.bincount(key, minlength=n_rot + 1)[:n_rot]))
        counts = torch.stack([m[5] for m in metas]).cpu().tolist()                                   # the one host read
        out = {rid: {} for rid in rids}
        for (k, indices, shape, qw, order, _), cnt in zip(metas, counts):
            off = 0
            for i, rid in enumerate(rids):
                keep = order[off: off + cnt[i]]
                off += cnt[i]
                idx = indices[keep].clone()
                idx[:, 3] -= i * qw
                out[rid][k] = (keep, idx.int(), [shape[0], shape[1], shape[2] // 4])
        return out

    def build_plan(self, batch_dict, rids, batch_size, calib):
        ref = batch_dict["voxel_coords"]
        plan = {"lidar": {}, "split": {}, "mm": {}}
        arenas = []     # every structure of a native chain plan is a view of one of its two arenas
        blocks = [(self.vir_conv1, 1), (self.vir_conv2, 2), (self.vir_conv3, 4), (self.vir_conv4, 8)]
        active = self._discard_active()

        def mm_inputs(i, rid):
            trans_param = batch_dict.get("aug_param")
            if "transform_param" in batch_dict:
                trans_param = batch_dict["transform_param"][:, i, :]
            tags = [f"mm_x_conv{bi + 1}{rid}" if (bi < 3 and active) else None for bi in range(4)]
            return trans_param, tags

        with _PlanScope(ref, batch_dict) as scope:
            idx_l = {rid: batch_dict["voxel_coords" + rid].int() for rid in rids} if self.training else {}
            idx_m = {rid: batch_dict["voxel_coords_mm" + rid].int() for rid in rids} if self.mm else {}
            if (self.training and all(native_plan.usable(v) for v in idx_l.values())
                    and all(native_plan.usable(v, blocks) for v in idx_m.values())):
                # every chain natively, one after the other (begin -> one count read -> tables).  The LiDAR stream's tables are integer
                # work and overlap the previous step freely; only the image-space branch of the virtual-point stream waits for the guard
                # event (PLAN_GUARD), inside its own vc_plan_finish -- behind its coordinate chain and count read, so no host read ever
                # waits behind the guard
                # every chain is BEGUN (coordinates, keeps, row counts: no host synchronisation) before any is finished: the count
                # reads of the later chains are on the host by the time the first one's has been waited for (one spin, not one per chain)
                begun_l = {rid: self._begin_lidar(idx_l[rid], batch_size, batch_dict, scope.deferred) for rid in rids}
                begun_m = list(rids) if self.mm else []
                cps_m = {}
                for i, rid in enumerate(begun_m):
                    trans_param, tags = mm_inputs(i, rid)
                    cps_m[rid] = native_plan.begin(self, blocks, None, idx_m[rid], batch_size, calib, trans_param, tags,
                                                   self.layer_discard_rate, batch_dict, NRConvBlock.IMAGE_SHAPE,
                                                   input_discard_tag=(f"mm_input{rid}" if active else None), deferred=scope.deferred)
                for rid in rids:
                    rbs, _ = self._finish_lidar(begun_l[rid], None, arenas)
                    arenas.append(idx_l[rid])
                    plan["lidar"][rid] = (idx_l[rid], rbs)
                # (Round 5 let the LiDAR stream's pass start behind its own tables -- plan["_lidar_ready"] -- while the virtual-point
                # stream's image-space branch was still being built: that pass is made of the very conv kernels the pixel projection
                # must not share a compute unit with, LOG.md A.17 / ADVICE r5.  The main stream now waits for the whole plan; the
                # configuration is host-bound and measures the same.)
                for i, rid in enumerate(begun_m):
                    trans_param, tags = mm_inputs(i, rid)
                    stages, _, keep0, kept0, ar = native_plan.finish_nrconv(cps_m[rid], blocks, scope.guard)
                    arenas.extend(ar)
                    arenas.append(idx_m[rid])
                    plan["mm"][rid] = {"keep0": keep0, "in_indices": kept0 if active else idx_m[rid], "stages": stages,
                                       "trans_param": trans_param}
                plan["_arenas"] = arenas
                return scope.publish(plan)
            # ---- test time (spconv_backbone.py:409-442): the rot_num copies concatenated along x into ONE tensor for the LiDAR stream
            cat_native = False
            if not self.training:
                coords = []
                for i, rid in enumerate(rids):
                    c = batch_dict["voxel_coords" + rid].clone()
                    c[:, 3] += i * self.sparse_shape[2]
                    coords.append(c)
                idx_cat = torch.cat(coords).int()
                new_shape = [self.sparse_shape[0], self.sparse_shape[1], self.sparse_shape[2] * 4]
                cells = batch_size * new_shape[0] * new_shape[1] * new_shape[2]
                cat_native = bool(NATIVE_CAT_PLAN and native_plan.usable(idx_cat) and cells < (1 << 31)
                                  and all(native_plan.usable(v, blocks) for v in idx_m.values()))
            if cat_native:
                # Round 6 (VERDICT r5 "missing" #4): every chain natively -- the LiDAR chain over the concatenated [41, 1600, 5632] grid
                # (one count read for its four strided convs) and the rot_num virtual-point chains, all BEGUN before any is finished;
                # the three slabs of x_conv3 / x_conv4 / out cut with ONE host read (_plan_splits) instead of nine `nonzero`s.  Integer
                # tables overlap the previous frame; only the image-space branches wait for the guard (inside vc_plan_finish).
                cp_l = self._begin_lidar(idx_cat, batch_size, batch_dict, None, new_shape)
                cps_m = {}
                for i, rid in enumerate(rids if self.mm else []):
                    trans_param, tags = mm_inputs(i, rid)
                    cps_m[rid] = native_plan.begin(self, blocks, None, idx_m[rid], batch_size, calib, trans_param, tags,
                                                   self.layer_discard_rate, batch_dict, NRConvBlock.IMAGE_SHAPE,
                                                   input_discard_tag=(f"mm_input{rid}" if active else None), deferred=scope.deferred)
                rbs, co = self._finish_lidar(cp_l, None, arenas)
                plan["lidar"]["cat"] = (idx_cat, rbs, new_shape)
                plan["split"] = self._plan_splits(co, rids)
                for i, rid in enumerate(rids if self.mm else []):
                    trans_param, tags = mm_inputs(i, rid)
                    stages, _, keep0, kept0, ar = native_plan.finish_nrconv(cps_m[rid], blocks, scope.guard)
                    plan["mm"][rid] = {"keep0": keep0, "in_indices": kept0 if active else idx_m[rid], "stages": stages,
                                       "trans_param": trans_param}
                return scope.publish(plan)
            # operator-by-operator plans (empty / CPU tensors, VIRCONV_NATIVE_CAT_PLAN=0): the whole plan waits
            scope.guard_tables()
            if self.training:
                for rid in rids:
                    rbs, _ = self._plan_lidar(idx_l[rid], self.sparse_shape, batch_size)
                    plan["lidar"][rid] = (idx_l[rid], rbs)
            else:
                rbs, co = self._plan_lidar(idx_cat, new_shape, batch_size)
                plan["lidar"]["cat"] = (idx_cat, rbs, new_shape)
                for i, rid in enumerate(rids):
                    plan["split"][rid] = {k: self._plan_split(co[k][0], co[k][1], i) for k in ("x3", "x4", "out")}
            if self.mm:
                for i, rid in enumerate(rids):
                    idx = idx_m[rid]
                    trans_param, tags = mm_inputs(i, rid)
                    if native_plan.usable(idx, blocks):
                        # input discard (spconv_backbone.py:488-489) + the four blocks + their layer discards: one native chain plan
                        stages, _, keep0, kept0, ar = native_plan.build(self, blocks, None, idx, batch_size, calib, trans_param, tags,
                                                                        self.layer_discard_rate, batch_dict, NRConvBlock.IMAGE_SHAPE,
                                                                        input_discard_tag=(f"mm_input{rid}" if active else None),
                                                                        deferred=scope.deferred, guard=scope.guard)
                        plan["mm"][rid] = {"keep0": keep0, "in_indices": kept0 if active else idx, "stages": stages,
                                           "trans_param": trans_param}
                        continue
                    keep0 = None
                    if active:  # the MM stream also discards its input (spconv_backbone.py:488-489)
                        keep0 = _draw_keep(self.layer_discard_rate, idx.shape[0], batch_dict, f"mm_input{rid}", idx.device)
                        _, idx = ops.get_backend().gather_rows(None, idx, keep0)
                    stages, _, _, _ = _plan_nrconv_chain(blocks, idx, list(self.sparse_shape), batch_size, calib, trans_param,
                                                         tags, self.layer_discard_rate, batch_dict)
                    plan["mm"][rid] = {"keep0": keep0, "in_indices": idx, "stages": stages, "trans_param": trans_param}
        return scope.publish(plan)

    @staticmethod
    def _apply_split(tensor, split, batch_size):
        keep, idx, shape = split
        feats = ops.GatherRowsFunction.apply(tensor.features, keep) if tensor.features.is_cuda else tensor.features[keep]
        return spconv.SparseConvTensor(feats, idx, shape, batch_size)

    def forward(self, batch_dict):
        rot_num = batch_dict["transform_param"].shape[1] if "transform_param" in batch_dict else 1
        batch_size = batch_dict["batch_size"]
        strides = {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}
        rids = ["" if i == 0 else str(i) for i in range(rot_num)]
        calib = None
        if self.mm:
            calib = batch_dict["calib"]
            if not torch.is_tensor(calib):
                calib = ops.calib_tensor(calib, batch_dict["voxel_features_mm"].device)
        plan = self.build_plan(batch_dict, rids, batch_size, calib) if self.plan_ahead else None

        if self.training:
            for rid in rids:
                native = None
                if plan is not None:
                    idx, rbs = plan["lidar"][rid]
                    f_in = batch_dict["voxel_features" + rid]
                    if feature_pass.usable_8x(self, f_in, "lidar"):
                        native = feature_pass.run_8x_lidar(self, f_in, rbs)   # the whole stream: one native call per direction
                    sp = spconv.SparseConvTensor(f_in, idx, self.sparse_shape, batch_size, indice_dict=dict(rbs))
                else:
                    sp = spconv.SparseConvTensor(batch_dict["voxel_features" + rid], batch_dict["voxel_coords" + rid].int(),
                                                 self.sparse_shape, batch_size)
                if native is not None:
                    geo = [(idx, self.sparse_shape)] + [(rbs[seq[0][0].indice_key].out_indices, list(rbs[seq[0][0].indice_key].out_shape))
                                                        for seq in (self.conv2, self.conv3, self.conv4)]
                    rbo = rbs[self.conv_out[0].indice_key]
                    geo.append((rbo.out_indices, list(rbo.out_shape)))
                    x1, x2, x3, x4, out = [spconv.SparseConvTensor(f, gi, gs, batch_size, indice_dict=dict(rbs))
                                           for f, (gi, gs) in zip(native, geo)]
                else:
                    x1, x2, x3, x4, out = self._lidar_stream(sp)
                batch_dict.update({"encoded_spconv_tensor" + rid: out, "encoded_spconv_tensor_stride" + rid: 8,
                                   "multi_scale_3d_features" + rid: {"x_conv1": x1, "x_conv2": x2, "x_conv3": x3, "x_conv4": x4},
                                   "multi_scale_3d_strides" + rid: dict(strides)})
        else:
            feats = torch.cat([batch_dict["voxel_features" + rid] for rid in rids], 0)
            if plan is not None:
                idx, rbs, new_shape = plan["lidar"]["cat"]
                sp = spconv.SparseConvTensor(feats, idx, new_shape, batch_size, indice_dict=dict(rbs))
            else:
                coords = []
                for i, rid in enumerate(rids):
                    c = batch_dict["voxel_coords" + rid].clone()
                    c[:, 3] += i * self.sparse_shape[2]
                    coords.append(c)
                new_shape = [self.sparse_shape[0], self.sparse_shape[1], self.sparse_shape[2] * 4]
                sp = spconv.SparseConvTensor(feats, torch.cat(coords).int(), new_shape, batch_size)
            native = None
            if plan is not None and feature_pass.usable_8x(self, feats, "lidar"):
                native = feature_pass.run_8x_lidar(self, feats, rbs)      # the whole stream as ONE native call (eval BatchNorm folded)
            if native is not None:
                geo = [(idx, new_shape)] + [(rbs[seq[0][0].indice_key].out_indices, list(rbs[seq[0][0].indice_key].out_shape))
                                            for seq in (self.conv2, self.conv3, self.conv4)]
                rbo = rbs[self.conv_out[0].indice_key]
                geo.append((rbo.out_indices, list(rbo.out_shape)))
                x1, x2, x3, x4, out = [spconv.SparseConvTensor(f, gi, gs, batch_size, indice_dict=dict(rbs)) for f, (gi, gs) in zip(native, geo)]
            else:
                x1, x2, x3, x4, out = self._lidar_stream(sp)
            for i, rid in enumerate(rids):
                if plan is not None:
                    sp_ = plan["split"][rid]
                    o, s3, s4 = (self._apply_split(out, sp_["out"], batch_size), self._apply_split(x3, sp_["x3"], batch_size),
                                 self._apply_split(x4, sp_["x4"], batch_size))
                else:
                    o, s3, s4 = (self.decompose_tensor(out, i, batch_size), self.decompose_tensor(x3, i, batch_size),
                                 self.decompose_tensor(x4, i, batch_size))
                batch_dict.update({
                    "encoded_spconv_tensor" + rid: o,
                    "encoded_spconv_tensor_stride" + rid: 8,
                    "multi_scale_3d_features" + rid: {"x_conv1": None, "x_conv2": None, "x_conv3": s3, "x_conv4": s4},
                    "multi_scale_3d_strides" + rid: dict(strides)})

        if plan is not None and "_fwd_ready" in plan:     # everything else of the plan (the virtual-point stream's tables)
            _current_stream().wait_event(plan.pop("_fwd_ready"))
        if self.mm:
            blocks = [(self.vir_conv1, 1), (self.vir_conv2, 2), (self.vir_conv3, 4), (self.vir_conv4, 8)]
            for i, rid in enumerate(rids):
                if plan is not None:
                    pm = plan["mm"][rid]
                    f = batch_dict["voxel_features_mm" + rid]
                    native = feature_pass.run_8x_mm(self, f, pm) if feature_pass.usable_8x(self, f, "mm") else None
                    if native is not None:   # input discard + the four NRConvBlocks + layer discards: one native call per direction
                        m1, m2, m3, m4 = [spconv.SparseConvTensor(ft, st["kept_indices"] if st["keep"] is not None else st["out_indices"],
                                                                  st["out_shape"], batch_size)
                                          for ft, st in zip(native, pm["stages"])]
                    else:
                        if pm["keep0"] is not None:
                            f = ops.GatherRowsFunction.apply(f, pm["keep0"])
                        sp = spconv.SparseConvTensor(f, pm["in_indices"], self.sparse_shape, batch_size)
                        m1, m2, m3, m4 = _run_nrconv_chain(blocks, pm["stages"], sp, batch_size, calib, pm["trans_param"])
                else:
                    sp = spconv.SparseConvTensor(batch_dict["voxel_features_mm" + rid],
                                                 batch_dict["voxel_coords_mm" + rid].int(), self.sparse_shape, batch_size)
                    sp = self._discard(sp, batch_dict, f"mm_input{rid}")  # the MM stream also discards its input (:488-489)
                    trans_param = batch_dict.get("aug_param")
                    if "transform_param" in batch_dict:
                        trans_param = batch_dict["transform_param"][:, i, :]
                    m1 = self._discard(self.vir_conv1(sp, batch_size, calib, 1, None, trans_param), batch_dict, f"mm_x_conv1{rid}")
                    m2 = self._discard(self.vir_conv2(m1, batch_size, calib, 2, None, trans_param), batch_dict, f"mm_x_conv2{rid}")
                    m3 = self._discard(self.vir_conv3(m2, batch_size, calib, 4, None, trans_param), batch_dict, f"mm_x_conv3{rid}")
                    m4 = self.vir_conv4(m3, batch_size, calib, 8, None, trans_param)
                batch_dict.update({"encoded_spconv_tensor_stride_mm" + rid: 8,
                                   "multi_scale_3d_features_mm" + rid: {"x_conv1": m1, "x_conv2": m2, "x_conv3": m3, "x_conv4": m4},
                                   "multi_scale_3d_strides" + rid: dict(strides)})
        if plan is not None:
            join_plan(plan)
        return batch_dict


class HeightCompression(nn.Module):
    """First consumer of the path's output (pcdet/models/backbones_2d/map_to_bev/height_compression.py:27-31).

    SURVEY 8f rank 3: with ``BEV_PAD: 1`` in the config the BEV map is emitted WITH the zero border the first BEV conv needs --
    (B, C*D, H + 2, W + 2), border and interior written once by the same kernel (vc_to_dense_fill_padded) -- so that the
    ``nn.ZeroPad2d(1)`` in front of ``Conv2d(k3, padding=0)`` (base_bev_backbone.py:31-36), a full copy of the 36 MB-per-frame
    map, disappears: ``adapt_bev_backbone`` drops that pad module from a BaseBEVBackbone.  Default (0): the reference's layout.
    ``adapt_bev_backbone(..., height_compression=self, sparse_stem=True)`` goes one step further: the first BEV block's
    conv + BatchNorm + ReLU run HERE, on the sparse rows (virconv_amd/bev_stem.py), and ``spatial_features`` is their output."""

    def __init__(self, model_cfg=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = _cfg_get(model_cfg, "NUM_BEV_FEATURES", 256) if model_cfg is not None else 256
        self.bev_pad = int(_cfg_get(model_cfg, "BEV_PAD", 0)) if model_cfg is not None else 0
        self.bev_stem = None      # SparseBEVStem, attached by adapt_bev_backbone(sparse_stem=True)

    def forward(self, batch_dict):
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        t = batch_dict["encoded_spconv_tensor"]
        if self.bev_stem is not None:
            batch_dict["spatial_features"] = self.bev_stem(t)       # (B, 64, H, W): the first BEV conv + BN + ReLU already applied
            batch_dict["spatial_features_pad"] = 0
            return batch_dict
        sp = t.dense(pad=(self.bev_pad, self.bev_pad)) if self.bev_pad else t.dense()
        n, c, d, h, w = sp.shape
        batch_dict["spatial_features"] = sp.view(n, c * d, h, w)
        batch_dict["spatial_features_pad"] = self.bev_pad
        return batch_dict


def adapt_bev_backbone(bev_backbone: nn.Module, pad: int = 1, height_compression: Optional["HeightCompression"] = None,
                       sparse_stem: bool = False) -> nn.Module:
    """Make a BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:27-43) consume what this package's
    ``HeightCompression`` hands over.  Default: the pre-padded BEV map of ``HeightCompression(BEV_PAD=pad)`` -- the first block's
    ``nn.ZeroPad2d(1)`` is replaced by an identity (state_dict keys are unchanged: ZeroPad2d has no parameters, indices stay).
    ``sparse_stem=True`` (needs ``height_compression``): the first block's pad + conv + BatchNorm + ReLU move into
    ``height_compression`` as a ``SparseBEVStem`` over the sparse rows; the block keeps its modules (same state_dict keys, the
    parameters are still the BEV backbone's) but its forward starts behind them."""
    first = bev_backbone.blocks[0]
    assert isinstance(first[0], nn.ZeroPad2d) and tuple(first[0].padding) == (pad,) * 4, "first BEV block does not start with ZeroPad2d"
    assert isinstance(first[1], nn.Conv2d) and first[1].padding == (0, 0) and first[1].kernel_size == (2 * pad + 1,) * 2
    if sparse_stem:
        from .bev_stem import SparseBEVStem, _StemSkippingBlock
        assert height_compression is not None, "sparse_stem=True needs the HeightCompression module that will run the stem"
        block = _StemSkippingBlock(*list(first))
        bev_backbone.blocks[0] = block
        height_compression.bev_stem = SparseBEVStem(block)
        return bev_backbone
    first[0] = nn.Identity()
    return bev_backbone


__all__ = {"VirConvL8x": VirConvL8x, "VirConv8x": VirConv8x}
