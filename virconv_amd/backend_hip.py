"""HipBackend: torch-tensor wrappers over the C ABI (device memory, streams: plumbing only).

Every method takes/returns contiguous CUDA tensors, launches on torch's CURRENT stream and allocates outputs and
scratch through torch's caching allocator (the library itself never allocates, include/virconv_hip.h).
Calling any of them with CPU tensors raises: there is no CPU path in the product.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import CONV_SORTED_ROWS, OPERAND_TYPES, check, i32arr, f32arr


import ctypes
import os

# weight gradient of a conv+BN+ReLU unit on a side stream underneath its backward-input conv (fork / join inside the C call)
# node-by-node path only (the feature pass has its own weight-gradient stream schedule, see csrc/pass.hip): fork the weight gradient
# of a unit onto a side stream inside vc_post_act_block_backward.  Measured: VirConv-L bs 4 6.13-6.14 -> 6.02-6.09 ms/step, but
# VirConv8x bs 2 (small tensors, host-bound) 5.98 -> 6.82 ms/step -- the fork/join pair per unit costs more host time than the
# overlap returns.  Off by default.
UNIT_OVERLAP_DW = os.environ.get("VIRCONV_UNIT_OVERLAP_DW", "0") != "0"
# .dense() / HeightCompression as a write-once fill (vc_to_dense_fill) instead of zero-fill + scatter; "0" = the scatter form
DENSE_WRITE_ONCE = os.environ.get("VIRCONV_DENSE_WRITE_ONCE", "1") != "0"


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.VirConvError(f"{name}: expected a CUDA/HIP tensor; virconv_amd has no CPU path")
    if t.dtype != dtype:
        raise _lib.VirConvError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _stream() -> int:
    # raw hipStream_t of torch's CURRENT stream on the current device (fast path of torch.cuda.current_stream().cuda_stream)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def conv_out_shape(in_shape, ksize, stride, padding, dilation):
    return tuple((int(i) + 2 * p - d * (k - 1) - 1) // s + 1 for i, k, s, p, d in zip(in_shape, ksize, stride, padding, dilation))


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self._pinned = None
        self._pin_ev = None
        self._trace_cap, self._trace_pairs = 0, None
        if os.environ.get("VIRCONV_CONV_NW"):   # waves per block of the direct gather-GEMM (4 | 8), see conv_kernels.hip
            check(self.lib.vc_debug_set(b"conv_nw", int(os.environ["VIRCONV_CONV_NW"])), "vc_debug_set")
        if os.environ.get("VIRCONV_CONV_PACKED"):   # 0 = ignore the fragment-ordered weight images the pass executor registers (A/B)
            check(self.lib.vc_debug_set(b"conv_packed", int(os.environ["VIRCONV_CONV_PACKED"])), "vc_debug_set")
        if os.environ.get("VIRCONV_BW_ROWS"):   # weight gradient: target rows per block
            check(self.lib.vc_debug_set(b"bw_rows_per_split", int(os.environ["VIRCONV_BW_ROWS"])), "vc_debug_set")
        if os.environ.get("VIRCONV_CONV_DXS"):   # 0 = no dx shift in the gather-GEMM (A/B)
            check(self.lib.vc_debug_set(b"conv_dxs", int(os.environ["VIRCONV_CONV_DXS"])), "vc_debug_set")
        if os.environ.get("VIRCONV_CONV_V4"):   # wave-autonomous gather-GEMM: 0 never | 1 every eligible shape | 2 library table
            check(self.lib.vc_debug_set(b"conv_v4", int(os.environ["VIRCONV_CONV_V4"])), "vc_debug_set")
        # generic A/B switches for every tool that builds a backend: VIRCONV_DEBUG_SET="key=value,key=value" -> vc_debug_set
        for kv in filter(None, os.environ.get("VIRCONV_DEBUG_SET", "").split(",")):
            key, val = kv.split("=")
            check(self.lib.vc_debug_set(key.strip().encode(), int(val)), f"vc_debug_set {kv}")

    # ------------------------------------------------------------------ kernel timing for bench.py's roofline
    native_pass = True   # vc_pass_forward / vc_pass_backward are available (virconv_amd/feature_pass.py)
    native_plan = True   # vc_plan_begin / _wait / _finish are available (virconv_amd/native_plan.py)

    @staticmethod
    def stream() -> int:
        return _stream()

    @staticmethod
    def pass_overlap_dw() -> bool:
        """Feature pass: weight gradients on the side stream (joined once at the end of the sweep).  Measured on VirConv-L
        bs 4: backward 3.87-3.98 ms without, 3.27-3.45 ms with (tools/step_phases.py)."""
        return os.environ.get("VIRCONV_PASS_OVERLAP_DW", "1") != "0"

    def trace_begin(self, direction: str, ck: int, cn: int, max_records: int = 4096) -> None:
        """Bracket every launch of the gather-GEMM instantiation <CK, CN, BWD=(direction=='bwd')> with HIP events on its launch
        stream, INSIDE the library (vc_trace_begin): whichever call issues the launch -- an operator, a post_act_block unit or
        the whole feature pass -- is timed as it runs in the product path.  The active pairs of each traced table are counted
        on the device right after the launch (outside the bracket)."""
        # "dw": the weight gradient <CI, CO>; "all": every conv launch (ck, cn ignored); "bn_dx": bn_bwd_dx_pow2_kernel, the largest
        # bandwidth-bound kernel of the step (ck = channel filter, 0 = every channel count)
        assert direction in ("fwd", "bwd", "dw", "all", "bn_dx")
        self._trace_pairs = torch.zeros((max_records,), dtype=torch.int64, device="cuda")
        check(self.lib.vc_trace_begin({"fwd": 0, "bwd": 1, "dw": 2, "all": -1, "bn_dx": 3}[direction], int(ck), int(cn), int(max_records),
                                      _ptr(self._trace_pairs)), "vc_trace_begin")
        self._trace_cap = int(max_records)

    def trace_end(self):
        """-> [{ms, flops, bytes, pairs, n_out, windowed}] per traced launch (algorithmic figures: SURVEY 8d formulas)."""
        cap = getattr(self, "_trace_cap", 0)
        if not cap:
            return []
        recs = (_lib.TraceRecord * cap)()
        n = ctypes.c_int(0)
        check(self.lib.vc_trace_end(recs, cap, ctypes.byref(n)), "vc_trace_end")
        self._trace_cap, self._trace_pairs = 0, None
        out = []
        for r in recs[:n.value]:
            if r.direction == 3:   # BatchNorm backward dx: reads x and dy, writes dx (N x C fp32 each) + the per-channel vectors
                out.append({"ms": r.ms, "flops": 0.0, "bytes": 4.0 * (3 * r.n_out * r.ck + 6 * r.ck), "pairs": 0, "n_out": int(r.n_out),
                            "windowed": False, "dir": "bn_dx", "ck": int(r.ck), "cn": int(r.cn), "kv": 0, "t0_ms": float(r.t0_ms)})
                continue
            out.append({"ms": r.ms, "flops": 2.0 * r.pairs * r.ck * r.cn,
                        "bytes": 4.0 * (r.n_src * r.ck + r.n_out * r.cn + r.kv * r.ck * r.cn) + 4.0 * r.kv * r.n_out,
                        "pairs": int(r.pairs), "n_out": int(r.n_out), "windowed": bool(r.windowed),
                        "dir": ("fwd", "bwd", "dw")[r.direction], "ck": int(r.ck), "cn": int(r.cn), "kv": int(r.kv),
                        "t0_ms": float(r.t0_ms)})
        return out

    def _read_count(self, dev_scalar: torch.Tensor) -> int:
        """Device int32 -> host: async copy into a pinned slot, then POLL the event.  (A blocking .item() goes through
        pageable-memory staging + a blocking stream sync; with an RCCL communicator alive in the process that costs
        ~0.8 ms instead of ~0.25 ms -- measured, 4 reads per step.)"""
        if self._pinned is None:
            self._pinned = torch.empty((16,), dtype=torch.int32).pin_memory()
            self._pin_ev = torch.cuda.Event()
        self._pinned[:1].copy_(dev_scalar, non_blocking=True)
        self._pin_ev.record()
        while not self._pin_ev.query():
            pass
        return int(self._pinned[0])

    # ------------------------------------------------------------------ rulebooks
    def subm_rulebook(self, indices: torch.Tensor, spatial_shape: Sequence[int], ksize, dilation, want_rep: bool):
        """-> pair_fwd (KV, N) int32, rep (N,) int32 or None."""
        indices = _need(indices, torch.int32, "indices")
        n, ndim = indices.shape[0], indices.shape[1] - 1
        kv = int(np.prod(ksize))
        dev = indices.device
        pair = torch.empty((kv, n), dtype=torch.int32, device=dev)
        rep = torch.empty((n,), dtype=torch.int32, device=dev) if want_rep else None
        if n == 0:
            return pair, rep
        ws_bytes = self.lib.vc_hash_workspace_bytes(n)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        shp = i32arr(spatial_shape)
        st = _stream()
        check(self.lib.vc_hash_build(_ptr(indices), n, ndim, shp, _ptr(ws), ws_bytes, st), "vc_hash_build")
        check(self.lib.vc_subm_rulebook(_ptr(indices), n, ndim, shp, i32arr(ksize), i32arr(dilation), _ptr(ws), ws_bytes,
                                        _ptr(pair), _ptr(rep), st), "vc_subm_rulebook")
        return pair, rep

    def sparse_rulebook_begin(self, indices: torch.Tensor, spatial_shape, batch_size: int, ksize, stride, padding, dilation):
        """First half of a strided-conv rulebook: mark the output cells, count them, start the count's trip to the host.
        The caller issues whatever other geometry kernels it has before `sparse_rulebook_finish` waits for the count -- the
        one data-dependent size of a strided conv then costs no idle time on the plan stream."""
        indices = _need(indices, torch.int32, "indices")
        n, ndim = indices.shape[0], indices.shape[1] - 1
        out_shape = conv_out_shape(spatial_shape, ksize, stride, padding, dilation)
        dev = indices.device
        oshp = i32arr(out_shape)
        ws_bytes = self.lib.vc_spconv_workspace_bytes(batch_size, ndim, oshp)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        n_out_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        ks, sd, pd, dl = i32arr(ksize), i32arr(stride), i32arr(padding), i32arr(dilation)
        check(self.lib.vc_spconv_mark_count(_ptr(indices), n, ndim, batch_size, oshp, ks, sd, pd, dl, _ptr(ws), ws_bytes,
                                            _ptr(n_out_dev), _stream()), "vc_spconv_mark_count")
        if self._pinned is None:
            self._pinned = torch.empty((16,), dtype=torch.int32).pin_memory()
            self._pin_ev = torch.cuda.Event()
        if getattr(self, "_pin_evs", None) is None:
            self._pin_evs = [torch.cuda.Event() for _ in range(8)]
            self._pin_next = 0
        slot = 1 + self._pin_next % 8      # slot 0 belongs to _read_count
        self._pin_next += 1
        self._pinned[slot:slot + 1].copy_(n_out_dev, non_blocking=True)
        ev = self._pin_evs[slot - 1]
        ev.record()
        return {"indices": indices, "n": n, "ndim": ndim, "batch_size": batch_size, "out_shape": out_shape, "oshp": oshp,
                "geom": (ks, sd, pd, dl), "kv": int(np.prod(ksize)), "ws": ws, "ws_bytes": ws_bytes, "n_out_dev": n_out_dev,
                "slot": slot, "event": ev}

    def sparse_rulebook_finish(self, h):
        """-> out_indices (M, ndim+1) int32 ascending, out_shape, pair_fwd (KV, M), pair_bwd (KV, N)."""
        while not h["event"].query():
            pass
        n_out = int(self._pinned[h["slot"]])  # the one host sync of a strided conv (data-dependent output size)
        indices, n, ndim, dev = h["indices"], h["n"], h["ndim"], h["indices"].device
        ks, sd, pd, dl = h["geom"]
        out_indices = torch.empty((n_out, ndim + 1), dtype=torch.int32, device=dev)
        pair_fwd = torch.empty((h["kv"], n_out), dtype=torch.int32, device=dev)
        pair_bwd = torch.empty((h["kv"], n), dtype=torch.int32, device=dev)
        check(self.lib.vc_spconv_emit_pairs(_ptr(indices), n, ndim, h["batch_size"], h["oshp"], ks, sd, pd, dl, _ptr(h["ws"]),
                                            h["ws_bytes"], n_out, _ptr(out_indices), _ptr(pair_fwd), _ptr(pair_bwd), _stream()),
              "vc_spconv_emit_pairs")
        return out_indices, h["out_shape"], pair_fwd, pair_bwd

    def sparse_rulebook_chain(self, indices: torch.Tensor, spatial_shape, batch_size: int, geoms):
        """Strided convs applied one after the other with nothing in between that changes the active set (the backbone's
        stage 2 -> 3 -> 4 -> conv_out when no layer discard is active; SubM convs keep the set).  geoms: [(ksize, stride, padding,
        dilation), ...].  Stage 1 and the coordinate emission of EVERY level run before the host reads anything: level l + 1
        marks from level l's coordinates with their count still on the device; buffers are sized by capacity (an input row
        reaches at most prod(ceil(k / s)) output cells).  ONE host read returns all counts, then the pair tables are built.
        -> [(out_indices, out_shape, pair_fwd, pair_bwd), ...] exactly as `sparse_rulebook` would return level by level."""
        indices = _need(indices, torch.int32, "indices")
        dev, ndim = indices.device, indices.shape[1] - 1
        st = _stream()
        levels = []
        cur, cur_cap, cur_n_dev, shape = indices, indices.shape[0], None, tuple(int(v) for v in spatial_shape)
        counts = torch.zeros((len(geoms),), dtype=torch.int32, device=dev)
        for li, (ksize, stride, padding, dilation) in enumerate(geoms):
            out_shape = conv_out_shape(shape, ksize, stride, padding, dilation)
            oshp = i32arr(out_shape)
            ws_bytes = self.lib.vc_spconv_workspace_bytes(batch_size, ndim, oshp)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            ks, sd, pd, dl = i32arr(ksize), i32arr(stride), i32arr(padding), i32arr(dilation)
            n_out_dev = counts[li:li + 1]
            if cur_n_dev is None:
                check(self.lib.vc_spconv_mark_count(_ptr(cur), cur_cap, ndim, batch_size, oshp, ks, sd, pd, dl, _ptr(ws), ws_bytes,
                                                    _ptr(n_out_dev), st), "vc_spconv_mark_count")
            else:
                check(self.lib.vc_spconv_mark_count_dev(_ptr(cur), cur_cap, _ptr(cur_n_dev), ndim, batch_size, oshp, ks, sd, pd, dl,
                                                        _ptr(ws), ws_bytes, _ptr(n_out_dev), st), "vc_spconv_mark_count_dev")
            reach = int(np.prod([-(-int(k) // int(s)) for k, s in zip(ksize, stride)]))
            cap = int(min(cur_cap * reach, batch_size * int(np.prod(out_shape))))
            out_idx = torch.empty((cap, ndim + 1), dtype=torch.int32, device=dev)
            check(self.lib.vc_spconv_emit_indices(ndim, batch_size, oshp, _ptr(ws), ws_bytes, cap, _ptr(out_idx), st),
                  "vc_spconv_emit_indices")
            levels.append({"in": cur, "ws": ws, "ws_bytes": ws_bytes, "oshp": oshp, "geom": (ks, sd, pd, dl), "out_shape": out_shape,
                           "out_cap": out_idx, "kv": int(np.prod(ksize))})
            cur, cur_cap, cur_n_dev, shape = out_idx, cap, n_out_dev, out_shape
        if self._pinned is None:
            self._pinned = torch.empty((16,), dtype=torch.int32).pin_memory()
            self._pin_ev = torch.cuda.Event()
        assert len(geoms) <= 7
        self._pinned[9:9 + len(geoms)].copy_(counts, non_blocking=True)   # slots 0-8 belong to the per-conv reads
        self._pin_ev.record()
        while not self._pin_ev.query():
            pass
        n_outs = [int(v) for v in self._pinned[9:9 + len(geoms)]]
        res, n_in = [], indices.shape[0]
        for lv, n_out in zip(levels, n_outs):
            ks, sd, pd, dl = lv["geom"]
            src = lv["in"][:n_in]                      # a contiguous prefix of the capacity buffer
            out_idx = lv["out_cap"][:n_out]
            pf = torch.empty((lv["kv"], n_out), dtype=torch.int32, device=dev)
            pb = torch.empty((lv["kv"], n_in), dtype=torch.int32, device=dev)
            check(self.lib.vc_spconv_pairs(_ptr(src), n_in, ndim, batch_size, lv["oshp"], ks, sd, pd, dl, _ptr(lv["ws"]),
                                           lv["ws_bytes"], n_out, _ptr(pf), _ptr(pb), st), "vc_spconv_pairs")
            res.append((out_idx, lv["out_shape"], pf, pb, src))
            n_in = n_out
        return res

    def sparse_rulebook(self, indices: torch.Tensor, spatial_shape, batch_size: int, ksize, stride, padding, dilation):
        """-> out_indices (M, ndim+1) int32 ascending, out_shape, pair_fwd (KV, M), pair_bwd (KV, N)."""
        return self.sparse_rulebook_finish(self.sparse_rulebook_begin(indices, spatial_shape, batch_size, ksize, stride, padding,
                                                                      dilation))

    # ------------------------------------------------------------------ convolution
    def row_order(self, tbl: torch.Tensor, rep: Optional[torch.Tensor] = None, centre: int = -1,
                  window: int = 1024) -> torch.Tensor:
        """(KV, n) pair table -> (n,) int32 permutation grouping rows with equal active-offset sets (vc_row_order)."""
        kv, n = tbl.shape
        order = torch.empty((n,), dtype=torch.int32, device=tbl.device)
        check(self.lib.vc_row_order(_ptr(tbl), n, kv, _ptr(rep), centre if rep is not None else -1, window, _ptr(order),
                                    _stream()), "vc_row_order")
        return order

    def rep_order(self, rep: torch.Tensor) -> torch.Tensor:
        """(n,) int32 stable partition of the rows, representatives (rep[r] == r) first (vc_rep_order)."""
        rep = _need(rep, torch.int32, "rep")
        n = rep.shape[0]
        order = torch.empty((n,), dtype=torch.int32, device=rep.device)
        ws_bytes = self.lib.vc_rep_order_workspace_bytes(n)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=rep.device)
        check(self.lib.vc_rep_order(_ptr(rep), n, _ptr(order), _ptr(ws), ws_bytes, _stream()), "vc_rep_order")
        return order

    def conv_forward(self, x: torch.Tensor, weight: torch.Tensor, pair_fwd: torch.Tensor,
                     order: Optional[torch.Tensor] = None, operand: str = "f32", sorted_rows: bool = False,
                     interleaved_rows: Optional[int] = None) -> torch.Tensor:
        """`sorted_rows`: VC_CONV_SORTED_ROWS hint (table rows and the rows they gather are in ascending coordinate order):
        the kernel stages the gathers through LDS row windows.  Results are bit-identical with and without it.
        `interleaved_rows` = n_in (round-3 experiment, VC_CONV_SRC_INTERLEAVED): `x` is the output of `interleave_rows`."""
        x = _need(x, torch.float32, "features")
        weight = _need(weight, torch.float32, "weight")
        pair_fwd = _need(pair_fwd, torch.int32, "pair_fwd")
        kv, n_out = pair_fwd.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        n_in = x.shape[0] if interleaved_rows is None else int(interleaved_rows)
        assert weight.numel() == cout * kv * cin and x.numel() >= n_in * cin
        y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
        flags = (CONV_SORTED_ROWS if sorted_rows else 0) | (2 if interleaved_rows is not None else 0)
        check(self.lib.vc_conv_forward(_ptr(x), n_in, _ptr(pair_fwd), n_out, kv, _ptr(weight), cin, cout,
                                       _ptr(order), OPERAND_TYPES[operand], flags, _ptr(y), _stream()), "vc_conv_forward")
        return y

    @staticmethod
    def interleave_rows(x: torch.Tensor) -> torch.Tensor:
        """(n, c) row-major -> the 16-row interleaved layout of VC_CONV_SRC_INTERLEAVED: [ceil(n / 16)][c / 4][16][4], zero rows
        appended up to a multiple of 16 (plain torch ops: the experiment measures the gather, not this copy)."""
        n, c = x.shape
        g = (n + 15) // 16
        xp = torch.zeros((g * 16, c), dtype=x.dtype, device=x.device)
        xp[:n] = x
        return xp.view(g, 16, c // 4, 4).permute(0, 2, 1, 3).contiguous()

    def conv_epilogue_supported(self, n_in: int, cin: int, cout: int, kv: int, operand: str = "f32") -> bool:
        return bool(self.lib.vc_conv_epilogue_supported(n_in, cin, cout, kv, OPERAND_TYPES[operand]))

    def conv_forward_stats(self, x: torch.Tensor, weight: torch.Tensor, pair_fwd: torch.Tensor,
                           order: Optional[torch.Tensor] = None, sorted_rows: bool = False):
        """Forward conv that also emits the per-block BatchNorm partial sums (VC_EPI_STATS) -> (y_raw, partial)."""
        x = _need(x, torch.float32, "features")
        weight = _need(weight, torch.float32, "weight")
        pair_fwd = _need(pair_fwd, torch.int32, "pair_fwd")
        kv, n_out = pair_fwd.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
        flags = CONV_SORTED_ROWS if sorted_rows else 0
        partial = torch.empty((self.lib.vc_conv_stats_partial_floats(x.shape[0], n_out, cin, cout, kv, flags),),
                              dtype=torch.float32, device=x.device)
        check(self.lib.vc_conv_forward_epilogue(_ptr(x), x.shape[0], _ptr(pair_fwd), n_out, kv, _ptr(weight), cin, cout,
                                                _ptr(order), 1, flags, _ptr(partial), None, None, None, None, 0.0, 0,
                                                _ptr(y), _stream()), "vc_conv_forward_epilogue")
        return y, partial

    def conv_forward_affine(self, x: torch.Tensor, weight: torch.Tensor, pair_fwd: torch.Tensor, order, mean, var, gamma,
                            beta, eps: float, relu: bool, sorted_rows: bool = False) -> torch.Tensor:
        """conv + eval-mode BatchNorm (+ReLU) in one launch (VC_EPI_AFFINE)."""
        x = _need(x, torch.float32, "features")
        weight = _need(weight, torch.float32, "weight")
        pair_fwd = _need(pair_fwd, torch.int32, "pair_fwd")
        kv, n_out = pair_fwd.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
        check(self.lib.vc_conv_forward_epilogue(_ptr(x), x.shape[0], _ptr(pair_fwd), n_out, kv, _ptr(weight), cin, cout,
                                                _ptr(order), 2, CONV_SORTED_ROWS if sorted_rows else 0, None, _ptr(mean),
                                                _ptr(var), _ptr(gamma), _ptr(beta), float(eps), 1 if relu else 0, _ptr(y),
                                                _stream()),
              "vc_conv_forward_epilogue")
        return y

    def conv_backward_input(self, dy: torch.Tensor, weight: torch.Tensor, tbl: torch.Tensor, n_in: int, mirror: bool,
                            centre: int = -1, rep: Optional[torch.Tensor] = None,
                            order: Optional[torch.Tensor] = None, operand: str = "f32",
                            group_ws: Optional[torch.Tensor] = None, sorted_rows: bool = False,
                            grp_plan: Optional[torch.Tensor] = None, grp: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`grp_plan`: the duplicate-pixel table's rows sorted by representative (group_plan): the group sum runs in that
        fixed order (vc_group_sum_sorted); built on the fly when absent.  `group_ws` is ignored (kept for call compatibility)."""
        dy = _need(dy, torch.float32, "grad_out")
        weight = _need(weight, torch.float32, "weight")
        tbl = _need(tbl, torch.int32, "pair table")
        kv = tbl.shape[0]
        assert tbl.shape[1] == n_in
        cout, cin = weight.shape[0], weight.shape[-1]
        dx = torch.empty((n_in, cin), dtype=torch.float32, device=dy.device)
        src, src_centre = dy, None
        if rep is not None:
            rep = _need(rep, torch.int32, "rep")
        if rep is not None and grp is not None:          # group sum computed by the caller (it also feeds the weight gradient)
            src, src_centre = grp, dy
        elif rep is not None:
            # no plan yet (a rulebook built under no_grad, a direct operator call): build it here -- the group sum always runs in
            # the fixed order of a plan (the order-free fixed-point sum of rounds 1-2 is an experiment build only)
            if (cout & (cout - 1)) != 0:
                raise _lib.VirConvError("duplicate-pixel backward: the output channel count must be a power of two")
            plan = grp_plan if grp_plan is not None else self.group_plan(rep)
            src, src_centre = self.group_sum_sorted(dy, plan), dy
        check(self.lib.vc_conv_backward_input(_ptr(src), _ptr(src_centre), dy.shape[0], _ptr(tbl), n_in, kv, _ptr(weight),
                                              cin, cout, 1 if mirror else 0, centre if rep is not None else -1,
                                              _ptr(rep), _ptr(order), OPERAND_TYPES[operand],
                                              CONV_SORTED_ROWS if sorted_rows else 0, _ptr(dx), _stream()),
              "vc_conv_backward_input")
        return dx

    def conv_backward_weight(self, x: torch.Tensor, dy: torch.Tensor, pair_fwd: torch.Tensor, weight_shape,
                             stream: Optional[int] = None, keep_alive: Optional[list] = None,
                             operand: str = "f32", rep: Optional[torch.Tensor] = None, centre: int = -1,
                             dy_grp: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`stream` (raw hipStream_t) overrides torch's current stream for the launches; the caller joins the streams and
        MUST pass `keep_alive`: the scratch buffer is appended to it so that torch's allocator (which only knows about the
        current stream) cannot hand its memory to another tensor before the join.
        `rep` / `centre` / `dy_grp` (duplicate-pixel tables): the non-centre offsets walk the representatives only, against the
        group-summed gradient (vc_conv_backward_weight_dup)."""
        assert stream is None or keep_alive is not None
        x = _need(x, torch.float32, "features")
        dy = _need(dy, torch.float32, "grad_out")
        pair_fwd = _need(pair_fwd, torch.int32, "pair_fwd")
        kv, n_out = pair_fwd.shape
        cout, cin = int(weight_shape[0]), int(weight_shape[-1])
        dw = torch.empty(tuple(weight_shape), dtype=torch.float32, device=x.device)
        ws_bytes = self.lib.vc_conv_backward_weight_workspace_bytes(n_out, kv, cin, cout)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
        if rep is not None and dy_grp is not None:
            rep = _need(rep, torch.int32, "rep")
            dy_grp = _need(dy_grp, torch.float32, "dy_grp")
            check(self.lib.vc_conv_backward_weight_dup(_ptr(x), _ptr(dy), _ptr(dy_grp), _ptr(rep), int(centre), _ptr(pair_fwd), n_out,
                                                       kv, cin, cout, OPERAND_TYPES[operand], _ptr(dw), _ptr(ws), ws_bytes,
                                                       _stream() if stream is None else stream),
                  "vc_conv_backward_weight_dup")
        else:
            check(self.lib.vc_conv_backward_weight(_ptr(x), _ptr(dy), _ptr(pair_fwd), n_out, kv, cin, cout,
                                                   OPERAND_TYPES[operand], _ptr(dw), _ptr(ws), ws_bytes,
                                                   _stream() if stream is None else stream),
                  "vc_conv_backward_weight")
        if keep_alive is not None:
            keep_alive.extend((ws, x, dy, pair_fwd, rep, dy_grp))
        return dw

    # ------------------------------------------------------------------ RoI grid pooling
    def voxel_index_build(self, indices: torch.Tensor, batch_size: int, spatial_shape) -> torch.Tensor:
        """Occupancy bitmap + coordinate hash over (N, 4) [b, z, y, x] -> opaque workspace tensor (vc_voxel_index_build)."""
        indices = _need(indices, torch.int32, "indices")
        assert indices.shape[1] == 4 and len(spatial_shape) == 3
        shp = i32arr(spatial_shape)
        nbytes = self.lib.vc_voxel_index_workspace_bytes(indices.shape[0], int(batch_size), shp)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=indices.device)
        check(self.lib.vc_voxel_index_build(_ptr(indices), indices.shape[0], int(batch_size), shp, _ptr(ws), nbytes,
                                            _stream()), "vc_voxel_index_build")
        return ws

    def voxel_query(self, ws: torch.Tensor, n: int, batch_size: int, spatial_shape, xyz: torch.Tensor,
                    new_xyz: torch.Tensor, new_coords: torch.Tensor, max_range, radius: float, nsample: int):
        xyz = _need(xyz, torch.float32, "xyz")
        new_xyz = _need(new_xyz, torch.float32, "new_xyz")
        new_coords = _need(new_coords, torch.int32, "new_coords")
        m = new_coords.shape[0]
        assert xyz.shape == (n, 3) and new_xyz.shape == (m, 3) and new_coords.shape[1] == 4
        idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
        empty = torch.empty((m,), dtype=torch.uint8, device=xyz.device)
        zr, yr, xr = (int(v) for v in max_range)
        check(self.lib.vc_voxel_query(_ptr(ws), ws.numel(), n, int(batch_size), i32arr(spatial_shape), _ptr(xyz),
                                      _ptr(new_xyz), _ptr(new_coords), m, zr, yr, xr, float(radius), int(nsample),
                                      _ptr(idx), _ptr(empty), _stream()), "vc_voxel_query")
        return idx, empty.bool()

    def group_points(self, features: torch.Tensor, features_batch_cnt: torch.Tensor, idx: torch.Tensor,
                     idx_batch_cnt: torch.Tensor) -> torch.Tensor:
        features = _need(features, torch.float32, "features")
        idx = _need(idx, torch.int32, "idx")
        fbc = _need(features_batch_cnt, torch.int32, "features_batch_cnt")
        ibc = _need(idx_batch_cnt, torch.int32, "idx_batch_cnt")
        m, nsample = idx.shape
        c = features.shape[1]
        out = torch.empty((m, c, nsample), dtype=torch.float32, device=features.device)
        check(self.lib.vc_group_points(ibc.shape[0], m, c, nsample, _ptr(features), _ptr(fbc), _ptr(idx), _ptr(ibc),
                                       _ptr(out), _stream()), "vc_group_points")
        return out

    def group_points_grad(self, grad_out: torch.Tensor, idx: torch.Tensor, idx_batch_cnt: torch.Tensor,
                          features_batch_cnt: torch.Tensor, n: int) -> torch.Tensor:
        grad_out = _need(grad_out, torch.float32, "grad_out")
        idx = _need(idx, torch.int32, "idx")
        fbc = _need(features_batch_cnt, torch.int32, "features_batch_cnt")
        ibc = _need(idx_batch_cnt, torch.int32, "idx_batch_cnt")
        m, c, nsample = grad_out.shape
        gf = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
        check(self.lib.vc_group_points_grad(ibc.shape[0], m, c, n, nsample, _ptr(grad_out), _ptr(idx), _ptr(ibc), _ptr(fbc),
                                            _ptr(gf), _stream()), "vc_group_points_grad")
        return gf

    # ------------------------------------------------------------------ projection / discard / dense
    def project_uv(self, indices: torch.Tensor, calib: torch.Tensor, trans: Optional[torch.Tensor], batch_size: int,
                   stride: int, want_depth: bool = False):
        indices = _need(indices, torch.int32, "indices")
        calib = _need(calib, torch.float32, "calib")
        assert indices.shape[1] == 4 and calib.shape == (batch_size, 33)
        if trans is not None:
            trans = _need(trans, torch.float32, "trans_param")
            assert trans.shape == (batch_size, 3)
        n = indices.shape[0]
        dev = indices.device
        params = torch.empty((batch_size, 32), dtype=torch.float32, device=dev)
        uv = torch.empty((n, 3), dtype=torch.int32, device=dev)
        depth = torch.empty((n,), dtype=torch.float32, device=dev) if want_depth else None
        st = _stream()
        check(self.lib.vc_project_prepare(_ptr(calib), _ptr(trans), batch_size, _ptr(params), st), "vc_project_prepare")
        check(self.lib.vc_project_uv(_ptr(indices), n, _ptr(params), batch_size, int(stride), _ptr(uv), _ptr(depth), st),
              "vc_project_uv")
        return uv, depth

    def gather_rows(self, features: Optional[torch.Tensor], indices: Optional[torch.Tensor], keep: torch.Tensor):
        keep = _need(keep, torch.int64, "keep")
        if features is None:  # index-only gather (geometry plan)
            indices = _need(indices, torch.int32, "indices")
            io = torch.empty((keep.shape[0], indices.shape[1]), dtype=torch.int32, device=indices.device)
            check(self.lib.vc_gather_rows(None, _ptr(indices), 0, indices.shape[1], _ptr(keep), keep.shape[0], None,
                                          _ptr(io), _stream()), "vc_gather_rows")
            return None, io
        features = _need(features, torch.float32, "features")
        nk, c = keep.shape[0], features.shape[1]
        fo = torch.empty((nk, c), dtype=torch.float32, device=features.device)
        io, icols = None, 0
        if indices is not None:
            indices = _need(indices, torch.int32, "indices")
            icols = indices.shape[1]
            io = torch.empty((nk, icols), dtype=torch.int32, device=features.device)
        check(self.lib.vc_gather_rows(_ptr(features), _ptr(indices), c, icols, _ptr(keep), nk, _ptr(fo), _ptr(io),
                                      _stream()), "vc_gather_rows")
        return fo, io

    def random_keep(self, n: int, n_keep: int, seed: int, device) -> torch.Tensor:
        """perm[:n_keep] of a pseudo-random permutation of range(n), evaluated point-wise on the device (vc_random_keep)."""
        keep = torch.empty((n_keep,), dtype=torch.int64, device=device)
        check(self.lib.vc_random_keep(n, n_keep, seed & 0xFFFFFFFFFFFFFFFF, _ptr(keep), _stream()), "vc_random_keep")
        return keep

    def scatter_rows(self, grad_out: torch.Tensor, keep: torch.Tensor, n_in: int) -> torch.Tensor:
        grad_out = _need(grad_out, torch.float32, "grad_out")
        keep = _need(keep, torch.int64, "keep")
        c = grad_out.shape[1]
        gi = torch.empty((n_in, c), dtype=torch.float32, device=grad_out.device)
        check(self.lib.vc_scatter_rows(_ptr(grad_out), c, _ptr(keep), keep.shape[0], n_in, _ptr(gi), _stream()),
              "vc_scatter_rows")
        return gi

    def to_dense(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int, pad=(0, 0)) -> torch.Tensor:
        """`pad` = (pad_h, pad_w): zero border around the last two axes, written by the same pass (vc_to_dense_fill_padded)."""
        features = _need(features, torch.float32, "features")
        indices = _need(indices, torch.int32, "indices")
        n, c = features.shape
        ndim = indices.shape[1] - 1
        shp = i32arr(spatial_shape)
        ws_bytes = self.lib.vc_to_dense_fill_workspace_bytes(batch_size, ndim, shp)
        if (pad[0] or pad[1]) and not (c <= 128 and ws_bytes <= (1 << 30)):
            # wide or huge maps: the write-once kernel does not serve them -- unpadded map, then a zero border
            return torch.nn.functional.pad(self.to_dense(features, indices, spatial_shape, batch_size), (pad[1], pad[1], pad[0], pad[0]))
        if pad[0] or pad[1]:
            out_shape = tuple(int(v) for v in spatial_shape[:-2]) + (int(spatial_shape[-2]) + 2 * pad[0], int(spatial_shape[-1]) + 2 * pad[1])
            dense = torch.empty((batch_size, c) + out_shape, dtype=torch.float32, device=features.device)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=features.device)
            check(self.lib.vc_to_dense_fill_padded(_ptr(features), _ptr(indices), n, c, ndim, batch_size, shp, int(pad[0]),
                                                   int(pad[1]), _ptr(dense), _ptr(ws), ws_bytes, _stream()),
                  "vc_to_dense_fill_padded")
            return dense
        if DENSE_WRITE_ONCE and ws_bytes <= (1 << 30) and c <= 128:
            # write-once fill (vc_to_dense_fill): no zero-fill of the (B, C, *spatial) output, one coalesced pass
            dense = torch.empty((batch_size, c) + tuple(int(s) for s in spatial_shape), dtype=torch.float32,
                                device=features.device)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=features.device)
            check(self.lib.vc_to_dense_fill(_ptr(features), _ptr(indices), n, c, ndim, batch_size, shp, _ptr(dense), _ptr(ws),
                                            ws_bytes, _stream()), "vc_to_dense_fill")
            return dense
        dense = torch.zeros((batch_size, c) + tuple(int(s) for s in spatial_shape), dtype=torch.float32,
                            device=features.device)
        check(self.lib.vc_to_dense(_ptr(features), _ptr(indices), n, c, ndim, batch_size, shp, _ptr(dense), _stream()),
              "vc_to_dense")
        return dense

    def bev_stem_conv(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int, w_passes, cout: int,
                      scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, relu: bool = False,
                      want_nhwc: bool = False, want_pairs: bool = False, ksize=None):
        """First BEV conv on the sparse rows (base_bev_backbone.py:31-38 over height_compression.py:27-31):
        features (n, C) at indices (n, 4) [b, z, y, x] of a (D, H, W) grid -> (B, cout, H, W) = Conv2d(C * D -> cout, k, pad k // 2)
        of the height-compressed map, then y * scale + shift (BatchNorm) and ReLU folded into the layout pass.
        `w_passes`: the conv weight cut into passes of <= 32 kernel offsets, each (C, kv_pass, cout) contiguous (see
        bev_stem.pack_stem_weight).  No host synchronisation: the pair table covers every BEV cell."""
        features = _need(features, torch.float32, "features")
        indices = _need(indices, torch.int32, "indices")
        n, c = features.shape
        D, H, W = (int(v) for v in spatial_shape)
        dev = features.device
        cells = batch_size * H * W
        kv_total = sum(int(w.shape[1]) for w in w_passes)
        assert kv_total % D == 0
        k2 = kv_total // D
        if ksize is not None:      # (ky, kx) of the Conv2d (a non-square kernel cannot be inferred from the offset count)
            ky, kx = int(ksize[0]), int(ksize[1])
            assert ky * kx == k2, (ky, kx, k2)
        else:
            ky = kx = int(round(k2 ** 0.5))
            assert ky * kx == k2, "bev_stem_conv: non-square kernel -- pass ksize=(ky, kx)"
        assert ky * kx == k2, "square 2-D kernels only"
        shp = i32arr((D, H, W))
        st = _stream()
        pair = torch.empty((kv_total, cells), dtype=torch.int32, device=dev)
        ws_bytes = self.lib.vc_bev_pairs_workspace_bytes(batch_size, shp)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        check(self.lib.vc_bev_pairs(_ptr(indices), n, batch_size, shp, ky, kx, _ptr(pair), _ptr(ws), ws_bytes, st), "vc_bev_pairs")
        acc, k0 = None, 0
        for w in w_passes:
            w = _need(w, torch.float32, "stem weight pass")
            kvp = int(w.shape[1])
            assert w.shape[0] == c and w.shape[2] == cout and kvp <= 32
            tbl = pair[k0:k0 + kvp]
            out = torch.empty((cells, cout), dtype=torch.float32, device=dev)
            if acc is None:
                # the backward-input form of the gather-GEMM reads its weight as (source channels, offsets, output channels):
                # exactly the layout of a pass
                check(self.lib.vc_conv_backward_input(_ptr(features), None, n, _ptr(tbl), cells, kvp, _ptr(w), cout, c, 0, -1, None,
                                                      None, OPERAND_TYPES["f32"], 0, _ptr(out), st), "vc_conv_backward_input")
            else:
                check(self.lib.vc_conv_backward_input_epilogue(_ptr(features), None, n, _ptr(tbl), cells, kvp, _ptr(w), cout, c, 0, -1,
                                                               None, None, 0, _ptr(acc), cout, 0, None, None, None, None, None, 0.0, 0,
                                                               None, _ptr(out), st), "vc_conv_backward_input_epilogue")
            acc, k0 = out, k0 + kvp
        if want_nhwc:
            return (acc, pair) if want_pairs else acc
        dense = torch.empty((batch_size, cout, H, W), dtype=torch.float32, device=dev)
        check(self.lib.vc_nhwc_to_nchw(_ptr(acc), batch_size, H * W, cout, _ptr(scale), _ptr(shift), 1 if relu else 0, _ptr(dense), st),
              "vc_nhwc_to_nchw")
        return dense

    def bev_stem_conv_backward(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int, w_passes, cout: int,
                               pair: torch.Tensor, gy: torch.Tensor, need_dx: bool, need_dw: bool, ksize=None):
        """Backward of `bev_stem_conv(..., want_nhwc=True)`: gy (cells, cout) -> (d features (n, C) | None, per-pass weight gradients
        [(cout, kv_pass, C), ...] | None).  dX: the forward-form gather-GEMM of gy over the transposed table (vc_bev_pairs_backward;
        a pass (C, kv_pass, cout) is exactly the (output channels, offsets, source channels) layout that kernel reads);
        dW: the weight-gradient kernel over the forward table `pair` (kept from the forward call)."""
        features = _need(features, torch.float32, "features")
        indices = _need(indices, torch.int32, "indices")
        gy = _need(gy, torch.float32, "grad_out")
        n, c = features.shape
        D, H, W = (int(v) for v in spatial_shape)
        kv_total = int(pair.shape[0])
        k2 = kv_total // D
        if ksize is not None:      # (ky, kx) of the Conv2d (a non-square kernel cannot be inferred from the offset count)
            ky, kx = int(ksize[0]), int(ksize[1])
            assert ky * kx == k2, (ky, kx, k2)
        else:
            ky = kx = int(round(k2 ** 0.5))
            assert ky * kx == k2, "bev_stem_conv_backward: non-square kernel -- pass ksize=(ky, kx)"
        dx, dws = None, None
        if need_dx:
            pair_bwd = torch.empty((kv_total, n), dtype=torch.int32, device=features.device)
            check(self.lib.vc_bev_pairs_backward(_ptr(indices), n, batch_size, i32arr((D, H, W)), ky, kx, _ptr(pair_bwd), _stream()),
                  "vc_bev_pairs_backward")
            k0 = 0
            for w in w_passes:
                kvp = int(w.shape[1])
                part = self.conv_forward(gy, w, pair_bwd[k0:k0 + kvp])       # (n, C)
                dx = part if dx is None else dx + part
                k0 += kvp
        if need_dw:
            dws, k0 = [], 0
            for w in w_passes:
                kvp = int(w.shape[1])
                dws.append(self.conv_backward_weight(features, gy, pair[k0:k0 + kvp], (cout, kvp, c)))
                k0 += kvp
        return dx, dws

    def from_dense(self, dense: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int, pad=(0, 0)) -> torch.Tensor:
        dense = _need(dense, torch.float32, "dense")
        indices = _need(indices, torch.int32, "indices")
        n, c = indices.shape[0], dense.shape[1]
        ndim = indices.shape[1] - 1
        f = torch.empty((n, c), dtype=torch.float32, device=dense.device)
        check(self.lib.vc_from_dense_padded(_ptr(dense), _ptr(indices), n, c, ndim, batch_size, i32arr(spatial_shape),
                                            int(pad[0]), int(pad[1]), _ptr(f), _stream()), "vc_from_dense_padded")
        return f

    # ------------------------------------------------------------------ voxelise + MeanVFE
    def voxelize_mean(self, points: torch.Tensor, pc_range, voxel_size, max_points: int, max_voxels: int,
                      vfe_max_last: bool):
        points = _need(points, torch.float32, "points")
        p, f = points.shape
        dev = points.device
        ws_bytes = self.lib.vc_voxelize_workspace_bytes(p, max_points)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        feats = torch.zeros((max_voxels, f), dtype=torch.float32, device=dev)
        coords = torch.zeros((max_voxels, 3), dtype=torch.int32, device=dev)
        num = torch.zeros((max_voxels,), dtype=torch.int32, device=dev)
        nv = torch.zeros((1,), dtype=torch.int32, device=dev)
        check(self.lib.vc_voxelize_mean(_ptr(points), p, f, f32arr(pc_range), f32arr(voxel_size), max_points, max_voxels,
                                        1 if vfe_max_last else 0, _ptr(ws), ws_bytes, _ptr(feats), _ptr(coords),
                                        _ptr(num), _ptr(nv), _stream()), "vc_voxelize_mean")
        m = self._read_count(nv)
        return feats[:m], coords[:m], num[:m]

    def voxelize(self, points: torch.Tensor, pc_range, voxel_size, max_points: int, max_voxels: int):
        """Un-fused voxeliser with the reference's return protocol (vc_voxelize): zero-padded voxels (M, max_points, F),
        coords (M, 3) [z, y, x], num_points (M,) -- what Point2VoxelCPU3d.point_to_voxel returns (data_processor.py:53-58)."""
        points = _need(points, torch.float32, "points")
        p, f = points.shape
        dev = points.device
        ws_bytes = self.lib.vc_voxelize_workspace_bytes(p, max_points)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        voxels = torch.empty((max_voxels, max_points, f), dtype=torch.float32, device=dev)
        coords = torch.empty((max_voxels, 3), dtype=torch.int32, device=dev)
        num = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
        nv = torch.zeros((1,), dtype=torch.int32, device=dev)
        check(self.lib.vc_voxelize(_ptr(points), p, f, f32arr(pc_range), f32arr(voxel_size), max_points, max_voxels,
                                   _ptr(ws), ws_bytes, _ptr(voxels), _ptr(coords), _ptr(num), _ptr(nv), _stream()),
              "vc_voxelize")
        m = self._read_count(nv)
        return voxels[:m], coords[:m], num[:m]

    # ------------------------------------------------------------------ input point discard + fused front-end
    @staticmethod
    def _perm_table(perms, bin_num: int, device):
        """perms: None | {bin: int64 tensor} | sequence of bin_num entries -> (ctypes host array of device pointers, keep-alive)."""
        if perms is None:
            return None, []
        import ctypes
        arr = (ctypes.c_void_p * bin_num)()
        alive = []
        for b in range(bin_num):
            t = perms.get(b) if isinstance(perms, dict) else perms[b]
            if t is None:
                arr[b] = None
                continue
            t = torch.as_tensor(t).to(device=device, dtype=torch.int64).contiguous()
            alive.append(t)
            arr[b] = t.data_ptr()
        return arr, alive

    def input_discard(self, points: torch.Tensor, bin_num: int, rate: float, max_dis: float = 60.0, perms=None,
                      seed: int = 0, sync: bool = True):
        """StVD input point discard (vc_input_discard; dataset.py:120-189).  points (P, F) float32 or float16.
        -> (out (P, F) float32 capacity buffer, n_out device int32[1]); with sync=True the buffer is sliced to its
        n_out rows (one count read)."""
        if not points.is_cuda:
            raise _lib.VirConvError("input_discard: expected a CUDA/HIP tensor; virconv_amd has no CPU path")
        assert points.dtype in (torch.float32, torch.float16) and points.dim() == 2
        points = points.contiguous()
        p, f = points.shape
        dev = points.device
        ws_bytes = self.lib.vc_input_discard_workspace_bytes(p)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        out = torch.empty((p, f), dtype=torch.float32, device=dev)
        n_out = torch.zeros((1,), dtype=torch.int32, device=dev)
        tbl, alive = self._perm_table(perms, bin_num, dev)
        check(self.lib.vc_input_discard(_ptr(points), 1 if points.dtype == torch.float16 else 0, p, f, int(bin_num),
                                        float(rate), float(max_dis), tbl, seed & 0xFFFFFFFFFFFFFFFF, _ptr(ws), ws_bytes,
                                        _ptr(out), _ptr(n_out), _stream()), "vc_input_discard")
        del alive  # the launches are ordered on the stream; torch's allocator keeps the blocks stream-ordered
        if not sync:
            return out, n_out
        return out[:self._read_count(n_out)], n_out

    def frontend_voxelize_mean(self, lidar: torch.Tensor, virtual: torch.Tensor, bin_num: int, rate: float, pc_range,
                               voxel_size, max_points: int, max_voxels: int, vfe_max_last: bool, max_dis: float = 60.0,
                               perms=None, seed: int = 0, intensity_div: float = 0.0, sync: bool = True):
        """Fused data front-end (vc_frontend_voxelize_mean): raw LiDAR (Pl, F) f32 + raw virtual (Pv, F) f32|f16 ->
        input discard -> LiDAR-first concat -> voxeliser + MeanVFE.  -> (features, coords, num_points[, n_voxels dev])."""
        lidar = _need(lidar, torch.float32, "lidar points")
        if not virtual.is_cuda:
            raise _lib.VirConvError("frontend_voxelize_mean: expected CUDA/HIP tensors; virconv_amd has no CPU path")
        assert virtual.dtype in (torch.float32, torch.float16)
        virtual = virtual.contiguous()
        pl, f = lidar.shape
        pv = virtual.shape[0]
        assert virtual.shape[1] == f
        dev = lidar.device
        ws_bytes = self.lib.vc_frontend_workspace_bytes(pl, pv, f, max_points)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        feats = torch.zeros((max_voxels, f), dtype=torch.float32, device=dev)
        coords = torch.zeros((max_voxels, 3), dtype=torch.int32, device=dev)
        num = torch.zeros((max_voxels,), dtype=torch.int32, device=dev)
        nv = torch.zeros((2,), dtype=torch.int32, device=dev)
        tbl, alive = self._perm_table(perms, bin_num, dev)
        check(self.lib.vc_frontend_voxelize_mean(_ptr(lidar), pl, _ptr(virtual), 1 if virtual.dtype == torch.float16 else 0,
                                                 pv, f, int(bin_num), float(rate), float(max_dis), tbl,
                                                 seed & 0xFFFFFFFFFFFFFFFF, float(intensity_div), f32arr(pc_range),
                                                 f32arr(voxel_size), max_points, max_voxels, 1 if vfe_max_last else 0,
                                                 _ptr(ws), ws_bytes, _ptr(feats), _ptr(coords), _ptr(num), _ptr(nv[:1]),
                                                 _ptr(nv[1:]), _stream()), "vc_frontend_voxelize_mean")
        del alive
        if not sync:
            return feats, coords, num, nv
        m = self._read_count(nv[:1])
        return feats[:m], coords[:m], num[:m]

    # ------------------------------------------------------------------ BatchNorm(+ReLU)
    def bn_forward(self, x: torch.Tensor, gamma, beta, running_mean, running_var, training: bool, momentum: float,
                   eps: float, relu: bool, out: Optional[torch.Tensor] = None, out_col0: int = 0,
                   num_batches_tracked: Optional[torch.Tensor] = None, partial: Optional[torch.Tensor] = None):
        """-> (y, mean, var).  `out` (N, Ctot) lets the result land at a column offset of a wider row (fused concat).
        `partial`: per-block sums from conv_forward_stats -- the statistics then need no pass over x."""
        x = _need(x, torch.float32, "features")
        n, c = x.shape
        dev = x.device
        st = _stream()
        if training:
            mean = torch.empty((c,), dtype=torch.float32, device=dev)
            var = torch.empty((c,), dtype=torch.float32, device=dev)
            if partial is not None:
                ws_bytes = self.lib.vc_bn_workspace_bytes(n, c)
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
                check(self.lib.vc_bn_stats_from_partial(_ptr(partial), partial.numel() // (2 * c), n, c, _ptr(mean),
                                                        _ptr(var), _ptr(running_mean), _ptr(running_var),
                                                        _ptr(num_batches_tracked), float(momentum), _ptr(ws), ws_bytes, st),
                      "vc_bn_stats_from_partial")
            else:
                ws_bytes = self.lib.vc_bn_workspace_bytes(n, c)
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
                check(self.lib.vc_bn_stats(_ptr(x), n, c, _ptr(mean), _ptr(var), _ptr(running_mean), _ptr(running_var),
                                           _ptr(num_batches_tracked), float(momentum), _ptr(ws), ws_bytes, st),
                      "vc_bn_stats")
        else:
            mean, var = running_mean, running_var
        if out is None:
            out = torch.empty((n, c), dtype=torch.float32, device=dev)
            out_col0 = 0
        check(self.lib.vc_bn_apply_relu(_ptr(x), n, c, _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), float(eps),
                                        1 if relu else 0, _ptr(out), out.shape[1], out_col0, st), "vc_bn_apply_relu")
        return out, mean, var

    # ------------------------------------------------------------------ post_act_block (conv + BN + ReLU) as one call each way
    def _side_stream(self, device) -> int:
        st = getattr(self, "_side", None)
        if st is None or st.device != torch.device(device):
            st = torch.cuda.Stream(device=device, priority=int(os.environ.get("VIRCONV_SIDE_PRIORITY", "0")))
            self._side = st
        return st.cuda_stream

    def post_act_block_forward(self, x, weight, tbl, order, operand: str, sorted_rows: bool, gamma, beta, running_mean,
                               running_var, nbt, momentum: float, eps: float, relu: bool):
        """conv -> training-mode BatchNorm1d -> (ReLU) in ONE C-ABI call (vc_post_act_block_forward).
        -> (y, y_raw, mean, var)"""
        x = _need(x, torch.float32, "features")
        weight = _need(weight, torch.float32, "weight")
        kv, n_out = tbl.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        dev = x.device
        flags = CONV_SORTED_ROWS if sorted_rows else 0
        y_raw = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
        y = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
        stats = torch.empty((2, cout), dtype=torch.float32, device=dev)
        ws_bytes = self.lib.vc_post_act_block_forward_workspace_bytes(x.shape[0], n_out, kv, cin, cout, flags)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        check(self.lib.vc_post_act_block_forward(_ptr(x), x.shape[0], _ptr(tbl), n_out, kv, _ptr(weight), cin, cout,
                                                 _ptr(order), OPERAND_TYPES[operand], flags, _ptr(gamma), _ptr(beta),
                                                 _ptr(running_mean), _ptr(running_var), _ptr(nbt), float(momentum),
                                                 float(eps), 1 if relu else 0, _ptr(y_raw), _ptr(y), cout, 0,
                                                 _ptr(stats[0]), _ptr(stats[1]), _ptr(ws), ws_bytes, _stream()),
              "vc_post_act_block_forward")
        return y, y_raw, stats[0], stats[1]

    def weighted_sum(self, x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        """sum_b sum_i x[b, i] * g[i] as ONE deterministic pass over x (vc_weighted_sum); x: (nb, ...) contiguous, g: the shape of
        one sample (or with a leading 1).  -> 0-dim float32 tensor."""
        x = _need(x, torch.float32, "x")
        g = _need(g, torch.float32, "g")
        nb = x.shape[0]
        e = x.numel() // max(nb, 1)
        if g.numel() != e:
            raise ValueError(f"weighted_sum: g has {g.numel()} elements, one sample of x has {e}")
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws_bytes = self.lib.vc_weighted_sum_workspace_bytes(nb, e)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
        check(self.lib.vc_weighted_sum(_ptr(x), nb, e, _ptr(g), _ptr(out), _ptr(ws), ws_bytes, _stream()), "vc_weighted_sum")
        return out

    def weighted_sum_backward(self, gout: torch.Tensor, g: torch.Tensor, nb_out: int) -> torch.Tensor:
        """(nb_out, e) rows gout * g (vc_weighted_sum_backward); nb_out = 1 gives the row every sample shares."""
        gout = _need(gout, torch.float32, "grad_out")
        g = _need(g, torch.float32, "g")
        e = g.numel()
        dx = torch.empty((nb_out, e), dtype=torch.float32, device=g.device)
        check(self.lib.vc_weighted_sum_backward(_ptr(gout), _ptr(g), nb_out, e, _ptr(dx), _stream()), "vc_weighted_sum_backward")
        return dx

    def group_plan(self, rep: torch.Tensor) -> torch.Tensor:
        """(2, n) int32 [rows sorted stably by representative | sorted representatives] of a duplicate-pixel table: what
        vc_group_sum_sorted walks.  Built once per table by the geometry plan (vc_group_plan: keys + one stable radix sort)."""
        rep = _need(rep, torch.int32, "rep")
        n = rep.shape[0]
        plan = torch.empty((2, n), dtype=torch.int32, device=rep.device)
        ws_bytes = self.lib.vc_group_plan_workspace_bytes(n)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=rep.device)
        check(self.lib.vc_group_plan(_ptr(rep), n, _ptr(plan), _ptr(ws), ws_bytes, _stream()), "vc_group_plan")
        return plan

    def group_sum_sorted(self, dy: torch.Tensor, plan: torch.Tensor) -> torch.Tensor:
        """dy_grp[rep] = sum of dy over the rows of rep's pixel group, written at representatives only (vc_group_sum_sorted)."""
        dy = _need(dy, torch.float32, "grad_out")
        n, c = dy.shape
        grp = torch.empty_like(dy)
        ws_bytes = self.lib.vc_group_sum_sorted_workspace_bytes(n, c)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dy.device)
        check(self.lib.vc_group_sum_sorted(_ptr(dy), _ptr(plan), n, c, _ptr(grp), _ptr(ws), ws_bytes, _stream()),
              "vc_group_sum_sorted")
        return grp

    def post_act_block_backward(self, x, weight, y_raw, dy_wide, dy_col0: int, mean, var, gamma, beta, eps: float, relu: bool,
                                pair_fwd, tbl_dx, n_dx: int, mirror: bool, centre: int, rep, grp_plan, order_dx, operand: str,
                                sorted_rows: bool, need_dx: bool, need_dw: bool):
        """BatchNorm+ReLU backward -> (group sum) -> backward-input conv -> weight gradient in ONE C-ABI call.
        -> (dx | None, dw | None, dgamma, dbeta)"""
        kv, n_out = pair_fwd.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        dev = x.device
        flags = CONV_SORTED_ROWS if sorted_rows else 0
        d_raw = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
        dx = torch.empty((n_dx, cin), dtype=torch.float32, device=dev) if need_dx else None
        dw = torch.empty(tuple(weight.shape), dtype=torch.float32, device=dev) if need_dw else None
        dgb = torch.empty((2, cout), dtype=torch.float32, device=dev)
        ws_bytes = self.lib.vc_post_act_block_backward_workspace_bytes(n_out, kv, cin, cout)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        x = _need(x, torch.float32, "features")          # saved tensors may be user views: the C call reads raw pointers
        weight = _need(weight, torch.float32, "weight")
        y_raw = _need(y_raw, torch.float32, "y_raw")
        if rep is not None and need_dx and grp_plan is None:
            grp_plan = self.group_plan(rep)
        check(self.lib.vc_post_act_block_backward(_ptr(x), x.shape[0], _ptr(y_raw), n_out, _ptr(dy_wide), dy_wide.shape[1],
                                                  dy_col0, _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), float(eps),
                                                  1 if relu else 0, _ptr(pair_fwd), _ptr(tbl_dx), n_dx, 1 if mirror else 0,
                                                  centre, _ptr(rep), _ptr(grp_plan), _ptr(order_dx), kv, _ptr(weight), cin, cout,
                                                  OPERAND_TYPES[operand], flags, 1 if need_dx else 0, 1 if need_dw else 0,
                                                  _ptr(d_raw), _ptr(dx), _ptr(dw), _ptr(dgb[0]), _ptr(dgb[1]), _ptr(ws), ws_bytes,
                                                  self._side_stream(dev) if UNIT_OVERLAP_DW else None, _stream()),
              "vc_post_act_block_backward")
        return dx, dw, dgb[0], dgb[1]

    def bn_backward(self, x, dy, dy_col0, mean, var, gamma, beta, eps: float, relu: bool,
                    absmax_ws: Optional[torch.Tensor] = None):
        x = _need(x, torch.float32, "features")
        dy = _need(dy, torch.float32, "grad_out")
        n, c = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        dgamma = torch.empty((c,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
        ws_bytes = self.lib.vc_bn_workspace_bytes(n, c)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        check(self.lib.vc_bn_relu_backward(_ptr(x), _ptr(dy), dy.shape[1], dy_col0, n, c, _ptr(mean), _ptr(var), _ptr(gamma),
                                           _ptr(beta), float(eps), 1 if relu else 0, _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                           _ptr(absmax_ws), _ptr(ws), ws_bytes, _stream()), "vc_bn_relu_backward")
        return dx, dgamma, dbeta
