"""The C-ABI library loads and exports every symbol include/virconv_hip.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from virconv_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "virconv_hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    assert len(syms) >= 24
    for must in ("vc_hash_build", "vc_subm_rulebook", "vc_spconv_mark_count", "vc_spconv_emit_pairs", "vc_conv_forward",
                 "vc_conv_backward_input", "vc_conv_backward_weight", "vc_project_uv", "vc_gather_rows", "vc_to_dense",
                 "vc_voxelize_mean", "vc_bn_stats"):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build(force=False, verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/virconv_hip.h but not exported by {path}"


def test_binding_table_covers_header_exactly():
    assert sorted(_lib.SIGNATURES.keys()) == header_symbols()
    lib = _lib.load()
    assert lib.vc_version().decode().startswith("virconv_hip")


def test_host_side_queries_and_argument_validation_without_gpu():
    lib = _lib.load()
    assert lib.vc_hash_workspace_bytes(1000) == 8 * 1024 * 12  # 8 slots per x-octet, octets = next_pow2(> n): a free slot in every chain
    assert lib.vc_spconv_workspace_bytes(1, 3, _lib.i32arr([41, 800, 704])) > 41 * 800 * 704 // 8
    assert lib.vc_conv_backward_weight_workspace_bytes(1000, 27, 64, 64) >= 16 * 27 * 64 * 64 * 4
    assert lib.vc_voxelize_workspace_bytes(1000, 5) > 0 and lib.vc_bn_workspace_bytes(1000, 64) > 0
    # invalid arguments are rejected with a status code + message, never exit()/abort (include/virconv_hip.h)
    st = lib.vc_hash_build(None, 10, 5, _lib.i32arr([1, 2, 3]), None, 0, None)
    assert st == _lib.VC_EINVAL and b"ndim" in lib.vc_last_error()
    st = lib.vc_conv_forward(None, 0, None, 10, 27, None, 8, 8, None, 0, 0, None, None)
    assert st == _lib.VC_EINVAL
    st = lib.vc_gather_rows(None, None, 7, 4, None, 0, None, None, None)
    assert st == _lib.VC_EINVAL and b"multiple of 4" in lib.vc_last_error()
    with pytest.raises(_lib.VirConvError):
        _lib.check(st, "vc_gather_rows")


def test_new_entry_points_validate_their_arguments_without_gpu():
    """Every check below fails BEFORE any HIP call, so it runs in the build container (no device)."""
    import ctypes
    lib = _lib.load()
    dummy = ctypes.c_void_p(64)  # never dereferenced on the host
    shp = _lib.i32arr([21, 400, 352])
    # sparse BEV stem, transposed table: invalid kernel sizes / shapes are rejected before any launch; an empty tensor is a no-op
    assert lib.vc_bev_pairs_backward(dummy, 10, 2, _lib.i32arr([4, 200, 176]), 2, 3, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_bev_pairs_backward(dummy, 10, 2, _lib.i32arr([40, 200, 176]), 3, 3, dummy, None) == _lib.VC_EINVAL and b"kernel volume" in lib.vc_last_error()
    assert lib.vc_bev_pairs_backward(None, 0, 2, _lib.i32arr([4, 200, 176]), 3, 3, None, None) == _lib.VC_OK
    # row order: window and kernel-volume limits
    assert lib.vc_row_order(dummy, 10, 27, None, -1, 512, dummy, None) == _lib.VC_EINVAL and b"window" in lib.vc_last_error()
    assert lib.vc_row_order(dummy, 10, 33, None, -1, 1024, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_row_order(None, 0, 27, None, -1, 1024, None, None) == _lib.VC_OK          # empty table: nothing to do
    # conv epilogues / operand types
    assert lib.vc_conv_forward(dummy, 4, dummy, 4, 27, dummy, 8, 8, None, 7, 0, dummy, None) == _lib.VC_EINVAL
    assert b"operand_type" in lib.vc_last_error()
    assert lib.vc_conv_forward_epilogue(dummy, 4, dummy, 4, 27, dummy, 8, 8, None, 5, 0, None, None, None, None, None, 0.0, 0,
                                        dummy, None) == _lib.VC_EINVAL
    assert lib.vc_conv_forward_epilogue(dummy, 4, dummy, 4, 27, dummy, 8, 8, None, 1, 0, None, None, None, None, None, 0.0, 0,
                                        dummy, None) == _lib.VC_EINVAL and b"stats_partial" in lib.vc_last_error()
    assert lib.vc_conv_epilogue_supported(1000, 64, 32, 27, 0) == 1 and lib.vc_conv_epilogue_supported(1000, 64, 32, 27, 1) == 0
    assert lib.vc_conv_epilogue_supported(1 << 24, 64, 32, 27, 0) == 0                      # source >= 2 GiB: fallback kernel
    # one partial row (sum, sum of squares) per 16-row wave tile: 4 per 64-row block
    assert lib.vc_conv_stats_partial_floats(130, 130, 8, 32, 27, 0) == 3 * 4 * 2 * 32
    # ... for launches of >= 16-channel shapes under 62 000 output rows the blocks have 8 waves of 16 rows (round 6, conv_nw8_below): 8 per 128-row block
    assert lib.vc_conv_stats_partial_floats(130, 130, 32, 32, 27, 1) == 2 * 8 * 2 * 32
    assert lib.vc_conv_stats_partial_floats(100000, 100000, 32, 32, 27, 1) == ((100000 + 63) // 64) * 4 * 2 * 32
    # RoI grid pooling
    assert lib.vc_voxel_index_workspace_bytes(1000, 2, shp) > 2 * 21 * 400 * 352 // 8
    assert lib.vc_voxel_query(dummy, 1 << 30, 10, 2, shp, dummy, dummy, dummy, 5, 1, 1, 32, 1.0, 4, dummy, dummy, None) == _lib.VC_EINVAL
    assert b"x_range" in lib.vc_last_error()
    assert lib.vc_voxel_query(dummy, 16, 10, 2, shp, dummy, dummy, dummy, 5, 1, 1, 1, 1.0, 4, dummy, dummy, None) == _lib.VC_ECAPACITY
    assert lib.vc_group_points(1, 5, 0, 4, dummy, dummy, dummy, dummy, dummy, None) == _lib.VC_EINVAL
    # layer discard
    assert lib.vc_random_keep(10, 11, 1, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_random_keep(10, 0, 1, None, None) == _lib.VC_OK
    # group sum (segmented, fixed order): channel count must be a power of two
    assert lib.vc_group_sum_sorted_workspace_bytes(100, 16) > 0
    assert lib.vc_group_sum_sorted(dummy, dummy, 100, 24, dummy, dummy, 1 << 20, None) == _lib.VC_EINVAL


def test_product_has_no_cpu_path():
    import torch
    from virconv_amd.backend_hip import HipBackend
    be = HipBackend()
    with pytest.raises(_lib.VirConvError, match="no CPU path"):
        be.conv_forward(torch.zeros(4, 8), torch.zeros(8, 3, 3, 3, 8), torch.zeros((27, 4), dtype=torch.int32))
    with pytest.raises(_lib.VirConvError, match="no CPU path"):
        be.subm_rulebook(torch.zeros((4, 4), dtype=torch.int32), (4, 4, 4), (3, 3, 3), (1, 1, 1), False)


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "virconv_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_front_end_entry_points_validate_their_arguments_without_gpu():
    """vc_voxelize / vc_input_discard / vc_frontend_voxelize_mean: every check below fails before any HIP call."""
    import ctypes
    lib = _lib.load()
    dummy = ctypes.c_void_p(64)
    rng, vs = _lib.f32arr([0, -40, -3, 70.4, 40, 1]), _lib.f32arr([0.05, 0.05, 0.05])
    assert lib.vc_input_discard_workspace_bytes(60000) > 60000 * 4 and lib.vc_input_discard_workspace_bytes(-1) == 0
    assert lib.vc_frontend_workspace_bytes(20000, 60000, 8, 5) > lib.vc_voxelize_workspace_bytes(80000, 5)
    # bin count out of range / bad rate / missing count pointer / odd fp16 feature count / short workspace
    assert lib.vc_input_discard(dummy, 0, 10, 8, 17, 0.8, 60.0, None, 0, dummy, 1 << 20, dummy, dummy, None) == _lib.VC_EINVAL
    assert b"bin_num" in lib.vc_last_error()
    assert lib.vc_input_discard(dummy, 0, 10, 8, 2, 1.0, 60.0, None, 0, dummy, 1 << 20, dummy, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_input_discard(dummy, 0, 10, 8, 2, 0.8, 60.0, None, 0, dummy, 1 << 20, dummy, None, None) == _lib.VC_EINVAL
    assert lib.vc_input_discard(dummy, 1, 10, 7, 2, 0.8, 60.0, None, 0, dummy, 1 << 20, dummy, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_input_discard(dummy, 0, 10, 8, 2, 0.8, 60.0, None, 0, dummy, 16, dummy, dummy, None) == _lib.VC_ECAPACITY
    assert lib.vc_voxelize(None, 10, 8, rng, vs, 5, 100, dummy, 1 << 20, dummy, dummy, dummy, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_voxelize(dummy, 10, 8, rng, vs, 5, 100, dummy, 16, dummy, dummy, dummy, dummy, None) == _lib.VC_ECAPACITY
    assert lib.vc_frontend_voxelize_mean(None, 10, dummy, 0, 10, 8, 2, 0.8, 60.0, None, 0, 0.0, rng, vs, 5, 100, 1, dummy,
                                         1 << 24, dummy, dummy, dummy, dummy, None, None) == _lib.VC_EINVAL
    assert lib.vc_frontend_voxelize_mean(dummy, 10, dummy, 0, 10, 8, 2, 0.8, 60.0, None, 0, 0.0, rng, vs, 5, 100, 1, dummy,
                                         16, dummy, dummy, dummy, dummy, None, None) == _lib.VC_ECAPACITY


def _tiny_pass_program(n=100, cin=8, cout=16, n_out=None, training=1):
    """One post_act_block on fake (never dereferenced) device pointers: enough for the host-side layout / validation code."""
    import ctypes as C
    n_out = n if n_out is None else n_out
    fake = 0x10000
    ops = (_lib.PassOp * 1)()
    ops[0].kind, ops[0].src, ops[0].dst, ops[0].dst_col0, ops[0].unit, ops[0].table, ops[0].relu = _lib.PASS_UNIT, 0, 1, 0, 0, 0, 1
    bufs = (_lib.PassBuf * 2)()
    bufs[0].rows, bufs[0].cols, bufs[0].external, bufs[0].ptr = n, cin, 1, fake
    bufs[1].rows, bufs[1].cols = n_out, cout
    units = (_lib.PassUnit * 1)()
    u = units[0]
    u.weight = u.gamma = u.beta = u.running_mean = u.running_var = fake
    u.cin, u.cout, u.momentum, u.eps = cin, cout, 0.01, 1e-3
    tables = (_lib.PassTable * 1)()
    t = tables[0]
    t.pair_fwd, t.n_in, t.n_out, t.kv, t.subm, t.centre = fake, n, n, 27, 1, 13
    prog = _lib.PassProgram()
    prog.ops, prog.n_ops, prog.bufs, prog.n_bufs = ops, 1, bufs, 2
    prog.units, prog.n_units, prog.tables, prog.n_tables = units, 1, tables, 1
    prog.keeps, prog.n_keeps, prog.training, prog.operand_type = None, 0, training, 0
    return prog, (ops, bufs, units, tables)


def test_feature_pass_layout_and_validation_without_gpu():
    """vc_pass_*: the arena layout and the program checks are host code -- exercised here on fake pointers (no launch)."""
    import ctypes as C
    lib = _lib.load()
    prog, keep = _tiny_pass_program()
    fwd = lib.vc_pass_forward_arena_bytes(C.byref(prog))
    # output buffer + y_raw (kept for the backward) + statistics + the unit's scratch
    assert fwd >= 2 * 100 * 16 * 4 + 2 * 16 * 4 + lib.vc_post_act_block_forward_workspace_bytes(100, 100, 27, 8, 16, 0)
    ext = (C.c_void_p * 2)()
    assert lib.vc_pass_backward_arena_bytes(C.byref(prog), ext, 0) > 0          # no gradient arrives: scratch only
    ext[1] = 0x20000
    b1 = lib.vc_pass_backward_arena_bytes(C.byref(prog), ext, 0)
    b2 = lib.vc_pass_backward_arena_bytes(C.byref(prog), ext, 1)
    assert b1 >= 100 * 16 * 4 + lib.vc_conv_backward_weight_workspace_bytes(100, 27, 8, 16)   # d_raw + weight-gradient partials
    assert b2 >= b1 + 100 * 8 * 4                                                  # + dx of the first unit
    # running statistics: smaller forward arena (no y_raw kept), no backward
    prog_e, keep_e = _tiny_pass_program(training=0)
    assert 0 < lib.vc_pass_forward_arena_bytes(C.byref(prog_e)) < fwd
    assert lib.vc_pass_backward_arena_bytes(C.byref(prog_e), ext, 0) == 0
    assert lib.vc_pass_backward(C.byref(prog_e), 0x1000, 1 << 30, ext, None, None, 0, 0x1000, 1 << 30, None, None) == _lib.VC_EINVAL
    assert b"running statistics" in lib.vc_last_error()
    # shape disagreements are caught before anything is launched
    bad, keep_b = _tiny_pass_program(n_out=90)
    assert lib.vc_pass_forward_arena_bytes(C.byref(bad)) == 0 and b"shapes disagree" in lib.vc_last_error()
    assert lib.vc_pass_forward(C.byref(bad), 0x1000, 1 << 30, None, None) == _lib.VC_EINVAL
    bad, keep_b = _tiny_pass_program()
    keep_b[0][0].dst_col0 = 2
    assert lib.vc_pass_forward_arena_bytes(C.byref(bad)) == 0 and b"multiple of 4" in lib.vc_last_error()
    bad, keep_b = _tiny_pass_program()
    keep_b[0][0].kind = 9
    assert lib.vc_pass_forward(C.byref(bad), 0x1000, 1 << 30, None, None) == _lib.VC_EINVAL and b"unknown kind" in lib.vc_last_error()
    assert lib.vc_pass_forward(None, None, 0, None, None) == _lib.VC_EINVAL
    # arena too small
    assert lib.vc_pass_forward(C.byref(prog), 0x1000, 16, None, None) == _lib.VC_ECAPACITY
    # kernel timing facility: argument checks; ending a trace that never began is a no-op
    n = C.c_int(-1)
    assert lib.vc_trace_end(None, 0, C.byref(n)) == _lib.VC_OK and n.value == 0
    assert lib.vc_trace_begin(4, 64, 32, 16, 0x1000) == _lib.VC_EINVAL   # directions: 0 fwd, 1 bwd-input, 2 weight gradient, 3 BatchNorm backward dx, -1 all
    assert lib.vc_trace_begin(0, 64, 32, 16, None) == _lib.VC_EINVAL


def test_second_session_entry_points_validate_their_arguments_without_gpu():
    """Weight images, the chained strided rulebook, rotated IoU / NMS: every check below fails (or returns) before any HIP call."""
    import ctypes
    lib = _lib.load()
    dummy = ctypes.c_void_p(64)
    # fragment-ordered weight images: only shapes whose two channel counts are multiples of 16, kv <= 32
    assert lib.vc_conv_packed_weight_floats(64, 32, 27, 0) == 27 * 64 * 32 and lib.vc_conv_packed_weight_floats(64, 32, 27, 1) == 27 * 64 * 32
    assert lib.vc_conv_packed_weight_floats(8, 32, 27, 0) == 0 and lib.vc_conv_packed_weight_floats(32, 16, 33, 0) == 0
    assert lib.vc_conv_pack_weights(49, dummy, dummy, dummy, dummy, 0, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_conv_pack_weights(0, None, None, None, None, 0, None, None) == _lib.VC_OK
    w = (ctypes.c_void_p * 1)(64)
    out = (ctypes.c_void_p * 1)(128)
    assert lib.vc_conv_pack_weights(1, w, _lib.i32arr([8]), _lib.i32arr([8]), _lib.i32arr([27]), 0, out, None) == _lib.VC_EINVAL
    assert b"takes no packed image" in lib.vc_last_error()
    assert lib.vc_conv_clear_packed_weights() == _lib.VC_OK
    # partial rows of the backward-input epilogue: per 16-row tile; a row-ordered <32, 64> launch keeps the 8-wave blocks
    # (round 6: launches under 62 000 rows take 8-wave blocks for every >= 16-channel shape -- 2 blocks x 8 waves for 130 rows; above: 4 per 64 rows)
    assert lib.vc_conv_bwd_stats_partial_floats(130, 32, 32, 0) == 2 * 8 * 2 * 32
    assert lib.vc_conv_bwd_stats_partial_floats(100000, 32, 32, 0) == ((100000 + 63) // 64) * 4 * 2 * 32
    assert lib.vc_conv_bwd_stats_partial_floats(130, 64, 32, 1) == 2 * 8 * 2 * 64
    # chained strided rulebook
    shp, k3 = _lib.i32arr([21, 400, 352]), _lib.i32arr([3, 3, 3])
    assert lib.vc_spconv_mark_count_dev(dummy, 10, None, 3, 1, shp, k3, k3, k3, k3, dummy, 1 << 30, dummy, None) == _lib.VC_EINVAL
    assert b"device row count" in lib.vc_last_error()
    assert lib.vc_spconv_emit_indices(4, 1, shp, dummy, 1 << 30, 10, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_spconv_emit_indices(3, 1, shp, dummy, 16, 10, dummy, None) == _lib.VC_ECAPACITY
    assert lib.vc_spconv_pairs(dummy, 10, 3, 1, shp, k3, k3, k3, k3, dummy, 16, 5, dummy, dummy, None) == _lib.VC_ECAPACITY
    assert lib.vc_spconv_pairs(dummy, 10, 3, 1, shp, k3, k3, k3, k3, dummy, 1 << 30, 5, None, dummy, None) == _lib.VC_EINVAL
    # rotated IoU / NMS
    assert lib.vc_boxes_iou_bev(None, 0, None, 5, None, None) == _lib.VC_OK            # an empty side: nothing to do
    assert lib.vc_boxes_iou3d(None, 3, dummy, 5, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_boxes_overlap_bev(dummy, -1, dummy, 5, dummy, None) == _lib.VC_EINVAL
    assert lib.vc_nms_workspace_bytes(4096) == 4096 * 64 * 8 + 256 and lib.vc_nms_workspace_bytes(65) == 65 * 2 * 8 + 256
    assert lib.vc_nms(dummy, 70000, 0.5, 1, dummy, dummy, dummy, 1 << 40, None) == _lib.VC_EINVAL and b"up to" in lib.vc_last_error()
    assert lib.vc_nms(dummy, 100, 0.5, 1, dummy, dummy, dummy, 8, None) == _lib.VC_EINVAL and b"workspace" in lib.vc_last_error()
    assert lib.vc_nms(dummy, 100, 0.5, 1, dummy, None, dummy, 1 << 20, None) == _lib.VC_EINVAL


def test_plan_finish_refuses_an_unfenced_image_space_branch_without_gpu():
    """LOG.md A.17 / VERDICT r5 #1d: a plan with a pixel projection and no `tables_wait_event` is refused unless the caller sets
    `allow_unfenced_projection` -- leaving the field zero must not give the loop that projects voxels onto wrong pixels.  The check sits in
    front of every launch (and of the state check), so it can be exercised on fake pointers."""
    import ctypes as C
    lib = _lib.load()
    d = _lib.PlanDesc()
    d.indices, d.n, d.batch_size = 0x1000, 100, 1
    d.spatial_shape[0], d.spatial_shape[1], d.spatial_shape[2] = 41, 1600, 1408
    d.calib, d.image_shape[0], d.image_shape[1] = 0x2000, 1600, 600
    d.n_blocks = 1
    B = d.blocks[0]
    for a in range(3):
        B.subm_ksize[a], B.subm_dilation[a] = 3, 1
    B.has_2d, B.uv_stride = 1, 1
    for a in range(2):
        B.ksize2d[a], B.dilation2d[a] = 3, 1
    state, out = _lib.PlanState(), _lib.PlanOut()
    args = (C.byref(state), 0x3000, 0x4000, 1 << 20, C.byref(out), None)
    assert lib.vc_plan_finish(C.byref(d), *args) == _lib.VC_EINVAL
    assert b"tables_wait_event is NULL" in lib.vc_last_error() and b"A.17" in lib.vc_last_error()
    # with the event named, or the explicit opt-out, the call gets as far as the state check (this state was never begun)
    d.tables_wait_event = 0x5000
    assert lib.vc_plan_finish(C.byref(d), *args) == _lib.VC_EINVAL and b"vc_plan_begin" in lib.vc_last_error()
    d.tables_wait_event, d.allow_unfenced_projection = None, 1
    assert lib.vc_plan_finish(C.byref(d), *args) == _lib.VC_EINVAL and b"vc_plan_begin" in lib.vc_last_error()
    # a chain without an image-space branch (VirConv8x LiDAR stream) needs neither
    d.allow_unfenced_projection, B.has_2d = 0, 0
    assert lib.vc_plan_finish(C.byref(d), *args) == _lib.VC_EINVAL and b"vc_plan_begin" in lib.vc_last_error()
    # the header says so where a C caller reads it
    hdr = open(HEADER).read()
    assert "allow_unfenced_projection" in hdr and "CAUTION (LOG.md A.17" in hdr
