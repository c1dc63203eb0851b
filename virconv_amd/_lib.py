"""ctypes binding of libvirconv_hip.so (the C ABI declared in include/virconv_hip.h).

This is the binder a maintainer of the reference would place under ``pcdet/ops/virconv/``:
raw device pointers + sizes + the current HIP stream, mirroring the reference's own native-op convention
(pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py:31-37 -> src/voxel_query.cpp:25-41).

There is NO CPU fallback: if the shared object cannot be loaded the import of any op raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 7     # = VC_ABI_VERSION of include/virconv_hip.h (struct layouts this file mirrors)
LIB_PATH = os.environ.get("VIRCONV_LIB", os.path.join(_HERE, "libvirconv_hip.so"))  # override: A/B builds only

OPERAND_TYPES = {"f32": 0, "f16": 1, "bf16": 2}  # vc_operand (include/virconv_hip.h)
CONV_SORTED_ROWS = 1                              # vc_conv_flags
VC_OK, VC_EINVAL, VC_ECAPACITY, VC_EHIP = 0, -1, -2, -3

_P, _I64, _I, _SZ, _F, _D = C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_float, C.c_double

# name -> (restype, argtypes); must list every symbol of include/virconv_hip.h (checked by tests/test_abi.py)
SIGNATURES = {
    "vc_version": (C.c_char_p, []),
    "vc_abi_version": (_I, []),
    "vc_last_error": (C.c_char_p, []),
    "vc_debug_set": (_I, [C.c_char_p, _I]),
    "vc_debug_get": (_I, [C.c_char_p, _P]),
    "vc_debug_stop_event_dependency": (_I, [_P, _P, _I64, _I, _I, _P]),
    "vc_weighted_sum_workspace_bytes": (_SZ, [_I64, _I64]),
    "vc_weighted_sum": (_I, [_P, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "vc_weighted_sum_backward": (_I, [_P, _P, _I64, _I64, _P, _P]),
    "vc_clip_adamw_workspace_bytes": (_SZ, [_I]),
    "vc_clip_adamw": (_I, [_P, _I, _F, _F, _F, _F, _F, _I64, _F, _I, _P, _P, _SZ, _P]),
    "vc_hash_workspace_bytes": (_SZ, [_I64]),
    "vc_hash_build": (_I, [_P, _I64, _I, _P, _P, _SZ, _P]),
    "vc_subm_rulebook": (_I, [_P, _I64, _I, _P, _P, _P, _P, _SZ, _P, _P, _P]),
    "vc_spconv_workspace_bytes": (_SZ, [_I, _I, _P]),
    "vc_spconv_mark_count": (_I, [_P, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "vc_spconv_emit_pairs": (_I, [_P, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _I64, _P, _P, _P, _P]),
    "vc_spconv_mark_count_dev": (_I, [_P, _I64, _P, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "vc_spconv_emit_indices": (_I, [_I, _I, _P, _P, _SZ, _I64, _P, _P]),
    "vc_spconv_pairs": (_I, [_P, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _I64, _P, _P, _P]),
    "vc_conv_forward": (_I, [_P, _I64, _P, _I64, _I, _P, _I, _I, _P, _I, _I, _P, _P]),
    "vc_conv_backward_input": (_I, [_P, _P, _I64, _P, _I64, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "vc_voxel_index_workspace_bytes": (_SZ, [_I64, _I, _P]),
    "vc_voxel_index_build": (_I, [_P, _I64, _I, _P, _P, _SZ, _P]),
    "vc_voxel_query": (_I, [_P, _SZ, _I64, _I, _P, _P, _P, _P, _I64, _I, _I, _I, _F, _I, _P, _P, _P]),
    "vc_group_points": (_I, [_I, _I64, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vc_group_points_grad": (_I, [_I, _I64, _I, _I64, _I, _P, _P, _P, _P, _P, _P]),
    "vc_boxes_overlap_bev": (_I, [_P, _I64, _P, _I64, _P, _P]),
    "vc_boxes_iou_bev": (_I, [_P, _I64, _P, _I64, _P, _P]),
    "vc_boxes_iou3d": (_I, [_P, _I64, _P, _I64, _P, _P]),
    "vc_nms_workspace_bytes": (_SZ, [_I64]),
    "vc_nms": (_I, [_P, _I64, _F, _I, _P, _P, _P, _SZ, _P]),
    "vc_conv_packed_weight_floats": (_SZ, [_I, _I, _I, _I]),
    "vc_conv_pack_weights": (_I, [_I, _P, _P, _P, _P, _I, _P, _P]),
    "vc_conv_clear_packed_weights": (_I, []),
    "vc_conv_epilogue_supported": (_I, [_I64, _I, _I, _I, _I]),
    "vc_conv_stats_partial_floats": (_SZ, [_I64, _I64, _I, _I, _I, _I]),
    "vc_conv_forward_epilogue": (_I, [_P, _I64, _P, _I64, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _F, _I, _P, _P]),
    "vc_bn_stats_from_partial": (_I, [_P, _I64, _I64, _I, _P, _P, _P, _P, _P, _F, _P, _SZ, _P]),
    "vc_random_keep": (_I, [_I64, _I64, C.c_uint64, _P, _P]),
    "vc_row_order": (_I, [_P, _I64, _I, _P, _I, _I, _P, _P]),
    "vc_rep_order_workspace_bytes": (_SZ, [_I64]),
    "vc_rep_order": (_I, [_P, _I64, _P, _P, _SZ, _P]),
    "vc_conv_backward_weight_workspace_bytes": (_SZ, [_I64, _I, _I, _I]),
    "vc_conv_backward_weight": (_I, [_P, _P, _P, _I64, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "vc_conv_backward_weight_dup": (_I, [_P, _P, _P, _P, _I, _P, _I64, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "vc_group_keys": (_I, [_P, _I64, _P, _P]),
    "vc_group_plan_workspace_bytes": (_SZ, [_I64]),
    "vc_group_plan": (_I, [_P, _I64, _P, _P, _SZ, _P]),
    "vc_group_sum_sorted_workspace_bytes": (_SZ, [_I64, _I]),
    "vc_group_sum_sorted": (_I, [_P, _P, _I64, _I, _P, _P, _SZ, _P]),
    "vc_project_prepare": (_I, [_P, _P, _I, _P, _P]),
    "vc_project_uv": (_I, [_P, _I64, _P, _I, _I, _P, _P, _P]),
    "vc_gather_rows": (_I, [_P, _P, _I, _I, _P, _I64, _P, _P, _P]),
    "vc_scatter_rows": (_I, [_P, _I, _P, _I64, _I64, _P, _P]),
    "vc_to_dense": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    "vc_from_dense": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    "vc_to_dense_fill_workspace_bytes": (_SZ, [_I, _I, _P]),
    "vc_to_dense_fill": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "vc_to_dense_fill_padded": (_I, [_P, _P, _I64, _I, _I, _I, _P, _I, _I, _P, _P, _SZ, _P]),
    "vc_from_dense_padded": (_I, [_P, _P, _I64, _I, _I, _I, _P, _I, _I, _P, _P]),
    "vc_voxelize_workspace_bytes": (_SZ, [_I64, _I]),
    "vc_voxelize_mean": (_I, [_P, _I64, _I, _P, _P, _I, _I, _I, _P, _SZ, _P, _P, _P, _P, _P]),
    "vc_voxelize": (_I, [_P, _I64, _I, _P, _P, _I, _I, _P, _SZ, _P, _P, _P, _P, _P]),
    "vc_input_discard_workspace_bytes": (_SZ, [_I64]),
    "vc_input_discard": (_I, [_P, _I, _I64, _I, _I, _D, _D, _P, C.c_uint64, _P, _SZ, _P, _P, _P]),
    "vc_frontend_workspace_bytes": (_SZ, [_I64, _I64, _I, _I]),
    "vc_frontend_voxelize_mean": (_I, [_P, _I64, _P, _I, _I64, _I, _I, _D, _D, _P, C.c_uint64, _F, _P, _P, _I, _I, _I, _P, _SZ,
                                       _P, _P, _P, _P, _P, _P]),
    "vc_post_act_block_forward_workspace_bytes": (_SZ, [_I64, _I64, _I, _I, _I, _I]),
    "vc_post_act_block_forward": (_I, [_P, _I64, _P, _I64, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P,
                                       _I, _I, _P, _P, _P, _SZ, _P]),
    "vc_post_act_block_backward_workspace_bytes": (_SZ, [_I64, _I, _I, _I]),
    "vc_post_act_block_backward": (_I, [_P, _I64, _P, _I64, _P, _I, _I, _P, _P, _P, _P, _F, _I, _P, _P, _I64, _I, _I, _P, _P, _P,
                                        _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "vc_pass_forward_arena_bytes": (_SZ, [_P]),
    "vc_pass_forward": (_I, [_P, _P, _SZ, _P, _P]),
    "vc_pass_backward_arena_bytes": (_SZ, [_P, _P, _I]),
    "vc_pass_backward": (_I, [_P, _P, _SZ, _P, _P, _P, _SZ, _P, _P]),
    "vc_trace_begin": (_I, [_I, _I, _I, _I, _P]),
    "vc_trace_end": (_I, [_P, _I, _P]),
    "vc_conv_bwd_stats_partial_floats": (_SZ, [_I64, _I, _I, _I]),
    "vc_conv_backward_input_epilogue": (_I, [_P, _P, _I64, _P, _I64, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P,
                                             _P, _P, _F, _I, _P, _P, _P]),
    "vc_bn_relu_backward_from_partial": (_I, [_P, _P, _I, _I, _I64, _I, _P, _P, _P, _P, _F, _I, _P, _I64, _P, _P, _P, _P, _P,
                                              _SZ, _P]),
    "vc_bn_workspace_bytes": (_SZ, [_I64, _I]),
    "vc_bn_stats": (_I, [_P, _I64, _I, _P, _P, _P, _P, _P, _F, _P, _SZ, _P]),
    "vc_bn_apply_relu": (_I, [_P, _I64, _I, _P, _P, _P, _P, _F, _I, _P, _I, _I, _P]),
    "vc_bn_relu_backward": (_I, [_P, _P, _I, _I, _I64, _I, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "vc_bev_pairs_workspace_bytes": (_SZ, [_I, _P]),
    "vc_bev_pairs": (_I, [_P, _I64, _I, _P, _I, _I, _P, _P, _SZ, _P]),
    "vc_bev_pairs_backward": (_I, [_P, _I64, _I, _P, _I, _I, _P, _P]),
    "vc_nhwc_to_nchw": (_I, [_P, _I, _I64, _I, _P, _P, _I, _P, _P]),
    "vc_plan_begin_arena_bytes": (_SZ, [_P]),
    "vc_plan_begin": (_I, [_P, _P, _SZ, _P, _P, _P]),
    "vc_plan_wait": (_I, [_P, _P]),
    "vc_plan_finish_arena_bytes": (_SZ, [_P, _P]),
    "vc_plan_finish": (_I, [_P, _P, _P, _P, _SZ, _P, _P]),
    "vc_plan_finish_backward": (_I, [_P, _P, _P, _P, _SZ, _P]),
}



# ---- structs of the feature pass / kernel timing (include/virconv_hip.h: vc_pass_*, vc_trace_record)
PASS_UNIT, PASS_COPY, PASS_GATHER = 1, 2, 3


class PassUnit(C.Structure):
    _fields_ = [("weight", _P), ("gamma", _P), ("beta", _P), ("running_mean", _P), ("running_var", _P),
                ("num_batches_tracked", _P), ("dweight", _P), ("dgamma", _P), ("dbeta", _P),
                ("cin", C.c_int32), ("cout", C.c_int32), ("momentum", _F), ("eps", _F)]


class PassTable(C.Structure):
    _fields_ = [("pair_fwd", _P), ("pair_bwd", _P), ("rep", _P), ("order_fwd", _P), ("order_bwd", _P),
                ("n_in", _I64), ("n_out", _I64), ("kv", C.c_int32), ("subm", C.c_int32), ("centre", C.c_int32),
                ("sorted_rows", C.c_int32), ("grp_plan", _P)]


class PassBuf(C.Structure):
    _fields_ = [("rows", _I64), ("cols", C.c_int32), ("external", C.c_int32), ("ptr", _P)]


class PassOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32), ("dst_col0", C.c_int32), ("unit", C.c_int32),
                ("table", C.c_int32), ("keep", C.c_int32), ("relu", C.c_int32)]


class PassProgram(C.Structure):
    _fields_ = [("ops", C.POINTER(PassOp)), ("n_ops", C.c_int32), ("bufs", C.POINTER(PassBuf)), ("n_bufs", C.c_int32),
                ("units", C.POINTER(PassUnit)), ("n_units", C.c_int32), ("tables", C.POINTER(PassTable)),
                ("n_tables", C.c_int32), ("keeps", C.POINTER(_P)), ("n_keeps", C.c_int32), ("training", C.c_int32),
                ("operand_type", C.c_int32), ("pack_all", C.c_int32), ("reserved_", C.c_int32)]


class TraceRecord(C.Structure):
    _fields_ = [("ms", _F), ("kv", C.c_int32), ("ck", C.c_int32), ("cn", C.c_int32), ("windowed", C.c_int32),
                ("n_src", _I64), ("n_out", _I64), ("pairs", _I64), ("direction", C.c_int32), ("t0_ms", _F)]


# ---- structs of the geometry plan (include/virconv_hip.h: vc_plan_*)
PLAN_MAX_BLOCKS = 8
_I32 = C.c_int32


class PlanConv(C.Structure):
    _fields_ = [("ksize", _I32 * 3), ("stride", _I32 * 3), ("padding", _I32 * 3), ("dilation", _I32 * 3)]


class PlanBlock(C.Structure):
    _fields_ = [("has_down", _I32), ("down", PlanConv), ("subm_ksize", _I32 * 3), ("subm_dilation", _I32 * 3), ("has_2d", _I32),
                ("uv_stride", _I32), ("ksize2d", _I32 * 2), ("dilation2d", _I32 * 2), ("discard", _I32), ("keep_seed", C.c_uint64),
                ("keep", _P), ("keep_rows", _I64)]


class PlanDesc(C.Structure):
    _fields_ = [("indices", _P), ("n", _I64), ("batch_size", _I32), ("spatial_shape", _I32 * 3), ("calib", _P), ("trans", _P),
                ("image_shape", _I32 * 2), ("input_discard", _I32), ("input_keep_seed", C.c_uint64), ("input_keep", _P),
                ("input_keep_rows", _I64), ("n_blocks", _I32), ("blocks", PlanBlock * PLAN_MAX_BLOCKS), ("has_tail", _I32),
                ("tail", PlanConv), ("discard_rate", _D), ("need_grad", _I32), ("row_order_fwd", _I32), ("defer_early_tables", _I32),
                ("allow_unfenced_projection", _I32), ("tables_wait_event", _P),
                ("debug_buf", _P), ("debug_bytes", _I64)]


class PlanView(C.Structure):
    _fields_ = [("arena", _I32), ("cols", _I32), ("offset", _I64), ("rows", _I64)]


class PlanTableOut(C.Structure):
    _fields_ = [("pair_fwd", PlanView), ("pair_bwd", PlanView), ("rep", PlanView), ("order_fwd", PlanView), ("order_bwd", PlanView),
                ("grp_plan", PlanView), ("in_indices", PlanView), ("out_indices", PlanView), ("n_in", _I64), ("n_out", _I64),
                ("kv", _I32), ("present", _I32), ("out_shape", _I32 * 3), ("pad_", _I32)]


class PlanBlockOut(C.Structure):
    _fields_ = [("down", PlanTableOut), ("subm3d", PlanTableOut), ("subm2d", PlanTableOut), ("uv", PlanView), ("keep", PlanView),
                ("kept_indices", PlanView), ("n", _I64), ("n_keep", _I64)]


class PlanOut(C.Structure):
    _fields_ = [("input_keep", PlanView), ("input_kept_indices", PlanView), ("n_input_kept", _I64),
                ("blocks", PlanBlockOut * PLAN_MAX_BLOCKS), ("tail", PlanTableOut)]


class PlanState(C.Structure):
    _fields_ = [("opaque", _I64 * 2048)]


_lib = None


ADAM_MAX_TENSORS = 16   # VC_ADAM_MAX_TENSORS


class AdamTensor(C.Structure):   # vc_adam_tensor
    _fields_ = [("param", _P), ("grad", _P), ("exp_avg", _P), ("exp_avg_sq", _P), ("n", _I64)]


class VirConvError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the shared object; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VirConvError(
            f"{LIB_PATH} is missing: build it with `python -m virconv_amd.build` (hipcc, gfx950). "
            "virconv_amd has no CPU or PyTorch fallback for its operators.")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be loaded FIRST so that this library binds to
    # the same runtime instance (same device context, same streams) instead of /opt/rocm's copy: loading ours first
    # gives two runtimes in one process and "no ROCm-capable device is detected" from the second one.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.vc_abi_version() != ABI_VERSION:
        raise VirConvError(f"{LIB_PATH} has struct layout version {lib.vc_abi_version()}, this binding was written against {ABI_VERSION}: "
                           "rebuild the library (python -m virconv_amd.build --force)")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != VC_OK:
        msg = load().vc_last_error().decode("utf-8", "replace")
        raise VirConvError(f"{what} failed with vc_status {status}: {msg}")


def i32arr(vals):
    vals = [int(v) for v in vals]
    return (C.c_int32 * len(vals))(*vals)


def f32arr(vals):
    vals = [float(v) for v in vals]
    return (C.c_float * len(vals))(*vals)
