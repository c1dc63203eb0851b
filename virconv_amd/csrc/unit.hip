// post_act_block as ONE C-ABI call: conv (no bias) -> BatchNorm1d (training mode) -> ReLU, forward and backward
// (pcdet/models/backbones_3d/spconv_backbone.py:86-107 post_act_block, :110-131 post_act_block2d -- every conv of VirConvL8x /
// VirConv8x is wrapped in one).  Pure host-side composition of the operators declared in include/virconv_hip.h: no new
// kernels, the same launches in the same order as calling them one by one, hence bit-identical results.  What it buys is the
// HOST: a train step is ~380 launches, and issuing each one from Python (ctypes marshalling + a scratch allocation + a torch
// allocator round trip per call) costs ~16 us of host time per launch, which bounds the step on a box with a slow host.
// Here a unit's forward is one call (2-4 launches) and its backward one call (5-9 launches) at ~4 us per launch.
#include "common.h"

namespace vc {
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace vc

using namespace vc;

extern "C" {

size_t vc_post_act_block_forward_workspace_bytes(int64_t n_in, int64_t n_out, int kv, int cin, int cout, int flags) {
  if (n_in < 0 || n_out < 0 || cin < 1 || cout < 1 || kv < 1) return 0;
  return al256(vc_bn_workspace_bytes(n_out, cout)) +
         al256(vc_conv_stats_partial_floats(n_in, n_out, cin, cout, kv, flags) * sizeof(float)) + 256;
}

int vc_post_act_block_forward(const float* x, int64_t n_in, const int32_t* pair_fwd, int64_t n_out, int kv,
                              const float* weight, int cin, int cout, const int32_t* row_order, int operand_type, int flags,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float momentum, float eps, int relu, float* y_raw, float* y,
                              int y_stride, int y_col0, float* mean, float* var, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(n_out >= 1 && y_raw && y && mean && var && ws, "vc_post_act_block_forward: null/invalid argument");
  if (ws_bytes < vc_post_act_block_forward_workspace_bytes(n_in, n_out, kv, cin, cout, flags)) {
    set_error("vc_post_act_block_forward: workspace too small");
    return VC_ECAPACITY;
  }
  char* bn_ws = (char*)ws;
  const size_t bn_bytes = vc_bn_workspace_bytes(n_out, cout);
  float* partial = (float*)(bn_ws + al256(bn_bytes));
  int rc;
  // BatchNorm statistics from the conv epilogue (per-wave partial sums, no extra barrier) wherever the epilogue kernels serve
  // the shape; else from a pass over y_raw
  const size_t pf = vc_conv_stats_partial_floats(n_in, n_out, cin, cout, kv, flags);
  if (operand_type == VC_OPERAND_F32 && vc_conv_epilogue_supported(n_in, cin, cout, kv, VC_OPERAND_F32)) {
    // the conv launch finishes the statistics itself where its kernel can (conv_finish_tail: no reduce / finalize launches)
    conv_finish_arm(BnFinishRequest{0, n_out, mean, var, running_mean, running_var, (long long*)num_batches_tracked, momentum,
                                    (double*)bn_ws});
    rc = vc_conv_forward_epilogue(x, n_in, pair_fwd, n_out, kv, weight, cin, cout, row_order, VC_EPI_STATS, flags, partial,
                                  nullptr, nullptr, nullptr, nullptr, 0.f, 0, y_raw, stream);
    const bool finished = conv_finish_take((hipStream_t)stream);
    if (rc != VC_OK) return rc;
    if (!finished)
      rc = vc_bn_stats_from_partial(partial, (int64_t)(pf / (2 * (size_t)cout)), n_out, cout, mean, var, running_mean,
                                    running_var, num_batches_tracked, momentum, bn_ws, bn_bytes, stream);
  } else {
    rc = vc_conv_forward(x, n_in, pair_fwd, n_out, kv, weight, cin, cout, row_order, operand_type, flags, y_raw, stream);
    if (rc != VC_OK) return rc;
    rc = vc_bn_stats(y_raw, n_out, cout, mean, var, running_mean, running_var, num_batches_tracked, momentum, bn_ws, bn_bytes,
                     stream);
  }
  if (rc != VC_OK) return rc;
  return vc_bn_apply_relu(y_raw, n_out, cout, mean, var, gamma, beta, eps, relu, y, y_stride, y_col0, stream);
}

size_t vc_post_act_block_backward_workspace_bytes(int64_t n_out, int kv, int cin, int cout) {
  if (n_out < 0 || cin < 1 || cout < 1 || kv < 1) return 0;
  return al256(vc_bn_workspace_bytes(n_out, cout)) + al256(vc_conv_backward_weight_workspace_bytes(n_out, kv, cin, cout)) +
         al256((size_t)n_out * cout * sizeof(float)) /* group-summed d_raw (duplicate-pixel convs) */ +
         al256(vc_group_sum_sorted_workspace_bytes(n_out, cout)) /* its chunk partials */ + 256;
}

/* tbl_dx: the table the backward-input gather-GEMM walks (SubM: pair_fwd with mirror = 1; strided: pair_bwd, mirror = 0), over
 * n_dx output rows (= the conv's INPUT rows); pair_fwd is always the forward table (n_out columns) for the weight gradient.
 * rep / centre: duplicate-pixel rule of the 2-D SubM convs (NULL / -1 otherwise); grp_plan: the table's rows sorted by
 * representative, [order | sorted keys] (vc_group_sum_sorted; required when rep != NULL and need_dx).                      */
// fork/join events for the optional side-stream weight gradient (created once per host thread; no device memory)
static hipEvent_t* unit_events() {
  static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  if (ev[0] == nullptr) {
    if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
      ev[0] = ev[1] = nullptr;
      return nullptr;
    }
  }
  return ev;
}

int vc_post_act_block_backward(const float* x, int64_t n_in, const float* y_raw, int64_t n_out, const float* dy,
                               int dy_stride, int dy_col0, const float* mean, const float* var, const float* gamma,
                               const float* beta, float eps, int relu, const int32_t* pair_fwd, const int32_t* tbl_dx,
                               int64_t n_dx, int mirror, int centre, const int32_t* rep, const int32_t* grp_plan,
                               const int32_t* row_order_dx, int kv,
                               const float* weight, int cin, int cout, int operand_type, int flags, int need_dx, int need_dw,
                               float* d_raw, float* dx, float* dw, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                               void* side_stream, void* stream) {
  VC_REQUIRE(n_out >= 1 && x && y_raw && dy && mean && var && d_raw && dgamma && dbeta && ws && pair_fwd && weight,
             "vc_post_act_block_backward: null/invalid argument");
  VC_REQUIRE(!need_dx || (tbl_dx && dx), "vc_post_act_block_backward: need_dx without table / output");
  VC_REQUIRE(!need_dw || dw, "vc_post_act_block_backward: need_dw without output");
  if (ws_bytes < vc_post_act_block_backward_workspace_bytes(n_out, kv, cin, cout)) {
    set_error("vc_post_act_block_backward: workspace too small");
    return VC_ECAPACITY;
  }
  char* bn_ws = (char*)ws;
  const size_t bn_bytes = vc_bn_workspace_bytes(n_out, cout);
  char* dw_ws = bn_ws + al256(bn_bytes);
  const size_t dw_bytes = vc_conv_backward_weight_workspace_bytes(n_out, kv, cin, cout);
  float* grp = (float*)(dw_ws + al256(dw_bytes));
  char* gpart = (char*)grp + al256((size_t)n_out * cout * sizeof(float));
  const size_t gpart_bytes = vc_group_sum_sorted_workspace_bytes(n_out, cout);
  const bool dup = rep != nullptr && need_dx;
  if (dup) {
    VC_REQUIRE(grp_plan && (cout & (cout - 1)) == 0,
               "vc_post_act_block_backward: duplicate-pixel conv needs the table's group plan (vc_group_sum_sorted)");
  }
  int rc = vc_bn_relu_backward(y_raw, dy, dy_stride, dy_col0, n_out, cout, mean, var, gamma, beta, eps, relu, d_raw, dgamma,
                               dbeta, nullptr, bn_ws, bn_bytes, stream);
  if (rc != VC_OK) return rc;
  // dW and dX both only read d_raw and are independent: with a side stream the weight gradient runs underneath the
  // backward-input conv (fork after the BatchNorm backward, join before returning -- the caller's stream-ordered allocator may
  // then recycle every buffer of this call).  Both kernels are latency-bound at ~50 % of the matrix pipes on their own.
  hipEvent_t* ev = (side_stream && need_dx && need_dw) ? unit_events() : nullptr;
  bool forked = false;
  if (dup) {  // the group sum feeds the backward-input conv AND the weight gradient (non-centre taps over representatives only)
    rc = vc_group_sum_sorted(d_raw, grp_plan, n_out, cout, grp, gpart, gpart_bytes, stream);
    if (rc != VC_OK) return rc;
  }
  auto weight_grad = [&](void* s_) -> int {
    return dup ? vc_conv_backward_weight_dup(x, d_raw, grp, rep, centre, pair_fwd, n_out, kv, cin, cout, operand_type, dw, dw_ws,
                                             dw_bytes, s_)
               : vc_conv_backward_weight(x, d_raw, pair_fwd, n_out, kv, cin, cout, operand_type, dw, dw_ws, dw_bytes, s_);
  };
  if (ev != nullptr) {
    VC_CHECK_HIP(hipEventRecord(ev[0], (hipStream_t)stream));
    VC_CHECK_HIP(hipStreamWaitEvent((hipStream_t)side_stream, ev[0], 0));
    rc = weight_grad(side_stream);
    if (rc != VC_OK) return rc;
    VC_CHECK_HIP(hipEventRecord(ev[1], (hipStream_t)side_stream));
    forked = true;
  }
  if (need_dx) {
    const float* src = d_raw;
    const float* src_centre = nullptr;
    if (dup) {
      src = grp;
      src_centre = d_raw;
    }
    rc = vc_conv_backward_input(src, src_centre, n_out, tbl_dx, n_dx, kv, weight, cin, cout, mirror, dup ? centre : -1,
                                dup ? rep : nullptr, row_order_dx, operand_type, flags, dx, stream);
    if (rc != VC_OK) return rc;
  }
  if (forked) {
    VC_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, ev[1], 0));
  } else if (need_dw) {
    rc = weight_grad(stream);
    if (rc != VC_OK) return rc;
  }
  (void)n_in;
  return VC_OK;
}

}  // extern "C"
