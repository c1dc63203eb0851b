// Feature pass executor: the conv/BN/ReLU/concat/discard chain of a whole backbone forward in ONE C-ABI call, and its
// reverse-mode sweep in ONE call (include/virconv_hip.h, "feature pass").  Reference path: VirConvL8x.forward
// (pcdet/models/backbones_3d/spconv_backbone.py:609-699) -> NRConvBlock.forward (:207-229) -> post_act_block units (:86-131) and
// layer_voxel_discard (:134-147); its backward is what torch's autograd engine does over spconv's Functions
// (tools/train_utils/train_utils.py:47).  No new arithmetic: every launch goes through the operator entry points of this
// library, in the same order as the node-by-node path, so results are bit-identical.  What this file owns is the HOST side of
// a step: buffer layout in one arena per direction, the gradient bookkeeping of the reverse sweep, and the side-stream
// schedule of the weight gradients.
#include <algorithm>
#include <vector>

#include "common.h"

namespace vc {

static inline size_t al256p(size_t v) { return (v + 255) & ~(size_t)255; }

// dst[:, col0 : col0 + 4*c4] = src   (src dense (n, 4*c4); dst row stride ds floats)
__global__ void __launch_bounds__(256) copy_cols_kernel(const float4* __restrict__ src, int64_t n, int c4, float* __restrict__ dst,
                                                        int ds, int col0) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * c4) return;
  const int64_t r = (e < (1LL << 31)) ? (int64_t)((uint32_t)e / (uint32_t)c4) : e / c4;
  const int j = (int)(e - r * c4);
  *reinterpret_cast<float4*>(dst + r * ds + col0 + 4 * j) = src[e];
}

// out (dense n x 4*c4) = a[:, ac : ac + c] (+ b[:, bc : bc + c])    -- the sum of two gradient contributions of a buffer
__global__ void __launch_bounds__(256) add_views_kernel(float4* out, int64_t n, int c4, const float* a, int as, int ac,
                                                        const float* b, int bs, int bc) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * c4) return;
  const int64_t r = (e < (1LL << 31)) ? (int64_t)((uint32_t)e / (uint32_t)c4) : e / c4;
  const int j = (int)(e - r * c4);
  float4 v = *reinterpret_cast<const float4*>(a + r * as + ac + 4 * j);
  if (b != nullptr) {
    const float4 w = *reinterpret_cast<const float4*>(b + r * bs + bc + 4 * j);
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  out[e] = v;
}

struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += al256p(bytes);
    return o;
  }
};

static int check_program(const vc_pass_program* p) {
  VC_REQUIRE(p && p->ops && p->bufs && p->units && p->tables && p->n_ops >= 1 && p->n_bufs >= 1, "vc_pass: null program field");
  VC_REQUIRE(p->operand_type >= VC_OPERAND_F32 && p->operand_type <= VC_OPERAND_BF16, "vc_pass: unknown operand_type");
  for (int b = 0; b < p->n_bufs; ++b) {
    const vc_pass_buf& B = p->bufs[b];
    VC_REQUIRE(B.rows >= 1 && B.cols >= 4 && B.cols % 4 == 0, "vc_pass: buffer %d has invalid shape (%lld x %d)", b, (long long)B.rows, B.cols);
    VC_REQUIRE(!B.external || B.ptr, "vc_pass: external buffer %d without memory", b);
  }
  for (int i = 0; i < p->n_ops; ++i) {
    const vc_pass_op& o = p->ops[i];
    VC_REQUIRE(o.src >= 0 && o.src < p->n_bufs && o.dst >= 0 && o.dst < p->n_bufs && o.src != o.dst, "vc_pass: op %d: bad buffer id", i);
    VC_REQUIRE(!p->bufs[o.dst].external, "vc_pass: op %d writes an external buffer", i);
    const vc_pass_buf &S = p->bufs[o.src], &D = p->bufs[o.dst];
    VC_REQUIRE(o.dst_col0 >= 0 && o.dst_col0 % 4 == 0, "vc_pass: op %d: dst_col0 must be a non-negative multiple of 4", i);
    if (o.kind == VC_PASS_UNIT) {
      VC_REQUIRE(o.unit >= 0 && o.unit < p->n_units && o.table >= 0 && o.table < p->n_tables, "vc_pass: op %d: bad unit/table id", i);
      const vc_pass_unit& u = p->units[o.unit];
      const vc_pass_table& t = p->tables[o.table];
      VC_REQUIRE(u.weight && u.gamma && u.beta && u.running_mean && u.running_var, "vc_pass: unit %d: null parameter", o.unit);
      VC_REQUIRE(t.pair_fwd && t.kv >= 1 && (t.subm || t.pair_bwd), "vc_pass: table %d: null pair table", o.table);
      VC_REQUIRE(S.cols == u.cin && S.rows == t.n_in && D.rows == t.n_out && o.dst_col0 + u.cout <= D.cols,
                 "vc_pass: op %d: unit/table/buffer shapes disagree (src %lld x %d, dst %lld x %d, table %lld -> %lld, %d -> %d channels)",
                 i, (long long)S.rows, S.cols, (long long)D.rows, D.cols, (long long)t.n_in, (long long)t.n_out, u.cin, u.cout);
      VC_REQUIRE(!t.subm || t.n_in == t.n_out, "vc_pass: table %d: submanifold table with n_in != n_out", o.table);
    } else if (o.kind == VC_PASS_COPY) {
      VC_REQUIRE(S.rows == D.rows && o.dst_col0 + S.cols <= D.cols, "vc_pass: op %d: copy shapes disagree", i);
    } else if (o.kind == VC_PASS_GATHER) {
      VC_REQUIRE(p->keeps && o.keep >= 0 && o.keep < p->n_keeps && p->keeps[o.keep], "vc_pass: op %d: bad keep id", i);
      VC_REQUIRE(S.cols == D.cols && o.dst_col0 == 0 && D.rows <= S.rows, "vc_pass: op %d: gather shapes disagree", i);
    } else {
      set_error("vc_pass: op %d: unknown kind %d", i, o.kind);
      return VC_EINVAL;
    }
  }
  return VC_OK;
}

// A fragment-ordered weight image pays for its share of the pack launch only when the conv re-reads W_k often enough: measured,
// VirConv-L bs 4 (76 k - 310 k rows per conv) 5.72 -> 5.58 ms per step with images, VirConv8x bs 2 (15 k - 60 k rows) 5.39 -> 5.49.
static constexpr int64_t kPackMinRows = 60000;
// vc_debug_set "pass_pack_all": -1 (default) = every unit gets an image when the split products are on (those kernels read the image on
// every route: a unit below the threshold would otherwise pack into the library's scratch right before its own launch), the
// threshold above otherwise; 0 / 1 = never / always (A/B)
int g_pass_pack_all = -1;
static inline int64_t pack_min_rows(const vc_pass_program* p) {
  // p->pack_all: the caller's snapshot of "split products on" at forward time (not the library global: a switch moved between the
  // forward and the backward pass of one call would make their layouts disagree -- ADVICE r4)
  const bool all = g_pass_pack_all < 0 ? p->pack_all != 0 : g_pass_pack_all != 0;
  return all ? 0 : kPackMinRows;
}

struct FwdLayout {
  std::vector<int64_t> buf_off;            // arena offset of every buffer (-1: external)
  std::vector<size_t> yraw_off, stats_off;  // per op (units, training only)
  std::vector<int64_t> wpk_off;             // per op: fragment-ordered weight image of the unit's conv (-1: the shape takes none)
  size_t scratch_off = 0, scratch_bytes = 0, total = 0;
};

static void fwd_layout(const vc_pass_program* p, FwdLayout& L) {
  Bump bump;
  L.buf_off.assign(p->n_bufs, -1);
  for (int b = 0; b < p->n_bufs; ++b)
    if (!p->bufs[b].external) L.buf_off[b] = (int64_t)bump.take((size_t)p->bufs[b].rows * p->bufs[b].cols * sizeof(float));
  L.yraw_off.assign(p->n_ops, 0);
  L.stats_off.assign(p->n_ops, 0);
  L.wpk_off.assign(p->n_ops, -1);
  size_t scratch = 256;
  for (int i = 0; i < p->n_ops; ++i) {
    const vc_pass_op& o = p->ops[i];
    if (o.kind != VC_PASS_UNIT) continue;
    const vc_pass_unit& u = p->units[o.unit];
    const vc_pass_table& t = p->tables[o.table];
    const int flags = t.sorted_rows ? VC_CONV_SORTED_ROWS : 0;
    if (p->operand_type == VC_OPERAND_F32 && t.n_out >= pack_min_rows(p)) {
      const size_t pk = vc_conv_packed_weight_floats(u.cin, u.cout, t.kv, 0);
      if (pk) L.wpk_off[i] = (int64_t)bump.take(pk * sizeof(float));
    }
    if (p->training) {
      L.yraw_off[i] = bump.take((size_t)t.n_out * u.cout * sizeof(float));
      L.stats_off[i] = bump.take((size_t)2 * u.cout * sizeof(float));
      const size_t w = vc_post_act_block_forward_workspace_bytes(t.n_in, t.n_out, t.kv, u.cin, u.cout, flags);
      if (w > scratch) scratch = w;
    } else {
      const size_t w = (size_t)t.n_out * u.cout * sizeof(float);  // y_raw of a unit the affine epilogue cannot serve
      if (w > scratch) scratch = w;
    }
  }
  L.scratch_off = bump.take(scratch);
  L.scratch_bytes = scratch;
  L.total = bump.off;
}

static inline float* buf_ptr(const vc_pass_program* p, const FwdLayout& L, const void* arena, int b) {
  return p->bufs[b].external ? (float*)p->bufs[b].ptr : (float*)((char*)arena + L.buf_off[b]);
}

// Repack the conv weights of the units in `ops_idx` into fragment order (ONE launch) and register the images for the conv
// launches of this call; `backward` selects the transposed image the backward-input kernels read.
static int pack_unit_weights(const vc_pass_program* p, const std::vector<int>& ops_idx, const std::vector<float*>& dst, int backward,
                             hipStream_t st) {
  const int chunk = 48;
  for (size_t b = 0; b < ops_idx.size(); b += chunk) {
    const int n = (int)std::min<size_t>(chunk, ops_idx.size() - b);
    const float* w[chunk];
    float* d[chunk];
    int ci[chunk], co[chunk], kv[chunk];
    for (int j = 0; j < n; ++j) {
      const vc_pass_op& o = p->ops[ops_idx[b + j]];
      w[j] = p->units[o.unit].weight; d[j] = dst[b + j];
      ci[j] = p->units[o.unit].cin; co[j] = p->units[o.unit].cout; kv[j] = p->tables[o.table].kv;
    }
    const int rc = vc_conv_pack_weights(n, w, ci, co, kv, backward, d, st);
    if (rc != VC_OK) return rc;
  }
  return VC_OK;
}

static hipEvent_t* pass_events() {
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

static int launch_add_views(float* out, int64_t n, int c, const float* a, int as, int ac, const float* b, int bs, int bc,
                            hipStream_t st) {
  const int c4 = c / 4;
  hipLaunchKernelGGL(add_views_kernel, dim3((unsigned)cdiv(n * c4, 256)), dim3(256), 0, st, (float4*)out, n, c4, a, as, ac, b,
                     bs, bc);
  VC_CHECK_LAUNCH("add_views_kernel");
  return VC_OK;
}

// scheduling knob (vc_debug_set "pass_dw_main_tail"): the weight gradients of the LAST k units of the reverse sweep run on the
// main stream after their backward-input conv instead of on the side stream.  The side stream's weight gradients trail the
// main chain (they can only start after each unit's BatchNorm backward), so at the end of the sweep the main stream idles in
// the join; moving the tail's weight gradients over fills that hole.
int g_pass_dw_main_tail = 0;
// conv_kernels.hip: the split-N reductions of the side-stream weight gradients of one sweep, deferred into ONE launch
void bw_defer_begin();
void bw_defer_enable(bool on);
int bw_defer_flush(hipStream_t st);
int g_pass_defer_dw_reduce = 1;   // vc_debug_set "pass_defer_dw_reduce": 0 = every weight gradient reduces at once (A/B)
// vc_debug_set "pass_dw_flush_mb": the deferred reductions are launched whenever this many MB of partial sums are waiting (0 = one launch at
// the end of the sweep, the round-5 form).  The single launch read ~170 MB in 39 us BEHIND the last weight gradient, with the main stream
// already waiting at the join; flushed on the way, the bulk (stages 4 and 3: 64 x 64 and 32 x 32 weights, 24 MB of partials per layer) is
// reduced under the main chain's kernels and the launch in the tail reads what stages 2 and 1 left (profiles/r06_step_sequence.txt).
int g_pass_dw_flush_mb = 48;
// vc_debug_set "pass_bwd_epilogue" (default 1): let the backward-input conv that delivers the LAST contribution to a buffer's
// gradient add the earlier contribution in its epilogue and, when that buffer is the whole output of a unit, also form that
// unit's BatchNorm-backward sums there (vc_conv_backward_input_epilogue): no gradient-add kernel, no reduction pass over
// (y_raw, dy) for 15 of the 20 units of VirConvL8x.  0 = every unit reduces for itself (the node-by-node arithmetic).
int g_pass_bwd_epilogue = 1;
// vc_debug_set "pass_fork_ext_event" (default 1): the weight-gradient fork takes the completion event of the unit's last main-stream
// launch (hipExtLaunchKernelGGL stop event, common.h) instead of a hipEventRecord marker packet in the main queue.
int g_pass_fork_ext_event = 1;

struct GradView {
  const float* p;
  int stride, col0;
};

// the reverse sweep; dry = only size the arena (no launches, no dereference)
static int backward_sweep(const vc_pass_program* p, const void* fwd_arena, const float* const* ext_grads, float* input_grad,
                          bool want_input_grad, char* arena, size_t arena_bytes,
                          hipStream_t side, hipStream_t st, bool dry, size_t* need_bytes) {
  FwdLayout L;
  fwd_layout(p, L);
  Bump bump;
  t_stop_event = StopEventSlot{};
  // shared scratch: BatchNorm partial sums (main stream), group-summed d_raw (main stream), weight-gradient partials (all weight
  // gradients run in order on ONE stream, the side stream when there is one)
  size_t bn_bytes = 256, grp_bytes = 256, gpart_bytes = 256, dw_bytes = 256;
  for (int i = 0; i < p->n_ops; ++i) {
    const vc_pass_op& o = p->ops[i];
    if (o.kind != VC_PASS_UNIT) continue;
    const vc_pass_unit& u = p->units[o.unit];
    const vc_pass_table& t = p->tables[o.table];
    bn_bytes = std::max(bn_bytes, vc_bn_workspace_bytes(t.n_out, u.cout));
    dw_bytes = std::max(dw_bytes, vc_conv_backward_weight_workspace_bytes(t.n_out, t.kv, u.cin, u.cout));
    if (t.rep) {
      grp_bytes = std::max(grp_bytes, (size_t)t.n_out * u.cout * sizeof(float));
      gpart_bytes = std::max(gpart_bytes, vc_group_sum_sorted_workspace_bytes(t.n_out, u.cout));
    }
  }
  const size_t bn_off = bump.take(bn_bytes), grp_off = bump.take(grp_bytes), gpart_off = bump.take(gpart_bytes),
               dw_off = bump.take(dw_bytes), dw2_off = bump.take(dw_bytes);
  auto at = [&](size_t off) -> float* { return dry ? nullptr : (float*)(arena + off); };

  // transposed fragment-ordered weight images for the backward-input convs (one pack launch for the whole sweep)
  struct ClearPacked { bool on; ~ClearPacked() { if (on) vc_conv_clear_packed_weights(); } } clear_packed_on_exit{!dry};
  if (p->operand_type == VC_OPERAND_F32) {
    std::vector<char> needs0(p->n_bufs, 0);
    needs0[0] = want_input_grad ? 1 : 0;
    for (int i = 0; i < p->n_ops; ++i) {
      const vc_pass_op& o = p->ops[i];
      if (o.kind == VC_PASS_UNIT) needs0[o.dst] = 1;
      else if (needs0[o.src]) needs0[o.dst] = 1;
    }
    std::vector<int> idx;
    std::vector<float*> dst;
    for (int i = 0; i < p->n_ops; ++i) {
      const vc_pass_op& o = p->ops[i];
      if (o.kind != VC_PASS_UNIT || !needs0[o.src] || p->tables[o.table].n_in < pack_min_rows(p)) continue;
      const size_t pk = vc_conv_packed_weight_floats(p->units[o.unit].cin, p->units[o.unit].cout, p->tables[o.table].kv, 1);
      if (!pk) continue;
      idx.push_back(i);
      dst.push_back(at(bump.take(pk * sizeof(float))));
    }
    if (!dry) {
      vc_conv_clear_packed_weights();
      const int rc = pack_unit_weights(p, idx, dst, 1, st);
      if (rc != VC_OK) return rc;
    }
  }

  auto dw_bytes_of = [](const vc_pass_table& t_, const vc_pass_unit& u_) {
    return vc_conv_backward_weight_workspace_bytes(t_.n_out, t_.kv, u_.cin, u_.cout);
  };
  if (!dry) bw_defer_begin();
  // whatever path leaves this function (error returns included): no armed stop-event slot, no deferral, and no queued split-N
  // reductions survive it -- a later direct call on this thread (vc_bn_relu_backward, vc_group_sum_sorted) must not bind its
  // launch to this sweep's event, and a failed sweep must not leave its reductions for the next one
  struct SweepGuard {
    bool dry;
    ~SweepGuard() {
      t_stop_event = StopEventSlot{};
      bw_defer_enable(false);
      if (!dry) bw_defer_begin();
    }
  } sweep_guard_on_exit{dry};
  std::vector<std::vector<GradView>> contrib(p->n_bufs);
  std::vector<int> state(p->n_bufs, 0);  // 0 unresolved, 1 resolved to `res`, 2 no gradient reaches the buffer
  std::vector<GradView> res(p->n_bufs);
  const float* marker = (const float*)(uintptr_t)256;  // dry run: any non-null stand-in for "a contribution exists"
  for (int b = 0; b < p->n_bufs; ++b)
    if (ext_grads && ext_grads[b]) contrib[b].push_back({ext_grads[b], p->bufs[b].cols, 0});

  auto resolve = [&](int b) -> int {
    if (state[b]) return VC_OK;
    auto& c = contrib[b];
    if (c.empty()) { state[b] = 2; return VC_OK; }
    if (c.size() == 1) { res[b] = c[0]; state[b] = 1; return VC_OK; }
    const vc_pass_buf& B = p->bufs[b];
    float* out = at(bump.take((size_t)B.rows * B.cols * sizeof(float)));
    if (!dry) {
      if (bump.off > arena_bytes) { set_error("vc_pass_backward: arena too small"); return VC_ECAPACITY; }
      int rc = launch_add_views(out, B.rows, B.cols, c[0].p, c[0].stride, c[0].col0, c[1].p, c[1].stride, c[1].col0, st);
      if (rc != VC_OK) return rc;
      for (size_t k = 2; k < c.size(); ++k) {
        rc = launch_add_views(out, B.rows, B.cols, out, B.cols, 0, c[k].p, c[k].stride, c[k].col0, st);
        if (rc != VC_OK) return rc;
      }
    }
    res[b] = {dry ? marker : out, B.cols, 0};
    state[b] = 1;
    return VC_OK;
  };
  // which buffers carry a gradient at all: everything downstream of a unit (it has parameters), and pure data movement
  // (copy / gather) of a buffer that does -- a discard of the raw input features, for instance, does not
  std::vector<char> needs(p->n_bufs, 0);
  needs[0] = want_input_grad ? 1 : 0;
  for (int i = 0; i < p->n_ops; ++i) {
    const vc_pass_op& o = p->ops[i];
    if (o.kind == VC_PASS_UNIT) needs[o.dst] = 1;
    else if (needs[o.src]) needs[o.dst] = 1;
  }
  auto wants_grad = [&](int b) { return needs[b] != 0; };

  // who writes a buffer as a whole (a unit whose output IS the buffer) and who reads it first (forward order = the last
  // contributor to its gradient in the reverse sweep)
  std::vector<int> prod_unit(p->n_bufs, -1), first_cons(p->n_bufs, -1), n_writers(p->n_bufs, 0);
  for (int i = 0; i < p->n_ops; ++i) {
    const vc_pass_op& o = p->ops[i];
    if (first_cons[o.src] < 0) first_cons[o.src] = i;
    ++n_writers[o.dst];
    if (o.kind == VC_PASS_UNIT && o.dst_col0 == 0 && p->bufs[o.dst].cols == p->units[o.unit].cout) prod_unit[o.dst] = i;
  }
  for (int b = 0; b < p->n_bufs; ++b)
    if (n_writers[b] != 1) prod_unit[b] = -1;
  // sums != nullptr: the epilogue launch finished them itself (conv_finish_tail), only the dx kernel is left
  struct FusedSums { const float* partial = nullptr; int64_t nblocks = 0; const float* sums = nullptr; };
  std::vector<FusedSums> fused(p->n_ops);  // per unit op: BatchNorm-backward sums delivered by a conv epilogue

  hipEvent_t* ev = (side != nullptr && !dry) ? pass_events() : nullptr;
  bool forked = false;
  size_t deferred_bytes = 0;   // partial sums waiting for their reduce launch (g_pass_dw_flush_mb)
  int n_units_left = 0;  // units still ahead in the reverse sweep (for the main-tail schedule)
  for (int i = 0; i < p->n_ops; ++i) n_units_left += p->ops[i].kind == VC_PASS_UNIT ? 1 : 0;

  for (int i = p->n_ops - 1; i >= 0; --i) {
    const vc_pass_op& o = p->ops[i];
    int rc = resolve(o.dst);
    if (rc != VC_OK) return rc;
    const vc_pass_buf &S = p->bufs[o.src], &D = p->bufs[o.dst];
    if (o.kind == VC_PASS_UNIT) {
      const vc_pass_unit& u = p->units[o.unit];
      const vc_pass_table& t = p->tables[o.table];
      const size_t wbytes = (size_t)t.kv * u.cin * u.cout * sizeof(float);
      const int units_left = n_units_left--;  // including this one
      if (state[o.dst] == 2) {  // no gradient reaches this unit: its parameters get exact zeros
        if (!dry) {
          if (u.dweight) VC_CHECK_HIP(hipMemsetAsync(u.dweight, 0, wbytes, st));
          if (u.dgamma) VC_CHECK_HIP(hipMemsetAsync(u.dgamma, 0, (size_t)u.cout * sizeof(float), st));
          if (u.dbeta) VC_CHECK_HIP(hipMemsetAsync(u.dbeta, 0, (size_t)u.cout * sizeof(float), st));
        }
        continue;
      }
      const bool need_dx = wants_grad(o.src), need_dw = u.dweight != nullptr;
      const bool dup = t.rep != nullptr && need_dx;
      const bool on_side = ev != nullptr && units_left > g_pass_dw_main_tail;
      float* d_raw = at(bump.take((size_t)t.n_out * u.cout * sizeof(float)));
      float* dx = need_dx ? at(bump.take((size_t)t.n_in * u.cin * sizeof(float))) : nullptr;
      float* dgb = (u.dgamma && u.dbeta) ? nullptr : at(bump.take((size_t)2 * u.cout * sizeof(float)));
      // Duplicate-pixel unit: the group sum feeds the backward-input conv AND the weight gradient (vc_conv_backward_weight_dup).  The
      // side stream runs its weight gradients in order but LATER than the main chain, so such a unit gets its own group buffer (the
      // shared one would be overwritten by the next duplicate-pixel unit of the main chain before the side stream has read it).
      float* grp = !dup ? nullptr : (need_dw ? at(bump.take((size_t)t.n_out * u.cout * sizeof(float))) : at(grp_off));
      // a side-stream weight gradient keeps its split-N partial sums until the sweep's single reduce launch: its own buffer
      // (the buffer is taken whenever the unit COULD run on the side stream, so that the sizing run -- which has no streams --
      // and the real run lay the arena out identically)
      const bool may_defer = need_dw && units_left > g_pass_dw_main_tail && g_pass_defer_dw_reduce;
      const size_t dwp_off = may_defer ? bump.take(dw_bytes_of(t, u)) : 0;
      const bool defer = may_defer && on_side;
      char* dwp = (defer && !dry) ? arena + dwp_off : nullptr;
      // epilogue fusion: this conv delivers the last contribution to the gradient of its source buffer
      bool fold = false;
      GradView addv{nullptr, 0, 0};
      int cprod = -1;
      float* fpart = nullptr;
      if (need_dx && g_pass_bwd_epilogue && p->operand_type == VC_OPERAND_F32 && first_cons[o.src] == i &&
          contrib[o.src].size() <= 1 && vc_conv_epilogue_supported(t.n_out, u.cout, u.cin, t.kv, VC_OPERAND_F32)) {
        fold = true;
        if (!contrib[o.src].empty()) {
          addv = contrib[o.src][0];
          contrib[o.src].clear();
        }
        cprod = prod_unit[o.src];
        if (cprod >= 0) {
          // partial rows, then 2 x cin floats for the finished sums and 256 fp64 group rows (in-kernel finish)
          const size_t pf = vc_conv_bwd_stats_partial_floats(t.n_in, u.cin, u.cout, t.order_bwd != nullptr);
          fpart = at(bump.take((pf + 2 * (size_t)u.cin) * sizeof(float) + 64 + (size_t)256 * 2 * u.cin * sizeof(double)));
          fused[cprod].partial = dry ? marker : fpart;
          fused[cprod].nblocks = (int64_t)(pf / (2 * (size_t)u.cin));
        }
      }
      if (need_dx) contrib[o.src].push_back({dry ? marker : dx, u.cin, 0});
      if (dry) continue;
      if (bump.off > arena_bytes) {   // never launch into memory the caller did not provide
        set_error("vc_pass_backward: arena too small");
        return VC_ECAPACITY;
      }
      const GradView g = res[o.dst];
      const float* x = buf_ptr(p, L, fwd_arena, o.src);
      const float* y_raw = (const float*)((const char*)fwd_arena + L.yraw_off[i]);
      const float* mean = (const float*)((const char*)fwd_arena + L.stats_off[i]);
      const float* var = mean + u.cout;
      if (dup) {
        VC_REQUIRE(t.grp_plan && (u.cout & (u.cout - 1)) == 0,
                   "vc_pass_backward: duplicate-pixel table needs its group plan (vc_pass_table.grp_plan, vc_group_sum_sorted)");
      }
      // the fork of the weight gradient waits for this unit's last main-stream launch in front of it: the BatchNorm backward's dx
      // kernel, or the group sum of a duplicate-pixel unit -- that launch carries the event
      const bool ext_fork = need_dw && on_side && g_pass_fork_ext_event != 0;
      t_stop_event = (ext_fork && !dup) ? StopEventSlot{ev[0], false} : StopEventSlot{};
      if (fused[i].sums != nullptr)
        rc = bn_bwd_dx_launch(y_raw, g.p, g.stride, g.col0 + o.dst_col0, t.n_out, u.cout, mean, var, u.gamma, u.beta, u.eps, o.relu,
                              fused[i].sums, d_raw, st);
      else if (fused[i].partial != nullptr)
        rc = vc_bn_relu_backward_from_partial(y_raw, g.p, g.stride, g.col0 + o.dst_col0, t.n_out, u.cout, mean, var, u.gamma,
                                              u.beta, u.eps, o.relu, fused[i].partial, fused[i].nblocks, d_raw,
                                              u.dgamma ? u.dgamma : dgb, u.dbeta ? u.dbeta : dgb + u.cout,
                                              nullptr, arena + bn_off, bn_bytes, st);
      else
        rc = vc_bn_relu_backward(y_raw, g.p, g.stride, g.col0 + o.dst_col0, t.n_out, u.cout, mean, var, u.gamma, u.beta, u.eps,
                                 o.relu, d_raw, u.dgamma ? u.dgamma : dgb, u.dbeta ? u.dbeta : dgb + u.cout,
                                 nullptr, arena + bn_off, bn_bytes, st);
      if (rc != VC_OK) return rc;
      if (dup) {
        if (ext_fork) t_stop_event = StopEventSlot{ev[0], false};
        rc = vc_group_sum_sorted(d_raw, t.grp_plan, t.n_out, u.cout, grp, arena + gpart_off, gpart_bytes, st);
        if (rc != VC_OK) return rc;
      }
      const bool fork_bound = t_stop_event.bound;
      t_stop_event = StopEventSlot{};
      auto weight_grad = [&](char* ws_, hipStream_t s_) -> int {
        return dup ? vc_conv_backward_weight_dup(x, d_raw, grp, t.rep, t.centre, t.pair_fwd, t.n_out, t.kv, u.cin, u.cout,
                                                 p->operand_type, u.dweight, ws_, dw_bytes, s_)
                   : vc_conv_backward_weight(x, d_raw, t.pair_fwd, t.n_out, t.kv, u.cin, u.cout, p->operand_type, u.dweight, ws_,
                                             dw_bytes, s_);
      };
      if (need_dw && on_side) {  // fork: the weight gradient only reads x (forward arena), d_raw and grp (never rewritten in this call)
        if (!fork_bound) VC_CHECK_HIP(hipEventRecord(ev[0], st));
        VC_CHECK_HIP(hipStreamWaitEvent(side, ev[0], 0));
        forked = true;
        bw_defer_enable(defer);
        rc = weight_grad(defer ? dwp : arena + dw_off, side);
        bw_defer_enable(false);
        if (rc != VC_OK) return rc;
        if (defer && !dry && g_pass_dw_flush_mb > 0) {
          deferred_bytes += dw_bytes_of(t, u);
          if (deferred_bytes >= ((size_t)g_pass_dw_flush_mb << 20) && units_left > 1) {
            rc = bw_defer_flush(side);
            if (rc != VC_OK) return rc;
            deferred_bytes = 0;
          }
        }
      }
      if (need_dx) {
        const float* src = d_raw;
        const float* src_centre = nullptr;
        if (dup) {
          src = grp;
          src_centre = d_raw;
        }
        const int flags = (t.sorted_rows && t.subm) ? VC_CONV_SORTED_ROWS : 0;
        if (fold) {
          const float *cy = nullptr, *cmean = nullptr, *cvar = nullptr, *cg = nullptr, *cb = nullptr;
          float ceps = 0.f;
          int crelu = 0;
          if (cprod >= 0) {  // the unit that produced this conv's input: its BatchNorm-backward sums are formed in the epilogue
            const vc_pass_unit& cu = p->units[p->ops[cprod].unit];
            cy = (const float*)((const char*)fwd_arena + L.yraw_off[cprod]);
            cmean = (const float*)((const char*)fwd_arena + L.stats_off[cprod]);
            cvar = cmean + cu.cout;
            cg = cu.gamma; cb = cu.beta; ceps = cu.eps; crelu = p->ops[cprod].relu;
            if (cu.dgamma && cu.dbeta) {  // this launch may finish the producing unit's sums itself
              const size_t pf = (size_t)fused[cprod].nblocks * 2 * u.cin;
              float* sums = fpart + pf;
              double* dpart = (double*)(((uintptr_t)(sums + 2 * u.cin) + 63) & ~(uintptr_t)63);
              conv_finish_arm(BnFinishRequest{1, (int64_t)p->tables[p->ops[cprod].table].n_out, cu.dbeta, cu.dgamma, sums, nullptr,
                                              nullptr, 0.f, dpart});
            }
          }
          rc = vc_conv_backward_input_epilogue(src, src_centre, t.n_out, t.subm ? t.pair_fwd : t.pair_bwd, t.n_in, t.kv, u.weight,
                                               u.cin, u.cout, t.subm ? 1 : 0, dup ? t.centre : -1, dup ? t.rep : nullptr,
                                               t.order_bwd, flags, addv.p, addv.stride, addv.col0, cy, cmean, cvar, cg, cb, ceps,
                                               crelu, fpart, dx, st);
          if (conv_finish_take(st) && cprod >= 0) fused[cprod].sums = fpart + (size_t)fused[cprod].nblocks * 2 * u.cin;
        } else {
          rc = vc_conv_backward_input(src, src_centre, t.n_out, t.subm ? t.pair_fwd : t.pair_bwd, t.n_in, t.kv, u.weight, u.cin,
                                      u.cout, t.subm ? 1 : 0, dup ? t.centre : -1, dup ? t.rep : nullptr, t.order_bwd,
                                      p->operand_type, flags, dx, st);
        }
        if (rc != VC_OK) return rc;
      }
      if (need_dw && !on_side) {  // main stream: its own partial-sum scratch (the side stream may still be using the shared one)
        rc = weight_grad(arena + dw2_off, st);
        if (rc != VC_OK) return rc;
      }
    } else if (o.kind == VC_PASS_COPY) {
      if (state[o.dst] == 2 || !wants_grad(o.src)) continue;
      contrib[o.src].push_back({res[o.dst].p, res[o.dst].stride, res[o.dst].col0 + o.dst_col0});
    } else {  // VC_PASS_GATHER: scatter the kept rows back, zeros elsewhere
      if (state[o.dst] == 2 || !wants_grad(o.src)) continue;
      GradView g = res[o.dst];
      if (g.stride != D.cols || g.col0 != 0) {
        float* dense = at(bump.take((size_t)D.rows * D.cols * sizeof(float)));
        if (!dry) {
          if (bump.off > arena_bytes) { set_error("vc_pass_backward: arena too small"); return VC_ECAPACITY; }
          rc = launch_add_views(dense, D.rows, D.cols, g.p, g.stride, g.col0, nullptr, 0, 0, st);
          if (rc != VC_OK) return rc;
        }
        g = {dry ? marker : dense, D.cols, 0};
      }
      float* gs = at(bump.take((size_t)S.rows * S.cols * sizeof(float)));
      contrib[o.src].push_back({dry ? marker : gs, S.cols, 0});
      if (dry) continue;
      if (bump.off > arena_bytes) {
        set_error("vc_pass_backward: arena too small");
        return VC_ECAPACITY;
      }
      rc = vc_scatter_rows(g.p, D.cols, p->keeps[o.keep], D.rows, S.rows, gs, st);
      if (rc != VC_OK) return rc;
    }
  }
  if (want_input_grad) {
    int rc = resolve(0);
    if (rc != VC_OK) return rc;
    if (!dry) {
      const vc_pass_buf& B = p->bufs[0];
      if (state[0] == 2) VC_CHECK_HIP(hipMemsetAsync(input_grad, 0, (size_t)B.rows * B.cols * sizeof(float), st));
      else {
        rc = launch_add_views(input_grad, B.rows, B.cols, res[0].p, res[0].stride, res[0].col0, nullptr, 0, 0, st);
        if (rc != VC_OK) return rc;
      }
    }
  }
  if (forked) {  // join: every weight gradient is complete before the caller's stream moves on (and may recycle the arenas)
    const int rcf = bw_defer_flush(side);   // ONE reduce launch for all deferred split-N partial sums
    if (rcf != VC_OK) return rcf;
    VC_CHECK_HIP(hipEventRecord(ev[1], side));
    VC_CHECK_HIP(hipStreamWaitEvent(st, ev[1], 0));
  }
  if (need_bytes) *need_bytes = bump.off;
  if (!dry && bump.off > arena_bytes) {  // cannot happen when the caller sized the arena with the same arguments
    set_error("vc_pass_backward: arena too small");
    return VC_ECAPACITY;
  }
  return VC_OK;
}

}  // namespace vc

using namespace vc;

extern "C" {

size_t vc_pass_forward_arena_bytes(const vc_pass_program* prog) {
  if (check_program(prog) != VC_OK) return 0;
  FwdLayout L;
  fwd_layout(prog, L);
  return L.total;
}

int vc_pass_forward(const vc_pass_program* p, void* arena, size_t arena_bytes, int64_t* buf_offsets, void* stream) {
  int rc = check_program(p);
  if (rc != VC_OK) return rc;
  VC_REQUIRE(arena, "vc_pass_forward: null arena");
  FwdLayout L;
  fwd_layout(p, L);
  if (arena_bytes < L.total) { set_error("vc_pass_forward: arena too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  if (buf_offsets)
    for (int b = 0; b < p->n_bufs; ++b) buf_offsets[b] = L.buf_off[b];
  char* scratch = (char*)arena + L.scratch_off;
  {  // fragment-ordered weight images for every conv whose shape takes one (conv_kernels.hip, "fragment-ordered weight images")
    std::vector<int> idx;
    std::vector<float*> dst;
    for (int i = 0; i < p->n_ops; ++i)
      if (p->ops[i].kind == VC_PASS_UNIT && L.wpk_off[i] >= 0) {
        idx.push_back(i);
        dst.push_back((float*)((char*)arena + L.wpk_off[i]));
      }
    vc_conv_clear_packed_weights();
    rc = pack_unit_weights(p, idx, dst, 0, st);
    if (rc != VC_OK) { vc_conv_clear_packed_weights(); return rc; }
  }
  struct ClearPacked { ~ClearPacked() { vc_conv_clear_packed_weights(); } } clear_packed_on_exit;
  for (int i = 0; i < p->n_ops; ++i) {
    const vc_pass_op& o = p->ops[i];
    const vc_pass_buf &S = p->bufs[o.src], &D = p->bufs[o.dst];
    const float* x = buf_ptr(p, L, arena, o.src);
    float* y = buf_ptr(p, L, arena, o.dst);
    if (o.kind == VC_PASS_UNIT) {
      const vc_pass_unit& u = p->units[o.unit];
      const vc_pass_table& t = p->tables[o.table];
      const int flags = t.sorted_rows ? VC_CONV_SORTED_ROWS : 0;
      if (p->training) {
        float* y_raw = (float*)((char*)arena + L.yraw_off[i]);
        float* mean = (float*)((char*)arena + L.stats_off[i]);
        rc = vc_post_act_block_forward(x, t.n_in, t.pair_fwd, t.n_out, t.kv, u.weight, u.cin, u.cout, t.order_fwd,
                                       p->operand_type, flags, u.gamma, u.beta, u.running_mean, u.running_var,
                                       u.num_batches_tracked, u.momentum, u.eps, o.relu, y_raw, y, D.cols, o.dst_col0, mean,
                                       mean + u.cout, scratch, L.scratch_bytes, stream);
      } else if (D.cols == u.cout && p->operand_type == VC_OPERAND_F32 &&
                 vc_conv_epilogue_supported(t.n_in, u.cin, u.cout, t.kv, VC_OPERAND_F32)) {
        rc = vc_conv_forward_epilogue(x, t.n_in, t.pair_fwd, t.n_out, t.kv, u.weight, u.cin, u.cout, t.order_fwd, VC_EPI_AFFINE,
                                      flags, nullptr, u.running_mean, u.running_var, u.gamma, u.beta, u.eps, o.relu, y, stream);
      } else {
        rc = vc_conv_forward(x, t.n_in, t.pair_fwd, t.n_out, t.kv, u.weight, u.cin, u.cout, t.order_fwd, p->operand_type, flags,
                             (float*)scratch, stream);
        if (rc != VC_OK) return rc;
        rc = vc_bn_apply_relu((const float*)scratch, t.n_out, u.cout, u.running_mean, u.running_var, u.gamma, u.beta, u.eps,
                              o.relu, y, D.cols, o.dst_col0, stream);
      }
    } else if (o.kind == VC_PASS_COPY) {
      hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)cdiv(S.rows * (S.cols / 4), 256)), dim3(256), 0, st, (const float4*)x,
                         S.rows, S.cols / 4, y, D.cols, o.dst_col0);
      VC_CHECK_LAUNCH("copy_cols_kernel");
      rc = VC_OK;
    } else {
      rc = vc_gather_rows(x, nullptr, S.cols, 0, p->keeps[o.keep], D.rows, y, nullptr, stream);
    }
    if (rc != VC_OK) return rc;
  }
  return VC_OK;
}

size_t vc_pass_backward_arena_bytes(const vc_pass_program* prog, const float* const* ext_grads, int need_input_grad) {
  if (check_program(prog) != VC_OK || !prog->training) return 0;
  size_t need = 0;
  if (backward_sweep(prog, nullptr, ext_grads, nullptr, need_input_grad != 0, nullptr, 0, nullptr, nullptr, true,
                     &need) != VC_OK)
    return 0;
  return need;
}

int vc_pass_backward(const vc_pass_program* p, const void* fwd_arena, size_t fwd_arena_bytes, const float* const* ext_grads,
                     float* input_grad, void* arena, size_t arena_bytes,
                     void* side_stream, void* stream) {
  int rc = check_program(p);
  if (rc != VC_OK) return rc;
  VC_REQUIRE(p->training, "vc_pass_backward: the program ran with running statistics (training = 0): no backward");
  VC_REQUIRE(fwd_arena && arena && ext_grads, "vc_pass_backward: null argument");
  FwdLayout L;
  fwd_layout(p, L);
  if (fwd_arena_bytes < L.total) { set_error("vc_pass_backward: forward arena smaller than the program's layout"); return VC_ECAPACITY; }
  size_t need = 0;
  rc = backward_sweep(p, nullptr, ext_grads, nullptr, input_grad != nullptr, nullptr, 0, nullptr, nullptr, true, &need);
  if (rc != VC_OK) return rc;
  if (arena_bytes < need) { set_error("vc_pass_backward: arena too small"); return VC_ECAPACITY; }
  return backward_sweep(p, fwd_arena, ext_grads, input_grad, input_grad != nullptr, (char*)arena,
                        arena_bytes, (hipStream_t)side_stream, (hipStream_t)stream, false, nullptr);
}

}  // extern "C"
