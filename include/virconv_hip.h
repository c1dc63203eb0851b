/*
 * virconv_hip.h -- C ABI of libvirconv_hip.so: the MI355X (gfx950) Virtual-Sparse-Convolution hot path.
 *
 * Drop-in boundary.  In the reference (hailanyi/VirConv) this arithmetic is NOT in the
 * repository: it is reached through the un-vendored `spconv` pip package (setup.py:41, README.md:52,60,70)
 * from pcdet/utils/spconv_utils.py:33-36 and pcdet/datasets/processor/data_processor.py:14-41.  The
 * reference's own native-op convention (the lower half we keep) is pcdet/ops/pointnet2/pointnet2_stack:
 *   Python caller pre-allocates torch tensors, asserts contiguity      voxel_query_utils.py:31-37
 *   C++ wrapper extracts raw pointers and calls a plain launcher        src/voxel_query.cpp:25-41
 *   launcher(int..., const float*, const int*, ...) on a stream         src/voxel_query_gpu.cu:92-113
 * Each entry point below cites the spconv interface (via its reference call site) that it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; the caller owns all memory (no allocation here;
 *     scratch is passed in, sized by the matching *_workspace_bytes query)
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream; never the legacy default)
 *   - features are float32 row-major (N, C); indices are int32 row-major (N, ndim+1) = [b, z, y, x] (3-D) or
 *     [b, u, v] (2-D); spatial shapes are host int32[ndim] in the same axis order
 *   - conv weights are the spconv-2.x canonical layout (Cout, kz, ky, kx, Cin) == (Cout, KV, Cin), float32
 *     (detector3d_template.py:358-370), so released checkpoints load unchanged
 *   - kernel offsets kappa are enumerated row-major over (kz, ky, kx); pair tables are dense int32
 *     (KV, N) with -1 = no neighbour
 *   - return value: VC_OK (0) or a negative vc_status; never exit()/abort (the reference ops fprintf+exit(-1):
 *     voxel_query.cpp:10-22); vc_last_error() gives a thread-local message
 *   - re-entrant; all launches are asynchronous on `stream`; no host synchronisation inside except where
 *     a function is documented as returning a host count
 */
#ifndef VIRCONV_HIP_H_
#define VIRCONV_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vc_status {
  VC_OK = 0,
  VC_EINVAL = -1,    /* bad argument (null pointer, unsupported channel count, ndim, ...) */
  VC_ECAPACITY = -2, /* workspace / output capacity too small */
  VC_EHIP = -3       /* HIP runtime error (see vc_last_error) */
} vc_status;

/* arithmetic type of the MFMA operands of the conv kernels (accumulation is always fp32; tensors are always fp32) */
typedef enum vc_operand {
  VC_OPERAND_F32 = 0, /* fp32 products, the default, the 1e-4 parity path: operands cut EXACTLY into three bf16 pieces, six of the nine
                         cross terms on v_mfma_f32_16x16x32_bf16 (the dropped ones stay below 2^-24 of a product), fp32 accumulation --
                         or, for layers with fewer than 16 channels and with vc_debug_set("f32_split" / "bw_split", 0), exact products
                         on v_mfma_f32_16x16x4_f32 */
  VC_OPERAND_F16 = 1, /* v_mfma_f32_16x16x16_f16 : operands rounded to fp16 in registers */
  VC_OPERAND_BF16 = 2 /* v_mfma_f32_16x16x16_bf16: operands rounded to bf16 in registers */
} vc_operand;

const char* vc_version(void);
/* Layout version of the structs that cross this boundary (vc_plan_desc, vc_pass_program, vc_trace_record, ...).  A binding compares
 * vc_abi_version() with the VC_ABI_VERSION it was written against before it fills any of them: a field added to a struct (round 5:
 * vc_pass_program.pack_all; round 6: vc_plan_desc.allow_unfenced_projection, vc_adam_tensor) otherwise reads as a silent zero on the other side. */
#define VC_ABI_VERSION 7
int vc_abi_version(void);
const char* vc_last_error(void);
/* developer switch for A/B measurements (tools/kbench.py): "conv_variant" = 1 | 2 */
int vc_debug_set(const char* key, int value);
/* developer counters: "conv_bn_finish_launches" = conv launches of this process that finished their BatchNorm sums in-kernel */
int vc_debug_get(const char* key, int64_t* value);
/* developer check (tests): a consumer on a second stream behind a slow producer on `stream_a`; mode 1 = the dependency is the
 * producer launch's own completion event (hipExtLaunchKernelGGL, what the feature pass uses for its weight-gradient forks),
 * 0 = hipEventRecord, 2 = none (negative control).  buf, out: n int32 each; out[i] = what the consumer read (1 = produced). */
int vc_debug_stop_event_dependency(int32_t* buf, int32_t* out, int64_t n, int spin, int mode, void* stream_a);

/* ------------------------------------------------------------------------------------------------ K3 hash
 * Coordinate -> row hash (open addressing, 64-bit linearised key, duplicate rule rep(c) = max row; SURVEY
 * App-A.5).  Locality-preserving: the 8 keys of an aligned x-octet occupy 8 consecutive slots, so the x-1/x/x+1 probes
 * of x-consecutive rows share cache lines; workspace = 96 bytes per 2n-rounded-up-to-a-power-of-two rows.  Replaces cumm's LinearHashTable insert inside spconv's indice generation, reached from every
 * spconv.SubMConv3d/SubMConv2d forward (spconv_backbone.py:89,113).                                         */
size_t vc_hash_workspace_bytes(int64_t n);
int vc_hash_build(const int32_t* indices, int64_t n, int ndim, const int32_t* host_spatial_shape,
                  void* hash_ws, size_t hash_ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ K4 subm rulebook
 * pair_fwd[k, i] = rep(coord_i + (kappa_k - ksize/2) * dilation) or -1; the centre tap of row i is i.
 * rep_out (nullable, int32[n]) receives rep(coord_i) (== i when coordinates are unique).
 * Replaces spconv generate_subm_conv_inds (call sites spconv_backbone.py:89,113).                            */
int vc_subm_rulebook(const int32_t* indices, int64_t n, int ndim, const int32_t* host_spatial_shape,
                     const int32_t* host_ksize, const int32_t* host_dilation, const void* hash_ws,
                     size_t hash_ws_bytes, int32_t* pair_fwd, int32_t* rep_out, void* stream);

/* ------------------------------------------------------------------------------------------------ K5 strided rulebook
 * Regular sparse conv index generation: p = q*stride - pad + kappa*dil.  Output rows are all in-bounds q with
 * an active (p, kappa), in ASCENDING linear-index order (spconv-CUDA order).  Implemented
 * with an occupancy bitmap over the output grid + popcount prefix scan (no sort, no hash).
 *   step 1  vc_spconv_mark_count : marks candidates, scans, writes the output-row count to *n_out_dev (device int32)
 *   step 2  (caller reads n_out, allocates out_indices / pair_fwd / pair_bwd)
 *   step 3  vc_spconv_emit_pairs : writes out_indices (n_out, ndim+1), pair_fwd (KV, n_out), pair_bwd (KV, n)
 * Replaces spconv generate_conv_inds_stage1 / unique / stage2 (call sites spconv_backbone.py:92,116,561-563). */
size_t vc_spconv_workspace_bytes(int batch_size, int ndim, const int32_t* host_out_shape);
int vc_spconv_mark_count(const int32_t* indices, int64_t n, int ndim, int batch_size,
                         const int32_t* host_out_shape, const int32_t* host_ksize, const int32_t* host_stride,
                         const int32_t* host_padding, const int32_t* host_dilation, void* ws, size_t ws_bytes,
                         int32_t* n_out_dev, void* stream);
int vc_spconv_emit_pairs(const int32_t* indices, int64_t n, int ndim, int batch_size,
                         const int32_t* host_out_shape, const int32_t* host_ksize, const int32_t* host_stride,
                         const int32_t* host_padding, const int32_t* host_dilation, const void* ws,
                         size_t ws_bytes, int64_t n_out, int32_t* out_indices, int32_t* pair_fwd,
                         int32_t* pair_bwd, void* stream);

/* A CHAIN of strided convs (stage 2 -> 3 -> 4 -> conv_out of the backbone when no layer discard sits between them) needs only
 * ONE host read for all of its output counts: every level's stage 1 can consume the previous level's output coordinates while
 * their count is still on the device, and the coordinate emission takes a row capacity instead of the exact count
 * ("capacity + device-side count"):
 *   vc_spconv_mark_count_dev : vc_spconv_mark_count whose input holds *n_dev (device int32) valid rows of n_capacity allocated
 *   vc_spconv_emit_indices   : the coordinate half of step 3 -- out_indices rows [0, count) for count <= capacity
 *   vc_spconv_pairs          : the table half of step 3, after vc_spconv_emit_indices on the same workspace, exact n / n_out  */
int vc_spconv_mark_count_dev(const int32_t* indices, int64_t n_capacity, const int32_t* n_dev, int ndim, int batch_size,
                             const int32_t* host_out_shape, const int32_t* host_ksize, const int32_t* host_stride,
                             const int32_t* host_padding, const int32_t* host_dilation, void* ws, size_t ws_bytes,
                             int32_t* n_out_dev, void* stream);
int vc_spconv_emit_indices(int ndim, int batch_size, const int32_t* host_out_shape, void* ws, size_t ws_bytes, int64_t capacity,
                           int32_t* out_indices, void* stream);
int vc_spconv_pairs(const int32_t* indices, int64_t n, int ndim, int batch_size, const int32_t* host_out_shape,
                    const int32_t* host_ksize, const int32_t* host_stride, const int32_t* host_padding,
                    const int32_t* host_dilation, const void* ws, size_t ws_bytes, int64_t n_out, int32_t* pair_fwd,
                    int32_t* pair_bwd, void* stream);

/* ------------------------------------------------------------------------------------------------ K6/K7 gather-GEMM
 * Output-stationary implicit GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32), no atomics, run-to-run bitwise stable:
 *     out[o, :] = sum_k  src_k[ tbl[k, o], : ] @ Wsel(k)
 * forward      : src = x (n_src, cin),  tbl = pair_fwd (KV, n_out), Wsel(k) = W[:, k, :]^T      -> out (n_out, cout)
 * backward-in  : src = dy (n_src, cout), tbl = pair_bwd (KV, n_out=n_in rows), Wsel(k) = W[:, k', :]
 *                with k' = mirror ? KV-1-k : k                                                   -> out (n_in, cin)
 *   SubM backward passes tbl = pair_fwd and mirror = 1 (pair_bwd[k] == pair_fwd[KV-1-k]).
 *   Duplicate-coordinate SubM backward (2-D image-space branch): src_centre = dy is used for the
 *   centre tap, src = group-summed dy (vc_group_sum_sorted) for the others, and rows with rep[o] != o take the centre
 *   tap only.  Pass centre = -1, rep = NULL, src_centre = NULL when not needed.
 *   operand_type: VC_OPERAND_F32 (fp32 products, see vc_operand; the parity path) | VC_OPERAND_F16 | VC_OPERAND_BF16 -- tensors stay
 *   fp32 in memory; with F16 / BF16 the MFMA operands are rounded (RNE) to 16 bit in registers and accumulate in fp32 (BASELINE configs[4],
 *   "fp16 MFMA contraction"; the reference has no reduced-precision path, tolerance 2e-2 relative).  Layers with fewer than
 *   16 channels on either side always contract in fp32.
 *   row_order (optional, NULL = natural order): a permutation of [0, n_out) from vc_row_order; tile slot s computes output
 *   row row_order[s].  A pure scheduling hint -- results are bit-identical with and without it.
 * Replaces spconv ops.implicit_gemm / indice_conv fwd and bwd-input (autograd of spconv_backbone.py:89-125).   */
/*   flags: VC_CONV_SORTED_ROWS -- the rows of `tbl` AND the rows they point to are both in ascending coordinate order (true
 *   for every SubM conv on a tensor produced by a strided conv, forward and backward): the three dx-offsets of one (dz, dy)
 *   then gather one nearly contiguous run of source rows per 16-row tile, and the kernel stages that run through LDS ("LDS
 *   staging of the per-kernel-offset feature gathers") instead of gathering L2 -> registers per offset; runs that do not fit
 *   the 32-row window gather directly.  Like row_order a pure performance hint: results are bit-identical either way.    */
typedef enum vc_conv_flags { VC_CONV_SORTED_ROWS = 1, VC_CONV_SRC_INTERLEAVED = 2 } vc_conv_flags;
/*   VC_CONV_SRC_INTERLEAVED (round-3 experiment, vc_conv_forward only, channel counts multiples of 16, a packed weight image
 *   registered): `x` is given in 16-row groups, chunk-major inside a group -- [row / 16][cin / 4][row % 16][4] floats, rows padded to
 *   a multiple of 16 -- instead of row-major; results are bit-identical, the gathers of consecutive rows coalesce.              */
int vc_conv_forward(const float* x, int64_t n_in, const int32_t* pair_fwd, int64_t n_out, int kv,
                    const float* weight, int cin, int cout, const int32_t* row_order, int operand_type, int flags,
                    float* y, void* stream);
int vc_conv_backward_input(const float* dy, const float* dy_centre, int64_t n_src, const int32_t* tbl,
                           int64_t n_in, int kv, const float* weight, int cin, int cout, int mirror, int centre,
                           const int32_t* rep, const int32_t* row_order, int operand_type, int flags, float* dx,
                           void* stream);

/* Fragment-ordered weight images.  The gather-GEMMs re-read W_k once per 64 output rows and kernel offset; from the canonical
 * (Cout, KV, Cin) layout those reads run at a third of the rate of contiguous ones (a quad of lanes touches four weight rows
 * KV*Cin*4 bytes apart; tools/ubench/gather_ubench.hip: 9.3 vs 30 TB/s).  vc_conv_pack_weights repacks up to 48 weight tensors
 * in ONE launch into the order the MFMA B fragments are consumed in -- backward = 0: the image vc_conv_forward* reads,
 * backward = 1: the transposed image vc_conv_backward_input* reads -- and registers (weight pointer, direction) -> image for the
 * calling thread: conv launches that follow look their weight pointer up and read the image instead.  The caller owns the
 * memory (vc_conv_packed_weight_floats floats per tensor; 0 = the shape takes no image, pass packed[i] = NULL), must repack
 * after every weight update, and ends the association with vc_conv_clear_packed_weights.  Results are bit-identical with and
 * without an image.  (spconv's implicit-GEMM path keeps a reordered filter copy the same way.)                            */
size_t vc_conv_packed_weight_floats(int cin, int cout, int kv, int backward);
int vc_conv_pack_weights(int n, const float* const* weights, const int* cin, const int* cout, const int* kv, int backward,
                         float* const* packed, void* stream);
int vc_conv_clear_packed_weights(void);

/* Forward conv with a BatchNorm epilogue (fp32 operands; query vc_conv_epilogue_supported for the shape first):
 *   VC_EPI_STATS   training: besides y the kernel writes per-channel (sum, sum of squares) partial rows [rows][2][cout], one
 *                  per 16-row wave tile (no barrier in the epilogue); vc_conv_stats_partial_floats gives the size -- for
 *                  vc_bn_stats_from_partial: the statistics pass reads 1/8 of the bytes of y instead of y;
 *   VC_EPI_AFFINE  eval: y = relu?(conv * (gamma / sqrt(var + eps)) + (beta - mean * gamma / sqrt(var + eps))) in the store,
 *                  i.e. conv + BatchNorm1d(eval) + ReLU (spconv_backbone.py:101-105) as ONE launch.                     */
typedef enum vc_epilogue { VC_EPI_NONE = 0, VC_EPI_STATS = 1, VC_EPI_AFFINE = 2 } vc_epilogue;
int vc_conv_epilogue_supported(int64_t n_in, int cin, int cout, int kv, int operand_type);
size_t vc_conv_stats_partial_floats(int64_t n_in, int64_t n_out, int cin, int cout, int kv, int flags);
int vc_conv_forward_epilogue(const float* x, int64_t n_in, const int32_t* pair_fwd, int64_t n_out, int kv,
                             const float* weight, int cin, int cout, const int32_t* row_order, int epilogue, int flags,
                             float* stats_partial, const float* mean, const float* var, const float* gamma,
                             const float* beta, float eps, int relu, float* y, void* stream);

/* Backward-input conv with an epilogue (fp32 operands; availability: vc_conv_epilogue_supported(n_src, cout, cin, kv, F32)):
 *   dx = conv^T(dy) + addend[:, add_col0 : add_col0 + cin]      (addend optional: a second gradient contribution of the same
 *        tensor given as a strided view, e.g. the NRConvBlock concat slice or the gradient arriving from a head)
 *   and, when y_raw != NULL, the BatchNorm-backward sums of the unit that PRODUCED this conv's input (its pre-BatchNorm
 *   output y_raw (n_in, cin), batch mean / var, gamma, beta, eps, relu):  per channel sum(d) and sum(d * xhat) with
 *   d = dx masked by that unit's ReLU -- written as per-wave partial rows [rows][2][cin]
 *   (vc_conv_bwd_stats_partial_floats floats), consumed by vc_bn_relu_backward_from_partial.  This removes the unit's own
 *   reduction pass over (y_raw, dy) and the kernel that would add the two contributions.                               */
size_t vc_conv_bwd_stats_partial_floats(int64_t n_in, int cin, int cout, int row_ordered /* the launch will take a row_order */);
int vc_conv_backward_input_epilogue(const float* dy, const float* dy_centre, int64_t n_src, const int32_t* tbl, int64_t n_in,
                                    int kv, const float* weight, int cin, int cout, int mirror, int centre, const int32_t* rep,
                                    const int32_t* row_order, int flags, const float* addend, int add_stride, int add_col0,
                                    const float* y_raw, const float* mean, const float* var, const float* gamma,
                                    const float* beta, float eps, int relu, float* stats_partial, float* dx, void* stream);

/* Row permutation that makes the gather-GEMM's 16-row tiles homogeneous: within each window of `window` (1024 | 2048 |
 * 4096) consecutive rows of `tbl` (KV, n) the rows are stably sorted by their active-offset bit mask (bit k set <=> tbl[k, r] >= 0; rows with
 * rep[r] != r count as {centre} only, matching the duplicate-pixel backward).  kv <= 32.  order (n) int32.
 * spconv's implicit-GEMM path does the equivalent with its `mask_argsort` (SURVEY §8 a7 "mask argsort"); here it is
 * one LDS bitonic sort per window and never changes results.                                                      */
int vc_row_order(const int32_t* tbl, int64_t n, int kv, const int32_t* rep, int centre, int window, int32_t* order,
                 void* stream);

/* Row order for the backward-input of a duplicate-coordinate SubM conv: a stable partition of [0, n) with the representative
 * rows (rep[r] == r: the only ones that own non-centre taps) first.  Tiles of the second part issue one offset instead of the
 * tile union.  Pass the result as row_order of vc_conv_backward_input; like every row order it never changes results.      */
size_t vc_rep_order_workspace_bytes(int64_t n);
int vc_rep_order(const int32_t* rep, int64_t n, int32_t* order, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ K8 weight gradient
 * dW[:, k, :] = sum_o dy[o, :]^T (outer) x[pair_fwd[k, o], :]; wave-ballot compaction of the active pairs, MFMA
 * outer products, deterministic two-stage split-N reduction through `ws`.
 * Replaces spconv implicit_gemm bwd-weight (autograd of spconv_backbone.py:89-125).                            */
size_t vc_conv_backward_weight_workspace_bytes(int64_t n_out, int kv, int cin, int cout);
int vc_conv_backward_weight(const float* x, const float* dy, const int32_t* pair_fwd, int64_t n_out, int kv,
                            int cin, int cout, int operand_type, float* dweight, void* ws, size_t ws_bytes,
                            void* stream);
/* Duplicate-coordinate SubM tables (2-D image-space branch): all rows of a pixel group read the same neighbour row through a
 * non-centre offset, so those offsets are summed over the REPRESENTATIVES only against the group-summed gradient dy_grp
 * (vc_group_sum_sorted; the backward-input conv needs it anyway), the centre offset over every row against dy.  Same dW up to
 * the order of the fp32 additions, 2-5x fewer pairs.                                                                   */
int vc_conv_backward_weight_dup(const float* x, const float* dy, const float* dy_grp, const int32_t* rep, int centre,
                                const int32_t* pair_fwd, int64_t n_out, int kv, int cin, int cout, int operand_type,
                                float* dweight, void* ws, size_t ws_bytes, void* stream);

/* dy_grp[rep[i], :] = sum over the rows i sharing representative rep[i] of dy[i, :]  (rows that are nobody's
 * representative are NOT written: vc_conv_backward_input reads dy_grp at representatives only, and 47-81 % of the rows of
 * the image-space tensors are not representatives).  Only used by the duplicate-coordinate SubM backward.
 * The order of the additions is fixed by the DATA: the caller sorts the rows once per table by representative -- keys from vc_group_keys
 * (keys[i] = rep[i] < 0 ? i : rep[i]), any STABLE ascending sort of (keys, row ids) -- and passes
 * grp_plan = [order (n) | sorted keys (n)] int32.  Runs of equal keys are summed in ascending sorted position (plain fp32 adds),
 * runs cut by a 32-row chunk border through one partial per chunk, combined in chunk order: no atomics, no max|dy| pass, no
 * 8-byte accumulators, bit-stable.  c must be a power of two.  (Rounds 1-2 carried the sum in 64-bit fixed point with atomics:
 * csrc/experiments/group_sum_fixed.inc, -DVC_EXPERIMENTS builds only.)                                                       */
int vc_group_keys(const int32_t* rep, int64_t n, int32_t* keys, void* stream);
/* ... or let the library build the plan: keys + one stable radix sort of the (key, row) pairs.  grp_plan: 2 * n int32.           */
size_t vc_group_plan_workspace_bytes(int64_t n);
int vc_group_plan(const int32_t* rep, int64_t n, int32_t* grp_plan, void* ws, size_t ws_bytes, void* stream);
size_t vc_group_sum_sorted_workspace_bytes(int64_t n, int c);
int vc_group_sum_sorted(const float* dy, const int32_t* grp_plan, int64_t n, int c, float* dy_grp, void* ws, size_t ws_bytes,
                        void* stream);

/* Weighted sum (fused elementwise product + full reduction): out[0] = sum_{b < nb} sum_{i < e} x[b * e + i] * g[i]; e % 4 == 0.
 * One pass over x, no temporary; deterministic (fixed partition, fp64 partials, fixed-order second stage).  The contraction the
 * benchmark's stand-in detection loss is made of (the heads are out of scope) -- replaces torch's
 * `(x * g).sum()` there.  vc_weighted_sum_backward: dx[b * e + i] = gout[0] * g[i] for b < nb_out (nb_out = 1: the row every
 * sample shares, for a broadcast view). */
size_t vc_weighted_sum_workspace_bytes(int64_t nb, int64_t e);
int vc_weighted_sum(const float* x, int64_t nb, int64_t e, const float* g, float* out, void* ws, size_t ws_bytes, void* stream);
int vc_weighted_sum_backward(const float* gout, const float* g, int64_t nb_out, int64_t e, float* dx, void* stream);

/* Gradient clip + optimizer step over a few flat fp32 parameter vectors (row a16).  Replaces, for a model whose parameters alias one
 * tensor per native pass (feature_pass.flatten_parameters), tools/train_utils/train_utils.py:50-51
 *     clip_grad_norm_(model.parameters(), optim_cfg.GRAD_NORM_CLIP); optimizer.step()
 * with the optimizer of tools/train_utils/optimization/__init__.py:19-32 (`adam_onecycle`: Adam, betas (0.9, 0.99), true_wd, bn_wd)
 * stepped as fastai_optim.py:132-149 does: p *= 1 - wd * lr, then Adam with weight_decay 0 (= torch.optim.AdamW):
 *     total_norm = sqrt(sum over ALL tensors of ||grad||^2);  coef = min(max_norm / (total_norm + 1e-6), 1)  (max_norm <= 0: coef = 1)
 *     g = grad * coef;  m += (1 - beta1) * (g - m);  v = beta2 * v + (1 - beta2) * g * g
 *     p = p * (1 - lr * wd) - lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `step` counts from 1; lr and beta1 are per call (the one-cycle schedule moves both every iteration).  Two launches, no atomics, no
 * state in the workspace (nothing to zero): run-to-run bit-stable.  total_norm: optional device float.  scale_grads != 0 writes g back
 * into `grad` (what clip_grad_norm_ leaves in .grad); otherwise `grad` is only read.  Tensors with n = 0 are skipped.            */
#define VC_ADAM_MAX_TENSORS 16
typedef struct vc_adam_tensor {
  float* param;        /* n floats, updated in place */
  float* grad;         /* n floats */
  float* exp_avg;      /* n floats, zero before step 1 */
  float* exp_avg_sq;   /* n floats, zero before step 1 */
  int64_t n;
} vc_adam_tensor;
size_t vc_clip_adamw_workspace_bytes(int n_tensors);
int vc_clip_adamw(const vc_adam_tensor* tensors, int n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, float max_norm, int scale_grads, float* total_norm, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ K9 projection
 * Voxel index -> image pixel index.  Replaces index2points + index2uv +
 * X_TRANS.backward_with_param + Calibration.lidar_to_rect_cuda/rect_to_img_cuda
 * (spconv_backbone.py:8-24,54-83; X_transform.py:139-154; calibration_kitti.py:120-153).
 *   calib : (B, 33) float32 = V2C (3x4 row-major) | R0 (3x3) | P2 (3x4)
 *   trans : (B, 3)  float32 = [rot, flip, scale] or NULL
 *   params: (B, 32) float32 scratch written by vc_project_prepare, read by vc_project_uv
 *   uv    : (n, 3) int32 [b, u, v];  depth (nullable): (n,) float32
 * CAUTION (LOG.md A.17, measured on MI355X / ROCm 7.2): while waves of this kernel share a compute unit with waves of this library's
 * bf16-split conv kernels (vc_conv_forward / vc_conv_backward_input / vc_conv_backward_weight at >= 32 channels with f32_split /
 * bw_split on, the default), lanes 48-63 of some of its waves compute wrong pixels.  Launch it where it cannot overlap them: on the
 * stream of the feature passes, behind an event recorded there (what vc_plan_finish does with tables_wait_event), or on a stream
 * whose CU mask is disjoint from theirs (profiles/r06_a17_cu_mask.md).                                            */
int vc_project_prepare(const float* calib, const float* trans, int batch_size, float* params, void* stream);
int vc_project_uv(const int32_t* indices, int64_t n, const float* params, int batch_size, int stride,
                  int32_t* uv, float* depth, void* stream);

/* ------------------------------------------------------------------------------------------------ K2 discard
 * Row gather of (features, indices) by keep indices: the device half of layer_voxel_discard
 * (spconv_backbone.py:134-147: features[randoms], indices[randoms]).                                            */
int vc_gather_rows(const float* features, const int32_t* indices, int c, int icols, const int64_t* keep,
                   int64_t n_keep, float* features_out, int32_t* indices_out, void* stream);
/* grad_in[keep[j], :] = grad_out[j, :], other rows zero (keep indices are unique).                              */
/* keep[i] = P_seed(i), i < n_keep, for a pseudo-random permutation P_seed of [0, n) (4-round Feistel + cycle walking): the
 * rows layer_voxel_discard keeps, `perm[:n_keep]` (spconv_backbone.py:137-141), without sorting n random keys.          */
int vc_random_keep(int64_t n, int64_t n_keep, uint64_t seed, int64_t* keep, void* stream);
int vc_scatter_rows(const float* grad_out, int c, const int64_t* keep, int64_t n_keep, int64_t n_in,
                    float* grad_in, void* stream);

/* ------------------------------------------------------------------------------------------------ K10 dense
 * SparseConvTensor.dense(): (B, C, *spatial) <- rows; `dense` must be zero-filled by the caller
 * (height_compression.py:29).  The backward is a gather (vc_from_dense).                                        */
int vc_to_dense(const float* features, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                const int32_t* host_spatial_shape, float* dense, void* stream);
int vc_from_dense(const float* dense, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                  const int32_t* host_spatial_shape, float* features, void* stream);
/* Write-once form of .dense() for HeightCompression (height_compression.py:27-31: encoded_spconv_tensor.dense() viewed as
 * (B, C*D, H, W), the BEV map the 2-D backbone reads): `dense` need NOT be zero-filled -- a row-id volume in `ws`
 * (batch * prod(shape) int32) decides per cell, and every element of `dense` is written exactly once (zeros where no voxel is
 * active), coalesced along x.  Same result as zero-fill + vc_to_dense (highest row wins on duplicate coordinates).          */
size_t vc_to_dense_fill_workspace_bytes(int batch_size, int ndim, const int32_t* host_spatial_shape);
int vc_to_dense_fill(const float* features, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                     const int32_t* host_spatial_shape, float* dense, void* ws, size_t ws_bytes, void* stream);
/* SURVEY §8f rank 3 -- HeightCompression emitting the BEV map in the form the FIRST BEV conv consumes: the first block of
 * BaseBEVBackbone is ZeroPad2d(1) + Conv2d(k3, padding 0) (base_bev_backbone.py:31-36), i.e. a full copy of the 36 MB-per-frame
 * map just to add a border.  Here `dense` is (B, C, D, H + 2 pad_h, W + 2 pad_w) and the border is written, once, as zeros by the
 * same pass that writes the interior; the conv then runs on it with padding 0 and no pad kernel.  vc_from_dense_padded is the
 * backward gather from such a volume.  Workspace as vc_to_dense_fill (the row-id volume covers the UNPADDED grid).          */
int vc_to_dense_fill_padded(const float* features, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                            const int32_t* host_spatial_shape, int pad_h, int pad_w, float* dense, void* ws, size_t ws_bytes,
                            void* stream);
int vc_from_dense_padded(const float* dense, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                         const int32_t* host_spatial_shape, int pad_h, int pad_w, float* features, void* stream);

/* ------------------------------------------------------------------------------------------------ K1 voxelize + a3 MeanVFE
 * First-touch voxelisation of one frame's points with the mean (+ 'max' on the last channel) fused
 * (SURVEY App-A.9, A.13).  Replaces spconv.utils.Point2VoxelCPU3d.point_to_voxel (data_processor.py:35-41,53)
 * + MeanVFE.forward (mean_vfe.py:39-49).
 *   points (p, f) float32, host_range = [xmin,ymin,zmin,xmax,ymax,zmax], host_vsize = [vx,vy,vz]
 *   outputs sized max_voxels: features (max_voxels, f), coords (max_voxels, 3) int32 [z,y,x],
 *   num_points (max_voxels,) int32, *n_voxels_dev device int32 = number of voxels produced.                      */
size_t vc_voxelize_workspace_bytes(int64_t p, int max_points);
int vc_voxelize_mean(const float* points, int64_t p, int f, const float* host_range, const float* host_vsize,
                     int max_points, int max_voxels, int vfe_max_last, void* ws, size_t ws_bytes,
                     float* features, int32_t* coords, int32_t* num_points, int32_t* n_voxels_dev, void* stream);

/* Un-fused form with the reference's exact return protocol: voxels (max_voxels, max_points, f) zero-padded, slot j = the
 * j-th point (input order) of the cell; coords / num_points / *n_voxels_dev as above.  This is what
 * Point2VoxelCPU3d.point_to_voxel hands to VoxelGeneratorWrapper.generate (data_processor.py:53-58), which MeanVFE then
 * reduces (mean_vfe.py:39-49).  Same workspace as vc_voxelize_mean.                                                */
int vc_voxelize(const float* points, int64_t p, int f, const float* host_range, const float* host_vsize,
                int max_points, int max_voxels, void* ws, size_t ws_bytes, float* voxels, int32_t* coords,
                int32_t* num_points, int32_t* n_voxels_dev, void* stream);

/* ------------------------------------------------------------------------------------------------ a2 input point discard
 * StVD input discard of the virtual points, bin-based (pcdet/datasets/dataset.py:120-189 partition + input_point_discard):
 * bins of width max_dis/bin_num over x, far -> near; the running retain test on the bin counts picks `position`; the nearest
 * `position` bins holding more than per_bin = int((int(p*(1-rate)) - distant_acc) / (position + 1e-4)) points keep
 * perm_i[:per_bin] of them (in that order), the other bins keep everything in input order; output = bins far -> near.
 *   points      (p, f) float32, or float16 when points_are_f16 (the layout of the depth-completion .npy files,
 *               tools/PENet/vis_utils.py:148-152; kitti_dataset_mm.py:70-73 widens them with .astype(float32))
 *   rate, max_dis  as Python floats (double): the reference evaluates 1 - rate, N * retain and the ratio test in float64
 *   host_perms  NULL, or a HOST array of bin_num DEVICE pointers: entry i = the permutation of [0, count_i) the reference
 *               would draw for bin i with np.random.permutation (parity tests inject it); a NULL entry / NULL array draws a
 *               point-wise pseudo-random permutation from `seed` (Feistel network + cycle walking, as vc_random_keep)
 *   out         (p, f) float32 capacity; the first *n_out_dev rows are the result (device-side count, no host sync)   */
size_t vc_input_discard_workspace_bytes(int64_t p);
int vc_input_discard(const void* points, int points_are_f16, int64_t p, int f, int bin_num, double rate, double max_dis,
                     const int64_t* const* host_perms, uint64_t seed, void* ws, size_t ws_bytes, float* out,
                     int32_t* n_out_dev, void* stream);

/* Fused data front-end: raw LiDAR points + raw virtual points -> input discard (above) -> LiDAR-first
 * concatenation (dataset.py:270-294; data_processor.py:152-155 LIDAR_FIRST) -> optional `points[:, 3] /= intensity_div`
 * (dataset.py:292; 0 = off) -> first-touch voxeliser + MeanVFE (vc_voxelize_mean).  One call, no host synchronisation:
 * the kept-point count stays on the device (the voxeliser runs over the whole capacity, unused rows are out-of-range
 * sentinels placed after every real point).  *n_points_dev (nullable) = p_lidar + kept virtual points.               */
size_t vc_frontend_workspace_bytes(int64_t p_lidar, int64_t p_virtual, int f, int max_points);
int vc_frontend_voxelize_mean(const float* lidar, int64_t p_lidar, const void* virt, int virt_is_f16, int64_t p_virtual,
                              int f, int bin_num, double rate, double max_dis, const int64_t* const* host_perms,
                              uint64_t seed, float intensity_div, const float* host_range, const float* host_vsize,
                              int max_points, int max_voxels, int vfe_max_last, void* ws, size_t ws_bytes, float* features,
                              int32_t* coords, int32_t* num_points, int32_t* n_voxels_dev, int32_t* n_points_dev,
                              void* stream);

/* ------------------------------------------------------------------------------------------------ K11 BN(+ReLU)
 * Per-channel batch statistics over the N active rows and the fused normalise(+ReLU) pass -- the
 * nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU that follow every conv (spconv_backbone.py:101-105,160).
 *   vc_bn_stats      : sums (2, c) float64-accumulated -> mean (c), biased var (c)  (two-stage, deterministic);
 *                      optionally updates running_mean / running_var (unbiased var) in place with `momentum`
 *   vc_bn_apply_relu : y = relu?( (x - mean) * rsqrt(var + eps) * gamma + beta ), optionally written at a column
 *                      offset of a wider row (fuses the channel concat of NRConvBlock, spconv_backbone.py:227)
 *   vc_bn_relu_backward : dx, dgamma, dbeta for the training-mode BN(+ReLU)                                       */
size_t vc_bn_workspace_bytes(int64_t n, int c);
int vc_bn_stats(const float* x, int64_t n, int c, float* mean, float* var, float* running_mean /*nullable*/,
                float* running_var /*nullable*/, int64_t* num_batches_tracked /*nullable, += 1*/, float momentum,
                void* ws, size_t ws_bytes, void* stream);
/* statistics from the per-block partial sums of vc_conv_forward_epilogue(VC_EPI_STATS): same outputs as vc_bn_stats */
int vc_bn_stats_from_partial(const float* partial, int64_t nblocks, int64_t n, int c, float* mean, float* var,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                             void* ws /* nullable: vc_bn_workspace_bytes; enables the coalesced two-stage reduce */,
                             size_t ws_bytes, void* stream);
int vc_bn_apply_relu(const float* x, int64_t n, int c, const float* mean, const float* var, const float* gamma,
                     const float* beta, float eps, int relu, float* y, int y_stride, int y_col0, void* stream);
int vc_bn_relu_backward(const float* x, const float* dy, int dy_stride, int dy_col0, int64_t n, int c,
                        const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                        int relu, float* dx, float* dgamma, float* dbeta, unsigned* absmax_out /* nullable: receives
                        max|dx| as float bits (cleared by the call itself) */, void* ws, size_t ws_bytes, void* stream);

/* vc_bn_relu_backward with the two per-channel sums taken from the partial rows of vc_conv_backward_input_epilogue
 * (`partial`: nblocks rows of [2][c] floats) instead of a reduction pass over (x, dy).                                   */
int vc_bn_relu_backward_from_partial(const float* x, const float* dy, int dy_stride, int dy_col0, int64_t n, int c,
                                     const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                     int relu, const float* partial, int64_t nblocks, float* dx, float* dgamma, float* dbeta,
                                     unsigned* absmax_out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ post_act_block
 * conv (no bias) -> BatchNorm1d(training) -> ReLU, the unit every conv of the backbone is wrapped in (spconv_backbone.py:86-107
 * post_act_block, :110-131 post_act_block2d), forward and backward as ONE call each.  Host-side composition of the entry
 * points above (same launches, same order, bit-identical results); exists to take the per-launch Python cost (~16 us) off a
 * ~380-launch train step.
 *   forward : y_raw = conv(x);  batch statistics -> mean, var (+ running stats, num_batches_tracked);
 *             y[:, y_col0 : y_col0 + cout] = relu?((y_raw - mean) * rsqrt(var + eps) * gamma + beta)   (row stride y_stride:
 *             the channel concat of NRConvBlock is written in place)
 *   backward: dy (row stride dy_stride, column dy_col0) -> d_raw (BatchNorm + ReLU backward), dgamma, dbeta; dx = conv^T(d_raw)
 *             through tbl_dx (SubM: pair_fwd, mirror = 1; strided: pair_bwd, mirror = 0; duplicate-pixel 2-D convs: rep /
 *             centre + the table's group plan `grp_plan`, see vc_group_sum_sorted); dw = weight gradient. */
size_t vc_post_act_block_forward_workspace_bytes(int64_t n_in, int64_t n_out, int kv, int cin, int cout, int flags);
int vc_post_act_block_forward(const float* x, int64_t n_in, const int32_t* pair_fwd, int64_t n_out, int kv,
                              const float* weight, int cin, int cout, const int32_t* row_order, int operand_type, int flags,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float momentum, float eps, int relu, float* y_raw, float* y,
                              int y_stride, int y_col0, float* mean, float* var, void* ws, size_t ws_bytes, void* stream);
size_t vc_post_act_block_backward_workspace_bytes(int64_t n_out, int kv, int cin, int cout);
int vc_post_act_block_backward(const float* x, int64_t n_in, const float* y_raw, int64_t n_out, const float* dy,
                               int dy_stride, int dy_col0, const float* mean, const float* var, const float* gamma,
                               const float* beta, float eps, int relu, const int32_t* pair_fwd, const int32_t* tbl_dx,
                               int64_t n_dx, int mirror, int centre, const int32_t* rep, const int32_t* grp_plan,
                               const int32_t* row_order_dx, int kv,
                               const float* weight, int cin, int cout, int operand_type, int flags, int need_dx, int need_dw,
                               float* d_raw, float* dx, float* dw, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                               void* side_stream /* nullable hipStream_t: the weight gradient runs on it underneath the
                               backward-input conv; forked and joined inside the call */, void* stream);

/* ------------------------------------------------------------------------------------------------ feature pass
 * The feature pass of a whole backbone -- every post_act_block, the NRConvBlock channel concat (spconv_backbone.py:207-229)
 * and the layer discard gathers (:134-147) of VirConvL8x.forward (:609-699) -- as ONE call forward and ONE call backward
 * (the reverse-mode sweep that spconv's autograd Functions + torch's engine perform node by node, train_utils.py:47).
 * Pure host-side composition of the entry points above: same kernels, same order, bit-identical results.  What it removes
 * is the per-node host cost (Python autograd node + ctypes + allocator round trips, ~25 us per launch against ~4 us for the
 * launch itself), which bounds a ~370-launch train step on the host; all intermediate tensors live in ONE caller-provided
 * arena per direction (bump-allocated, 256-B aligned; layout is a pure function of the program and the row counts).
 *
 * A program is a list of ops over numbered row-major float32 buffers (rows x cols):
 *   VC_PASS_UNIT   dst[:, dst_col0 : dst_col0 + cout] = post_act_block(unit, table)(src)     (src dense: cols == cin)
 *   VC_PASS_COPY   dst[:, dst_col0 : dst_col0 + cols(src)] = src                               (channel concat, written in place)
 *   VC_PASS_GATHER dst = src[keeps[keep], :]                                                   (layer discard)
 * Buffer 0.. may be `external` (caller memory, e.g. the input features); all others are carved from the arena and their
 * byte offsets are reported so the caller can hand some of them out (x_conv1..4, the encoded tensor).
 * Backward: gradients arriving from outside per buffer (`ext_grads[b]`, dense rows x cols, or NULL) are propagated in
 * reverse op order; a buffer's gradient is the sum of its consumers' contributions (column slices are consumed as strided
 * views, two contributions are added by one kernel); parameter gradients are written to the units' dweight/dgamma/dbeta
 * (zero-filled when no gradient reaches a unit).  Weight gradients run on `side_stream` underneath the backward-input
 * chain and are joined once at the end.                                                                               */
typedef enum vc_pass_kind { VC_PASS_UNIT = 1, VC_PASS_COPY = 2, VC_PASS_GATHER = 3 } vc_pass_kind;
typedef struct vc_pass_unit {    /* one post_act_block: conv weight (Cout, *k, Cin), BatchNorm1d parameters/state, gradient outputs */
  const float* weight; const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* dweight; float* dgamma; float* dbeta;           /* vc_pass_backward outputs (NULL: not wanted) */
  int32_t cin, cout; float momentum, eps;
} vc_pass_unit;
typedef struct vc_pass_table {   /* one rulebook (vc_subm_rulebook / vc_spconv_emit_pairs outputs + hints) */
  const int32_t* pair_fwd; const int32_t* pair_bwd /* strided only */; const int32_t* rep /* duplicate-pixel rule or NULL */;
  const int32_t* order_fwd; const int32_t* order_bwd;     /* vc_row_order hints or NULL */
  int64_t n_in, n_out; int32_t kv, subm, centre, sorted_rows;
  const int32_t* grp_plan;                                /* with rep: [order | sorted keys] of vc_group_sum_sorted */
} vc_pass_table;
typedef struct vc_pass_buf { int64_t rows; int32_t cols; int32_t external; void* ptr /* external only */; } vc_pass_buf;
typedef struct vc_pass_op { int32_t kind, src, dst, dst_col0, unit, table, keep, relu; } vc_pass_op;
typedef struct vc_pass_program {
  const vc_pass_op* ops; int32_t n_ops;
  const vc_pass_buf* bufs; int32_t n_bufs;
  const vc_pass_unit* units; int32_t n_units;
  const vc_pass_table* tables; int32_t n_tables;
  const int64_t* const* keeps; int32_t n_keeps;
  int32_t training;        /* 1: batch statistics (+ running-stat update), backward available; 0: running statistics */
  int32_t operand_type;    /* vc_operand */
  int32_t pack_all;        /* 1: every unit's conv reads a fragment-ordered weight image (what the split-product kernels want), 0: only
                              convs of >= 60 000 rows.  The CALLER's snapshot, taken once per forward call, of the library switch the
                              arena layouts depend on: the forward and the backward layout of one call must agree even if the
                              switch (vc_debug_set f32_split) moves in between */
  int32_t reserved_;
} vc_pass_program;
size_t vc_pass_forward_arena_bytes(const vc_pass_program* prog);
int vc_pass_forward(const vc_pass_program* prog, void* arena, size_t arena_bytes,
                    int64_t* buf_offsets /* n_bufs: byte offset of every arena buffer, -1 for external ones */, void* stream);
size_t vc_pass_backward_arena_bytes(const vc_pass_program* prog, const float* const* ext_grads, int need_input_grad);
int vc_pass_backward(const vc_pass_program* prog, const void* fwd_arena, size_t fwd_arena_bytes,
                     const float* const* ext_grads /* n_bufs */, float* input_grad /* gradient of buffer 0 or NULL */,
                     void* arena, size_t arena_bytes, void* side_stream /* nullable */, void* stream);

/* ------------------------------------------------------------------------------------------------ kernel timing
 * Brackets every launch of ONE conv kernel instantiation (direction: 0 forward, 1 backward-input gather-GEMM with ck, cn = its
 * gathered / produced channel counts; 2 the weight gradient with ck = cin, cn = cout) with HIP events on the launch stream, inside whatever call issues it (vc_conv_*,
 * vc_post_act_block_*, vc_pass_*), and counts the table's active pairs on the device right after it (outside the
 * bracket; the first 16 records of a one-kernel trace (128 with direction -1) count exactly, later ones take the pairs-per-row ratio of the last counted launch of the
 * same shape -- the count is measurement work inside the timed step).  bench.py's roofline figure comes from here.
 * `dev_pairs`: device int64[max_records], caller-owned.
 * direction = -1 records EVERY gather-GEMM launch (record.direction 0 / 1) and every weight-gradient launch (2: kernel + its
 * reduce; ck = cin, cn = cout): the family- and step-level roofline figures of bench.py.
 * direction = 3 records the launches of the BatchNorm-backward dx kernel (the largest bandwidth-bound kernel of a train step by time):
 * ck = channel filter (0 = every channel count), cn ignored; record = {ck = cn = channels, n_src = n_out = rows, pairs = 0}: the HBM side
 * of bench.py's roofline (algorithmic bytes 3 x 4 x rows x channels).                                                      */
typedef struct vc_trace_record { float ms; int32_t kv, ck, cn, windowed; int64_t n_src, n_out, pairs; int32_t direction;
                                 float t0_ms /* start of the launch, since the first traced launch */; } vc_trace_record;
int vc_trace_begin(int direction, int ck, int cn, int max_records, int64_t* dev_pairs);
int vc_trace_end(vc_trace_record* out, int capacity, int* n_records /* synchronises the traced events */);

/* ================================================================================================ RoI grid pooling
 * the operators that consume multi_scale_3d_features['x_conv3'/'x_conv4'] right after the backbone
 * (pcdet/models/roi_heads/ted_head.py:450-650 -> pointnet2_stack/voxel_pool_modules.py:70-130).
 *
 * Voxel index: occupancy bitmap + coordinate hash over the (N, 4) [b, z, y, x] indices of a sparse tensor.  Replaces the
 * dense (B, Z, Y, X) int32 volume of generate_voxel2pinds (pcdet/utils/spconv_utils.py:4-21).                        */
size_t vc_voxel_index_workspace_bytes(int64_t n, int batch_size, const int32_t* host_spatial_shape /* [Z, Y, X] */);
int vc_voxel_index_build(const int32_t* indices, int64_t n, int batch_size, const int32_t* host_spatial_shape, void* ws,
                         size_t ws_bytes, void* stream);

/* Voxel query: for query m (metric position new_xyz[m], voxel coordinate new_coords[m] = [b, z, y, x]) the first `nsample`
 * voxels, in dz, dy, dx ascending scan order over [-range, +range]^3, whose centre xyz[row] lies within `radius`.
 * idx (m, nsample) int32 = rows of `xyz` (GLOBAL rows, as the reference kernel writes them); unused slots repeat the
 * first hit; empty_mask[m] = 1 and a zero row when nothing was found (the post-processing of VoxelQuery.forward is
 * included).  x_range <= 31.  Replaces voxel_query_wrapper_stack (pointnet2_stack/src/voxel_query.cpp:25-41,
 * voxel_query_gpu.cu:10-113, voxel_query_utils.py:12-46).                                                            */
int vc_voxel_query(const void* ws, size_t ws_bytes, int64_t n, int batch_size, const int32_t* host_spatial_shape,
                   const float* xyz, const float* new_xyz, const int32_t* new_coords, int64_t m, int z_range, int y_range,
                   int x_range, float radius, int nsample, int32_t* idx, uint8_t* empty_mask, void* stream);

/* out[m, c, s] = features[start(batch of m) + idx[m, s], c]  (idx batch-LOCAL, as the reference's stacked layout) and its
 * transpose (scatter-add with fp32 atomics, like the reference).  Argument order follows group_points_wrapper_stack /
 * group_points_grad_wrapper_stack (pointnet2_stack/src/group_points.cpp, group_points_gpu.cu:15-118,
 * pointnet2_utils.py:48-105).  grad_features is zeroed by the call.                                                   */
int vc_group_points(int batch_size, int64_t m, int c, int nsample, const float* features,
                    const int32_t* features_batch_cnt, const int32_t* idx, const int32_t* idx_batch_cnt, float* out,
                    void* stream);
int vc_group_points_grad(int batch_size, int64_t m, int c, int64_t n, int nsample, const float* grad_out,
                         const int32_t* idx, const int32_t* idx_batch_cnt, const int32_t* features_batch_cnt,
                         float* grad_features, void* stream);

/* ---- rotated-box BEV overlap / IoU and NMS (replaces pcdet/ops/iou3d_nms) --------------------------------
 * boxes: (n, 7) float32 [x, y, z, dx, dy, dz, heading], contiguous.
 *   vc_boxes_overlap_bev  -> (n_a, n_b) overlap areas   = iou3d_nms_cuda.boxes_overlap_bev_gpu (iou3d_nms.cpp:47-69, kernel .cu:235-246)
 *   vc_boxes_iou_bev      -> (n_a, n_b) BEV IoU         = iou3d_nms_cuda.boxes_iou_bev_gpu     (iou3d_nms.cpp:71-96, kernel .cu:248-261)
 *   vc_boxes_iou3d        -> (n_a, n_b) 3-D IoU         = iou3d_nms_utils.boxes_iou3d_gpu (iou3d_nms_utils.py:67-99) as ONE launch
 *   vc_nms                : boxes ALREADY sorted by descending score; keep (capacity n) receives the selected positions in
 *                           ascending order, *num_out (device) their count -- iou3d_nms_cuda.nms_gpu (rotated = 1,
 *                           iou3d_nms.cpp:98-150) / nms_normal_gpu (rotated = 0, :153-187) with the selection loop on the
 *                           device instead of on the host.  n <= 65536.  ws: vc_nms_workspace_bytes(n) bytes.              */
int vc_boxes_overlap_bev(const float* boxes_a, int64_t n_a, const float* boxes_b, int64_t n_b, float* overlap, void* stream);
int vc_boxes_iou_bev(const float* boxes_a, int64_t n_a, const float* boxes_b, int64_t n_b, float* iou, void* stream);
int vc_boxes_iou3d(const float* boxes_a, int64_t n_a, const float* boxes_b, int64_t n_b, float* iou, void* stream);
size_t vc_nms_workspace_bytes(int64_t n);
int vc_nms(const float* boxes, int64_t n, float thresh, int rotated, int64_t* keep, int64_t* num_out, void* ws, size_t ws_bytes,
           void* stream);

/* ------------------------------------------------------------------------------------------------ f3: first BEV conv on the sparse rows
 * SURVEY 8f rank 3, "fused": ZeroPad2d(1) + Conv2d(C*D -> 64, k3) of BaseBEVBackbone's first block (base_bev_backbone.py:31-38) over
 * the map HeightCompression builds (height_compression.py:27-31) is a sparse conv from the encoded tensor's (b, z, y, x) rows onto
 * the (b, y, x) cells with kernel (D, ky, kx): offset (z, a, c) of cell (y, x) reads the voxel at height z of (y + a - ky/2,
 * x + c - kx/2).  vc_bev_pairs writes that pair table, (D * ky * kx, B * H * W) int32, for EVERY cell in dense (b, y, x) order
 * (ws: B * D * H * W int32 row-id volume): no compaction and no count read; vc_conv_forward / vc_conv_backward_input over it
 * produce the NHWC map directly (cells that see no voxel cost a table read).  vc_nhwc_to_nchw turns (B * HW, C) rows into the
 * (B, C, HW) map, optionally with y = x * scale[c] + shift[c] (the BatchNorm2d behind the conv, :37) and a ReLU (:38).     */
size_t vc_bev_pairs_workspace_bytes(int batch_size, const int32_t* host_spatial_shape /* D, H, W */);
int vc_bev_pairs(const int32_t* indices, int64_t n, int batch_size, const int32_t* host_spatial_shape, int ky, int kx, int32_t* pair,
                 void* ws, size_t ws_bytes, void* stream);
/* The transposed pair table of vc_bev_pairs for the stem's backward pass: pair_bwd (D * ky * kx, n) int32, pair_bwd[k][i] = the BEV
 * cell (b * H + y') * W + x' that voxel row i = (b, z, y, x) feeds through offset k = (kz, a, c) -- kz must be the row's own z -- or -1
 * (training through base_bev_backbone.py:31-38: dX = a forward-form gather-GEMM of dY over this table, dW = the weight-gradient
 * kernel over vc_bev_pairs' table).                                                                                          */
int vc_bev_pairs_backward(const int32_t* indices, int64_t n, int batch_size, const int32_t* shape /* D, H, W */, int ky, int kx,
                          int32_t* pair_bwd, void* stream);
int vc_nhwc_to_nchw(const float* x, int batch_size, int64_t hw, int c, const float* scale /* nullable */, const float* shift,
                    int relu, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------ geometry plan
 * Every index structure of a chain of NRConvBlocks (spconv_backbone.py:150-229) -- per block: the strided-conv rulebook of its
 * down_layer (:164-171) with the block's output coordinates, the 3-D SubM rulebook shared by d3_conv1/2 (:186,199), the pixel
 * coordinates of index2uv (:54-83, :214-216), the 2-D SubM rulebook shared by d2_conv1/2 over the duplicate pixel coordinates
 * (:217-222) with its representatives and group plan, the rows layer_voxel_discard keeps (:134-147) -- plus the strided conv
 * that follows the chain (conv_out, :561-567), built by TWO calls around ONE host read.  In the reference this is what spconv's
 * indice generation does conv by conv inside VirConvL8x.forward (:609-699), with one device-to-host sync per strided conv.
 *
 *   vc_plan_begin   enqueues, with every row count still on the device: for each level the output-cell bitmap, its scan, the
 *                   coordinate emission (row CAPACITY buffers) and the layer discard's keep / kept coordinates; then the counts'
 *                   copy to `host_counts` (pinned host memory, >= 64 int32) and an event; then the 3-D SubM table of a first
 *                   block that has no strided conv (its row count is the caller's).
 *   vc_plan_wait    polls that event; checks injected keeps against the counts.  The ONE host synchronisation of the plan.
 *   vc_plan_finish  enqueues every remaining table a FORWARD pass reads, with exact row counts, into `arena_b` (all views of
 *                   `vc_plan_out` are final; those named next are filled by the call after it): first the integer tables, then --
 *                   behind `tables_wait_event` -- the image-space branch of ALL blocks in three launches (one clear of the pixel
 *                   images, projection + pixel marking, pair tables + representatives).
 *   vc_plan_finish_backward  enqueues what only backward passes read -- the duplicate-pixel group plans (`grp_plan`, the sorts:
 *                   a third of the plan's stream time) and the backward row orders (`order_bwd`).  A caller records an event
 *                   between the two calls and lets the forward pass wait for that one only.  No-op when need_grad == 0.
 * Host-side composition of the operators above plus index kernels that exploit what the chain knows (plan.hip): the coordinates
 * of a strided conv's output are the set bits of its bitmap in ascending order, so the SubM rulebook of that tensor ranks
 * neighbours in the SAME bitmap (no hash build); the pixel tensors index a dense per-sample image (no hash build, 6x fewer
 * bytes to clear); the backward row order of a strided table is a counting sort on the stride-parity class of the input
 * coordinate (no pass over the 27-row table).  Tables are bit-identical
 * to vc_subm_rulebook / vc_spconv_emit_pairs / vc_group_plan on the same coordinates (tests); row orders are hints.
 * The caller owns both arenas; `vc_plan_out` reports every structure as (arena, byte offset, rows, cols).                  */
#define VC_PLAN_MAX_BLOCKS 8
typedef struct vc_plan_conv { int32_t ksize[3], stride[3], padding[3], dilation[3]; } vc_plan_conv;
typedef struct vc_plan_block {
  int32_t has_down;                         /* a strided conv (down_layer) in front of the block */
  vc_plan_conv down;
  int32_t subm_ksize[3], subm_dilation[3];  /* the 3-D SubM rulebook shared by d3_conv1 / d3_conv2 */
  int32_t has_2d;                           /* image-space branch: index2uv + the shared 2-D SubM rulebook */
  int32_t uv_stride;                        /* index2uv stride of the block: 1 / 2 / 4 / 8 */
  int32_t ksize2d[2], dilation2d[2];
  int32_t discard;                          /* layer_voxel_discard after the block */
  uint64_t keep_seed;                       /* discard, keep == NULL: rows = vc_random_keep(n, int(n * (1 - rate)), keep_seed) */
  const int64_t* keep;                      /* injected kept rows (device int64) or NULL */
  int64_t keep_rows;
} vc_plan_block;
typedef struct vc_plan_desc {
  const int32_t* indices; int64_t n;        /* (n, 4) int32 [b, z, y, x] */
  int32_t batch_size; int32_t spatial_shape[3];
  const float* calib; const float* trans;   /* (B, 33) | (B, 3) or NULL, as vc_project_prepare */
  int32_t image_shape[2];                   /* spatial shape of the 2-D tensors ([1600, 600], spconv_backbone.py:220) */
  int32_t input_discard; uint64_t input_keep_seed; const int64_t* input_keep; int64_t input_keep_rows;  /* discard of the chain's input (VirConv8x MM stream, :488-489) */
  int32_t n_blocks; vc_plan_block blocks[VC_PLAN_MAX_BLOCKS];
  int32_t has_tail; vc_plan_conv tail;
  double discard_rate;
  int32_t need_grad;                        /* also build what only a backward pass needs: group plans, backward row orders */
  int32_t row_order_fwd;                    /* 1: also a row order for the strided convs' FORWARD tables */
  int32_t defer_early_tables;               /* 1: vc_plan_begin builds NO table (also not those of a first block whose row count is the
                                               caller's): vc_plan_finish builds them all.  For callers that begin several plans -- or one
                                               plan a step early -- before finishing any */
  int32_t allow_unfenced_projection;        /* 0 (default): vc_plan_finish REFUSES (VC_EINVAL) a plan with an image-space branch and no
                                               tables_wait_event.  1: the caller asserts that no conv kernel of this library runs on the
                                               device while the branch does (single stream, an idle device, a diagnostics run) */
  void* tables_wait_event;                  /* hipEvent_t or NULL: vc_plan_finish lets the stream wait for it between the integer tables and
                                               the image-space branch (pixel projection -- the plan's only floating-point kernel -- and
                                               the pixel tables of all blocks).  The caller records it behind the previous step's feature
                                               passes: the projection never runs beside conv kernels (LOG.md A.15 / A.17).  May be set
                                               between vc_plan_begin and vc_plan_finish */
  void* debug_buf; int64_t debug_bytes;     /* developer diagnostics (tools/det_check.py) or NULL: zeroed int32 buffer: 64-int header ([3] = rows per stage of the
                                               intermediates area, 0 = none), 4096 32-int records, then 4 x [3] x 8 floats; a log of
                                               every projection thread that read "no augmentation" from a plan that has one */
} vc_plan_desc;
typedef struct vc_plan_view { int32_t arena /* 0: arena_a, 1: arena_b, -1: absent */; int32_t cols; int64_t offset /* bytes */; int64_t rows; } vc_plan_view;
typedef struct vc_plan_table_out {
  vc_plan_view pair_fwd, pair_bwd, rep, order_fwd, order_bwd, grp_plan, in_indices, out_indices;
  int64_t n_in, n_out; int32_t kv, present; int32_t out_shape[3], pad_;
} vc_plan_table_out;
typedef struct vc_plan_block_out {
  vc_plan_table_out down, subm3d, subm2d;
  vc_plan_view uv, keep, kept_indices;
  int64_t n, n_keep;
} vc_plan_block_out;
typedef struct vc_plan_out {
  vc_plan_view input_keep, input_kept_indices; int64_t n_input_kept;
  vc_plan_block_out blocks[VC_PLAN_MAX_BLOCKS];
  vc_plan_table_out tail;
} vc_plan_out;
typedef struct vc_plan_state { int64_t opaque[2048]; } vc_plan_state;   /* caller-held, written by vc_plan_begin */
size_t vc_plan_begin_arena_bytes(const vc_plan_desc* desc);
int vc_plan_begin(const vc_plan_desc* desc, void* arena_a, size_t arena_a_bytes, int32_t* host_counts, vc_plan_state* state,
                  void* stream);
int vc_plan_wait(const vc_plan_desc* desc, vc_plan_state* state);
size_t vc_plan_finish_arena_bytes(const vc_plan_desc* desc, const vc_plan_state* state);
int vc_plan_finish(const vc_plan_desc* desc, vc_plan_state* state, void* arena_a, void* arena_b, size_t arena_b_bytes,
                   vc_plan_out* out, void* stream);
int vc_plan_finish_backward(const vc_plan_desc* desc, vc_plan_state* state, void* arena_a, void* arena_b, size_t arena_b_bytes,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIRCONV_HIP_H_ */
