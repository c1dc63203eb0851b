// Duplicate-pixel group sum of the 2-D image-space SubM backward, round 3: a SEGMENTED sum over rows sorted by representative.
//
// Reference semantics (pcdet/models/backbones_3d/spconv_backbone.py:110-131,217-222): the image-space tensors
// repeat pixel coordinates; a neighbour tap reads the representative row of the neighbouring pixel, so in the backward every
// representative receives, per non-centre tap, the SUM of the gradients of all rows of the consuming pixel group:
//     dy_grp[rep, :] = sum over rows i with rep[i] == rep of dy[i, :]
// Rounds 1-2 carried this sum in 64-bit fixed point with one int64 atomic per run (bit-stable, but it needed max|dy| first, an
// 8-byte accumulator per element and a convert pass: 0.375 ms per train step).  Here the order of the additions is FIXED by
// the data instead: the geometry plan sorts the rows once per table by representative (stable: ties in ascending row order;
// `grp_plan` = [order (n)][sorted keys (n)], int32), and
//   * seg_sum_kernel     thread = (chunk of 32 consecutive sorted positions, channel): walks its rows in order, sums runs of equal
//                        keys in a register (plain fp32 adds in ascending sorted position), stores a run that begins and ends
//                        inside the chunk straight to dy_grp[key] and a run cut by a chunk border as a partial
//                        (slot 0: the run came in from the previous chunk, slot 1: it began here and continues);
//   * seg_fixup_kernel   block = a chunk whose last run began there and continues: binary-searches the end of the run and adds the
//                        continuation partials of the following chunks in a fixed interleaved order (256 / c lanes per channel).
// No atomics, no absmax pass, no accumulator buffer, no convert pass; every group's additions happen in one fixed order, so
// the result is bit-stable run to run.  HBM-bound: one gathered read of dy (rows are 32-256 B contiguous) + 8 B of plan per row.
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>

#include "common.h"

namespace vc {

static constexpr int kSegRows = 32;

__global__ void __launch_bounds__(256) seg_sum_kernel(const float* __restrict__ dy, const int32_t* __restrict__ order,
                                                      const int32_t* __restrict__ keys, int64_t n, int c, int lg_c,
                                                      float* __restrict__ grp, float* __restrict__ part) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t chunk = t >> lg_c;
  const int ch = (int)(t & (c - 1));
  const int64_t j0 = chunk * kSegRows;
  if (j0 >= n) return;
  const int64_t j1 = min(j0 + (int64_t)kSegRows, n);
  const int prevkey = j0 > 0 ? keys[j0 - 1] : -1;
  const int nextkey = j1 < n ? keys[j1] : -1;
  int cur = keys[j0];
  float acc = dy[(int64_t)order[j0] * c + ch];
  bool first_run = true;
  auto flush = [&](bool open) {
    const bool head = first_run && cur == prevkey;
    if (head) part[(chunk * 2 + 0) * c + ch] = acc;        // continuation of a run that began in an earlier chunk
    else if (open) part[(chunk * 2 + 1) * c + ch] = acc;   // began here, continues in the next chunk
    else grp[(int64_t)cur * c + ch] = acc;                 // complete
  };
  int64_t j = j0 + 1;
  for (; j + 8 <= j1; j += 8) {   // two dependent loads per row (order -> dy): 8 rows in flight
    int k[8], r[8];
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { k[u] = keys[j + u]; r[u] = order[j + u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = dy[(int64_t)r[u] * c + ch];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k[u] != cur) { flush(false); cur = k[u]; acc = v[u]; first_run = false; }
      else acc += v[u];
    }
  }
  for (; j < j1; ++j) {
    const int k = keys[j];
    const float v = dy[(int64_t)order[j] * c + ch];
    if (k != cur) { flush(false); cur = k; acc = v; first_run = false; }
    else acc += v;
  }
  flush(cur == nextkey);
}

// One 256-thread block per chunk; only a chunk whose last run BEGAN there and continues does anything.  The continuation partials
// (one per following chunk; hundreds for the border-pixel groups that collect thousands of rows) are split over 256 / c
// segment lanes per channel -- lane s adds partials s, s + S, ... in that order -- and the S lane sums are combined in lane
// order: a fixed tree for a given run length, hence bit-stable.  (The first version gave a whole run to ONE thread per channel:
// 23 us per launch, almost all of it the 600-partial chain of the biggest group.)
__global__ void __launch_bounds__(256) seg_fixup_kernel(const int32_t* __restrict__ keys, int64_t n, int c, int lg_c,
                                                        const float* __restrict__ part, float* __restrict__ grp) {
  __shared__ float red[256];
  const int64_t chunk = blockIdx.x;
  const int64_t j0 = chunk * kSegRows, j1 = j0 + kSegRows;
  if (j1 >= n) return;                        // the last chunk cannot have an open run
  // everything the decision needs in ONE round of loads (they were three dependent rounds and, for a head, a binary search of ~18 more:
  // with 2-6 rows per pixel most chunk borders cut a run, so most blocks are heads -- and most runs end inside the next chunk)
  const int64_t jp = min(j1 + kSegRows - 1, n - 1);   // last position of the next chunk
  const int key = keys[j1 - 1], k_next = keys[j1], k_first = keys[j0], k_before = j0 > 0 ? keys[j0 - 1] : 0, k_probe = keys[jp];
  if (k_next != key) return;                  // last run ends with the chunk
  if (k_first == key && j0 > 0 && k_before == key) return;   // the whole chunk is a continuation: not the run's first chunk
  // end of the run: first position e in (j1, n] with keys[e] != key (keys ascending); every thread searches for itself (uniform)
  int64_t c_end = chunk + 1;                  // k_probe != key: the run ends inside the next chunk
  if (k_probe == key) {
    int64_t lo = jp, hi = n;                  // invariant: keys[lo] == key, (hi == n or keys[hi] != key)
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] == key) lo = mid; else hi = mid;
    }
    c_end = (hi - 1) / kSegRows;              // last chunk holding rows of the run
  }
  const int ch = threadIdx.x & (c - 1), seg = threadIdx.x >> lg_c, S = 256 >> lg_c;
  float acc = 0.f;
  int64_t q = chunk + 1 + seg;
  for (; q + 3 * (int64_t)S <= c_end; q += 4 * (int64_t)S) {
    const float v0 = part[(q * 2) * c + ch], v1 = part[((q + S) * 2) * c + ch];
    const float v2 = part[((q + 2 * S) * 2) * c + ch], v3 = part[((q + 3 * S) * 2) * c + ch];
    acc += v0; acc += v1; acc += v2; acc += v3;
  }
  for (; q <= c_end; q += S) acc += part[(q * 2) * c + ch];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (seg == 0) {
    float total = part[(chunk * 2 + 1) * c + ch];   // the run's own first piece, then the segment sums in lane order
    for (int u = 0; u < S; ++u) total += red[(u << lg_c) + ch];
    grp[(int64_t)key * c + ch] = total;
  }
}
// (Round 4 tried blocks that scan chunk ranges for such heads -- <= 1024 blocks instead of ~10 000 that return at once, short runs
// summed by c threads each, a parallel probe for the end of long runs; bit-identical, and 15.3 us per call in-step (rocprof, r4F2)
// against 12.2 us for this form (r4p): the dispatch of empty blocks is cheaper than the scan's dependent loads.  LOG.md A.12.)

// keys[i] = rep[i] < 0 ? i : rep[i]  (what the plan sorts; rep = -1 marks "own representative" in some tables); rows[i] = i
__global__ void __launch_bounds__(256) group_keys_kernel(const int32_t* __restrict__ rep, int64_t n, uint32_t* __restrict__ keys,
                                                         int32_t* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int g = rep[i];
  keys[i] = g < 0 ? (uint32_t)i : (uint32_t)g;
  if (rows != nullptr) rows[i] = (int32_t)i;
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
int g_group_plan_radix = 0;   // vc_debug_set plan_radix_sort (experiment builds): 1 = the hand-written LDS radix sort of csrc/experiments/

#ifdef VC_EXPERIMENTS
#include "experiments/group_plan_radix.inc"
#endif

static inline int key_bits(int64_t n) {
  int b = 1;
  while (b < 32 && (1LL << b) < n) ++b;
  return b;
}


// --------------------------------------------------------------------------------------------- weighted sum (fused mul + sum)
// out = sum_b sum_i x[b, i] * g[i]: the contraction the benchmark's stand-in loss is made of (bench.py synthetic_loss: the dense
// BEV map against a fixed weight map, every scale's features against a per-channel weight), as ONE pass over x instead of an
// elementwise product (a full-size temporary) plus a reduction -- torch's `(x * g).sum()` cost 19 + 47 us on the BEV map and
// 19 + 5 us per scale.  Deterministic: the partition of x over blocks and threads is fixed, every thread adds its elements in
// ascending order in fp32, a block adds its threads in a fixed tree in fp64, and the second kernel adds the block partials in a
// fixed order in fp64.  HBM-bound: 4 bytes of x per element (+ g from cache).
static constexpr int kWsThreads = 256, kWsUnroll = 8, kWsMaxBlocks = 2048;

// MODE 0: x is (nb, e) with a large e (grid.y = sample, grid.x over e): thread reads x and g at the same offset
// MODE 1: x is (nb, e) rows with a small power-of-two e <= 1024 (1024 % e == 0): flat walk, a thread's four g values are fixed
template <int MODE>
__global__ void __launch_bounds__(kWsThreads) weighted_sum_kernel(const float* __restrict__ x, int64_t nb, int64_t e,
                                                                 const float* __restrict__ g, double* __restrict__ partial) {
  __shared__ double red[kWsThreads / 64];
  float acc = 0.f;
  if (MODE == 0) {
    const int64_t e4 = e >> 2;
    const float4* xb = reinterpret_cast<const float4*>(x + (int64_t)blockIdx.y * e);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t j = (int64_t)blockIdx.x * kWsThreads + threadIdx.x; j < e4; j += (int64_t)gridDim.x * kWsThreads * kWsUnroll) {
      float4 xv[kWsUnroll], gv[kWsUnroll];
#pragma unroll
      for (int u = 0; u < kWsUnroll; ++u) {
        const int64_t jj = j + (int64_t)u * gridDim.x * kWsThreads;
        const bool in = jj < e4;
        xv[u] = in ? xb[jj] : float4{0.f, 0.f, 0.f, 0.f};
        gv[u] = in ? g4[jj] : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < kWsUnroll; ++u) {
        acc += xv[u].x * gv[u].x; acc += xv[u].y * gv[u].y; acc += xv[u].z * gv[u].z; acc += xv[u].w * gv[u].w;
      }
    }
  } else {
    const int64_t n4 = (nb * e) >> 2;
    const float4 gv = *reinterpret_cast<const float4*>(g + ((threadIdx.x * 4) & (e - 1)));   // (4 * block stride) % e == 0
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t j = (int64_t)blockIdx.x * kWsThreads + threadIdx.x; j < n4; j += (int64_t)gridDim.x * kWsThreads * kWsUnroll) {
      float4 xv[kWsUnroll];
#pragma unroll
      for (int u = 0; u < kWsUnroll; ++u) {
        const int64_t jj = j + (int64_t)u * gridDim.x * kWsThreads;
        xv[u] = jj < n4 ? x4[jj] : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < kWsUnroll; ++u) {
        acc += xv[u].x * gv.x; acc += xv[u].y * gv.y; acc += xv[u].z * gv.z; acc += xv[u].w * gv.w;
      }
    }
  }
  double d = (double)acc;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = red[0];
    for (int w_ = 1; w_ < kWsThreads / 64; ++w_) t += red[w_];
    partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(256) weighted_sum_final_kernel(const double* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ double red[4];
  double t = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) t += partial[j];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + red[2]) + red[3]);
}

// dx[b, i] = gout[0] * g[i], b < nb_out
__global__ void __launch_bounds__(256) weighted_sum_backward_kernel(const float* __restrict__ gout, const float* __restrict__ g,
                                                                    int64_t nb_out, int64_t e, float* __restrict__ dx) {
  const float s = gout[0];
  const int64_t e4 = e >> 2, total = nb_out * e4;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total; j += (int64_t)gridDim.x * 256) {
    const float4 gv = reinterpret_cast<const float4*>(g)[j % e4];
    reinterpret_cast<float4*>(dx)[j] = float4{s * gv.x, s * gv.y, s * gv.z, s * gv.w};
  }
}

__global__ void __launch_bounds__(256) dbg_slow_fill_kernel(int32_t* __restrict__ buf, int64_t n, int spin) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i;
  for (int s_ = 0; s_ < spin; ++s_) x = x * 1664525u + 1013904223u;   // a dependent chain the compiler cannot fold
  buf[i] = 1 + (int)(x == 0xdeadbeefu && spin < 0);
}
__global__ void __launch_bounds__(256) dbg_copy_kernel(const int32_t* __restrict__ a, int32_t* __restrict__ b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline void weighted_sum_grid(int64_t nb, int64_t e, int& mode, dim3& grid) {
  if (e <= 1024 && (e & (e - 1)) == 0) {
    mode = 1;
    int64_t want = cdiv((nb * e) >> 2, (int64_t)kWsThreads * kWsUnroll);
    if (want < 1) want = 1;
    if (want > kWsMaxBlocks) want = kWsMaxBlocks;
    grid = dim3((unsigned)want, 1);
  } else {
    mode = 0;
    int64_t per = kWsMaxBlocks / (nb < 1 ? 1 : nb);
    if (per < 1) per = 1;
    int64_t want = cdiv(e >> 2, (int64_t)kWsThreads * kWsUnroll);
    if (want < 1) want = 1;
    if (want > per) want = per;
    grid = dim3((unsigned)want, (unsigned)nb);
  }
}

// rocPRIM's radix_sort_pairs picks a merge sort below 1 M items; OnesweepCfg (merge-sort limit 0) forces its Onesweep radix
// passes instead (vc_debug_set group_plan_onesweep, A/B)
using OnesweepCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
int g_group_plan_onesweep = 0;
template <class Cfg>
static size_t rocprim_temp_bytes(int64_t n) {
  size_t temp = 0;
  if (rocprim::radix_sort_pairs<Cfg>(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                     (unsigned)n, 0u, (unsigned)key_bits(n)) != hipSuccess)
    return 0;
  return temp;
}

// ---- all tables of a plan by one sort (common.h GroupMulti)
__global__ void __launch_bounds__(256) group_keys_multi_kernel(GroupMulti g, uint32_t* __restrict__ keys, int32_t* __restrict__ rows) {
  int s = 0;
#pragma unroll
  for (int t = 1; t < 8; ++t)
    if (t < g.n_tables && blockIdx.x >= g.block0[t]) s = t;
  const int64_t i = (int64_t)(blockIdx.x - g.block0[s]) * 256 + threadIdx.x;
  if (i >= g.n[s]) return;
  const int r = g.rep[s][i];
  keys[g.off[s] + i] = ((uint32_t)s << g.key_bits) | (r < 0 ? (uint32_t)i : (uint32_t)r);
  rows[g.off[s] + i] = (int32_t)i;
}

__global__ void __launch_bounds__(256) group_split_multi_kernel(GroupMulti g, const uint32_t* __restrict__ keys,
                                                                const int32_t* __restrict__ rows) {
  int s = 0;
#pragma unroll
  for (int t = 1; t < 8; ++t)
    if (t < g.n_tables && blockIdx.x >= g.block0[t]) s = t;
  const int64_t i = (int64_t)(blockIdx.x - g.block0[s]) * 256 + threadIdx.x;
  if (i >= g.n[s]) return;
  // table s holds exactly n[s] keys with tag s, and the tags sort first: its rows are positions [off[s], off[s] + n[s])
  int32_t* o = g.plan[s];
  o[i] = rows[g.off[s] + i];
  o[g.n[s] + i] = (int32_t)(keys[g.off[s] + i] & ((1u << g.key_bits) - 1u));
}

static size_t multi_temp_bytes(int64_t total) {
  size_t a = 0, b = 0;
  if (rocprim::radix_sort_pairs<rocprim::default_config>(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                                         (int32_t*)nullptr, (unsigned)total, 0u, 32u) != hipSuccess) return 0;
  if (rocprim::radix_sort_pairs<OnesweepCfg>(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                             (int32_t*)nullptr, (unsigned)total, 0u, 32u) != hipSuccess) return 0;
  return std::max(a, b);
}

size_t group_plan_multi_workspace_bytes(int64_t total) {
  if (total <= 0) return 256;
  const size_t t = multi_temp_bytes(total);
  if (t == 0) return 0;
  return al256(t) + 4 * al256((size_t)total * 4) + 256;
}

int g_group_plan_multi_onesweep = 1;   // vc_debug_set "group_plan_multi_onesweep": 0 = rocPRIM's own choice (merge sort below 1 M rows)

int group_plan_multi(GroupMulti& g, void* ws, size_t ws_bytes, hipStream_t st) {
  int64_t total = 0, nmax = 1;
  unsigned blocks = 0;
  for (int s = 0; s < g.n_tables; ++s) {
    g.off[s] = total;
    g.block0[s] = blocks;
    total += g.n[s];
    blocks += (unsigned)cdiv(g.n[s], 256);
    nmax = std::max(nmax, g.n[s]);
  }
  if (total == 0) return VC_OK;
  int tag_bits = 0;
  while ((1 << tag_bits) < g.n_tables) ++tag_bits;
  g.key_bits = key_bits(nmax);
  if (g.key_bits + tag_bits > 32 || total >= (1LL << 31)) { set_error("group_plan_multi: keys do not fit 32 bits"); return VC_ECAPACITY; }
  const size_t arr = al256((size_t)total * 4);
  const size_t need = group_plan_multi_workspace_bytes(total);
  if (need == 0 || ws_bytes < need) { set_error("group_plan_multi: workspace too small"); return VC_ECAPACITY; }
  uint32_t* keys_in = (uint32_t*)ws;
  int32_t* rows_in = (int32_t*)((char*)ws + arr);
  uint32_t* keys_out = (uint32_t*)((char*)ws + 2 * arr);
  int32_t* rows_out = (int32_t*)((char*)ws + 3 * arr);
  void* temp = (char*)ws + 4 * arr;
  size_t temp_bytes = ws_bytes - 4 * arr;
  hipLaunchKernelGGL(group_keys_multi_kernel, dim3(blocks), dim3(256), 0, st, g, keys_in, rows_in);
  VC_CHECK_LAUNCH("group_keys_multi_kernel");
  const unsigned bits = (unsigned)(g.key_bits + tag_bits);
  if (g_group_plan_multi_onesweep)
    VC_CHECK_HIP(rocprim::radix_sort_pairs<OnesweepCfg>(temp, temp_bytes, (const uint32_t*)keys_in, keys_out, (const int32_t*)rows_in, rows_out,
                                                        (unsigned)total, 0u, bits, st));
  else
    VC_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys_in, keys_out, (const int32_t*)rows_in, rows_out,
                                           (unsigned)total, 0u, bits, st));
  hipLaunchKernelGGL(group_split_multi_kernel, dim3(blocks), dim3(256), 0, st, g, (const uint32_t*)keys_out, (const int32_t*)rows_out);
  VC_CHECK_LAUNCH("group_split_multi_kernel");
  return VC_OK;
}
}  // namespace vc

using namespace vc;

extern "C" {

int vc_group_keys(const int32_t* rep, int64_t n, int32_t* keys, void* stream) {
  VC_REQUIRE(n >= 0 && (n == 0 || (rep && keys)), "vc_group_keys: null/invalid argument");
  if (n == 0) return VC_OK;
  hipLaunchKernelGGL(group_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rep, n, (uint32_t*)keys,
                     (int32_t*)nullptr);
  VC_CHECK_LAUNCH("group_keys_kernel");
  return VC_OK;
}

// The plan itself: keys + row ids, one stable LSD radix sort of the (key, row) pairs over the ceil(log2 n) significant key bits
// (rocPRIM's device radix sort: geometry-plan index work, like the rulebooks; a handful of launches).
static size_t rocprim_plan_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  const size_t a = rocprim_temp_bytes<rocprim::default_config>(n), b = rocprim_temp_bytes<OnesweepCfg>(n);
  if (a == 0 || b == 0) return 0;
  return ((std::max(a, b) + 255) & ~(size_t)255) + 2 * (((size_t)n * 4 + 255) & ~(size_t)255) + 256;
}

// The plan itself: keys + row ids, then rocPRIM's device radix sort of the (key, row) pairs over the ceil(log2 n) significant bits.
// (A hand-written two-pass LDS radix sort was built and measured slower on these key distributions: csrc/experiments/.)
size_t vc_group_plan_workspace_bytes(int64_t n) {
  const size_t a = rocprim_plan_workspace_bytes(n);
  if (a == 0) return 0;
#ifdef VC_EXPERIMENTS
  return std::max(a, radix_plan_workspace_bytes(n > 0 ? n : 1));   // either route fits (the switch is an A/B knob)
#else
  return a;
#endif
}

int vc_group_plan(const int32_t* rep, int64_t n, int32_t* grp_plan, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(n >= 0 && n < (1LL << 31), "vc_group_plan: invalid row count");
  if (n == 0) return VC_OK;
  VC_REQUIRE(rep && grp_plan && ws, "vc_group_plan: null argument");
  const size_t need = vc_group_plan_workspace_bytes(n);
  if (need == 0 || ws_bytes < need) { set_error("vc_group_plan: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
#ifdef VC_EXPERIMENTS
  if (g_group_plan_radix && rs_digit_bits(n) != 0) return radix_group_plan(rep, n, grp_plan, ws, ws_bytes, st);
#endif
  const size_t arr = ((size_t)n * 4 + 255) & ~(size_t)255;
  uint32_t* keys_in = (uint32_t*)ws;
  int32_t* rows_in = (int32_t*)((char*)ws + arr);
  void* temp = (char*)ws + 2 * arr;
  size_t temp_bytes = ws_bytes - 2 * arr;
  hipLaunchKernelGGL(group_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, rep, n, keys_in, rows_in);
  VC_CHECK_LAUNCH("group_keys_kernel");
  if (g_group_plan_onesweep)
    VC_CHECK_HIP(rocprim::radix_sort_pairs<OnesweepCfg>(temp, temp_bytes, (const uint32_t*)keys_in, (uint32_t*)(grp_plan + n),
                                                        (const int32_t*)rows_in, grp_plan, (unsigned)n, 0u, (unsigned)key_bits(n), st));
  else
    VC_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys_in, (uint32_t*)(grp_plan + n), (const int32_t*)rows_in,
                                           grp_plan, (unsigned)n, 0u, (unsigned)key_bits(n), st));
  return VC_OK;
}

size_t vc_group_sum_sorted_workspace_bytes(int64_t n, int c) {
  if (n < 0 || c < 1) return 0;
  return (size_t)cdiv(n, kSegRows) * 2 * c * sizeof(float) + 256;
}

int vc_group_sum_sorted(const float* dy, const int32_t* grp_plan, int64_t n, int c, float* dy_grp, void* ws, size_t ws_bytes,
                        void* stream) {
  VC_REQUIRE(n >= 0 && c >= 1 && c <= 256 && (c & (c - 1)) == 0, "vc_group_sum_sorted: channel count must be a power of two <= 256 (got %d)", c);
  if (n == 0) return VC_OK;
  VC_REQUIRE(dy && grp_plan && dy_grp && ws, "vc_group_sum_sorted: null argument");
  VC_REQUIRE(n < (1LL << 31), "vc_group_sum_sorted: too many rows");
  if (ws_bytes < vc_group_sum_sorted_workspace_bytes(n, c)) { set_error("vc_group_sum_sorted: workspace too small"); return VC_ECAPACITY; }
  int lg = 0;
  while ((1 << lg) < c) ++lg;
  hipStream_t st = (hipStream_t)stream;
  const int32_t* order = grp_plan;
  const int32_t* keys = grp_plan + n;
  const int64_t threads = cdiv(n, kSegRows) * c;
  if (n > kSegRows) {
    hipLaunchKernelGGL(seg_sum_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, dy, order, keys, n, c, lg, dy_grp,
                       (float*)ws);
    VC_CHECK_LAUNCH("seg_sum_kernel");
    VC_LAUNCH_WITH_STOP_EVENT(seg_fixup_kernel, dim3((unsigned)(cdiv(n, kSegRows) - 1)), dim3(256), 0, st, keys, n, c, lg,
                              (const float*)ws, dy_grp);
    VC_CHECK_LAUNCH("seg_fixup_kernel");
  } else {
    VC_LAUNCH_WITH_STOP_EVENT(seg_sum_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, dy, order, keys, n, c, lg, dy_grp,
                              (float*)ws);
    VC_CHECK_LAUNCH("seg_sum_kernel");
  }
  return VC_OK;
}

// ---- developer check of the stop-event dependency (tests/test_round3_gpu.py): a slow producer on `stream_a`, a consumer on a
// second stream.  mode 1: the producer's completion event bound by its launch (VC_LAUNCH_WITH_STOP_EVENT); mode 0: hipEventRecord
// behind it; mode 2: no dependency at all (negative control).  out[i] = what the consumer saw of buf[i].
int vc_debug_stop_event_dependency(int32_t* buf, int32_t* out, int64_t n, int spin, int mode, void* stream_a) {
  VC_REQUIRE(buf && out && n >= 1 && spin >= 0 && mode >= 0 && mode <= 2, "vc_debug_stop_event_dependency: invalid argument");
  hipStream_t a = (hipStream_t)stream_a, b = nullptr;
  hipEvent_t ev = nullptr;
  VC_CHECK_HIP(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  VC_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  VC_CHECK_HIP(hipMemsetAsync(buf, 0, (size_t)n * 4, a));
  VC_CHECK_HIP(hipMemsetAsync(out, 0xFF, (size_t)n * 4, a));
  VC_CHECK_HIP(hipStreamSynchronize(a));
  const dim3 grid((unsigned)cdiv(n, 256));
  if (mode == 1) {
    t_stop_event = StopEventSlot{ev, false};
    VC_LAUNCH_WITH_STOP_EVENT(dbg_slow_fill_kernel, grid, dim3(256), 0, a, buf, n, spin);
    const bool bound = t_stop_event.bound;
    t_stop_event = StopEventSlot{};
    if (!bound) { set_error("vc_debug_stop_event_dependency: the launch did not take the event"); return VC_EHIP; }
  } else {
    hipLaunchKernelGGL(dbg_slow_fill_kernel, grid, dim3(256), 0, a, buf, n, spin);
    if (mode == 0) VC_CHECK_HIP(hipEventRecord(ev, a));
  }
  VC_CHECK_LAUNCH("dbg_slow_fill_kernel");
  if (mode != 2) VC_CHECK_HIP(hipStreamWaitEvent(b, ev, 0));
  hipLaunchKernelGGL(dbg_copy_kernel, grid, dim3(256), 0, b, (const int32_t*)buf, out, n);
  VC_CHECK_LAUNCH("dbg_copy_kernel");
  VC_CHECK_HIP(hipStreamSynchronize(b));
  VC_CHECK_HIP(hipStreamSynchronize(a));
  VC_CHECK_HIP(hipEventDestroy(ev));
  VC_CHECK_HIP(hipStreamDestroy(b));
  return VC_OK;
}

size_t vc_weighted_sum_workspace_bytes(int64_t nb, int64_t e) {
  if (nb < 1 || e < 4) return 0;
  // one fp64 partial per block of the launch: the grid of weighted_sum_grid (mode 0 with more than kWsMaxBlocks samples launches
  // one block per sample, i.e. MORE than kWsMaxBlocks partials)
  int mode;
  dim3 grid;
  weighted_sum_grid(nb, e, mode, grid);
  const size_t blocks = (size_t)grid.x * grid.y;
  return (blocks > (size_t)kWsMaxBlocks ? blocks : (size_t)kWsMaxBlocks) * sizeof(double) + 256;
}

int vc_weighted_sum(const float* x, int64_t nb, int64_t e, const float* g, float* out, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(x && g && out && ws && nb >= 1 && e >= 4 && e % 4 == 0, "vc_weighted_sum: null/invalid argument (e must be a multiple of 4)");
  VC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)g % 16) == 0, "vc_weighted_sum: x and g must be 16-byte aligned");
  int mode;
  dim3 grid;
  weighted_sum_grid(nb, e, mode, grid);
  VC_REQUIRE(mode == 1 || nb <= 65535, "vc_weighted_sum: too many samples for a large row length (nb = %lld)", (long long)nb);
  if (ws_bytes < vc_weighted_sum_workspace_bytes(nb, e)) { set_error("vc_weighted_sum: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  if (mode == 1) hipLaunchKernelGGL((weighted_sum_kernel<1>), grid, dim3(kWsThreads), 0, st, x, nb, e, g, partial);
  else hipLaunchKernelGGL((weighted_sum_kernel<0>), grid, dim3(kWsThreads), 0, st, x, nb, e, g, partial);
  VC_CHECK_LAUNCH("weighted_sum_kernel");
  hipLaunchKernelGGL(weighted_sum_final_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, (int)(grid.x * grid.y), out);
  VC_CHECK_LAUNCH("weighted_sum_final_kernel");
  return VC_OK;
}

int vc_weighted_sum_backward(const float* gout, const float* g, int64_t nb_out, int64_t e, float* dx, void* stream) {
  VC_REQUIRE(gout && g && dx && nb_out >= 1 && e >= 4 && e % 4 == 0, "vc_weighted_sum_backward: null/invalid argument");
  const int64_t total = nb_out * (e >> 2);
  int64_t blocks = cdiv(total, 256 * 4);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(weighted_sum_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gout, g, nb_out, e, dx);
  VC_CHECK_LAUNCH("weighted_sum_backward_kernel");
  return VC_OK;
}

}  // extern "C"
