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
  const int key = keys[j1 - 1];
  if (keys[j1] != key) return;                // last run ends with the chunk
  if (keys[j0] == key && j0 > 0 && keys[j0 - 1] == key) return;   // the whole chunk is a continuation: not the run's first chunk
  // end of the run: first position e in (j1, n] with keys[e] != key (keys ascending); every thread searches for itself (uniform)
  int64_t lo = j1, hi = n;                    // invariant: keys[lo] == key, (hi == n or keys[hi] != key)
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] == key) lo = mid; else hi = mid;
  }
  const int64_t c_end = (hi - 1) / kSegRows;  // last chunk holding rows of the run
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

// keys[i] = rep[i] < 0 ? i : rep[i]  (what the plan sorts; rep = -1 marks "own representative" in some tables); rows[i] = i
__global__ void __launch_bounds__(256) group_keys_kernel(const int32_t* __restrict__ rep, int64_t n, uint32_t* __restrict__ keys,
                                                         int32_t* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int g = rep[i];
  keys[i] = g < 0 ? (uint32_t)i : (uint32_t)g;
  if (rows != nullptr) rows[i] = (int32_t)i;
}

static inline int key_bits(int64_t n) {
  int b = 1;
  while (b < 32 && (1LL << b) < n) ++b;
  return b;
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
size_t vc_group_plan_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  size_t temp = 0;
  if (rocprim::radix_sort_pairs(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                (unsigned)n, 0u, (unsigned)key_bits(n)) != hipSuccess)
    return 0;
  return ((temp + 255) & ~(size_t)255) + 2 * (((size_t)n * 4 + 255) & ~(size_t)255) + 256;
}

int vc_group_plan(const int32_t* rep, int64_t n, int32_t* grp_plan, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(n >= 0 && n < (1LL << 31), "vc_group_plan: invalid row count");
  if (n == 0) return VC_OK;
  VC_REQUIRE(rep && grp_plan && ws, "vc_group_plan: null argument");
  const size_t need = vc_group_plan_workspace_bytes(n);
  if (need == 0 || ws_bytes < need) { set_error("vc_group_plan: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  const size_t arr = ((size_t)n * 4 + 255) & ~(size_t)255;
  uint32_t* keys_in = (uint32_t*)ws;
  int32_t* rows_in = (int32_t*)((char*)ws + arr);
  void* temp = (char*)ws + 2 * arr;
  size_t temp_bytes = ws_bytes - 2 * arr;
  hipLaunchKernelGGL(group_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, rep, n, keys_in, rows_in);
  VC_CHECK_LAUNCH("group_keys_kernel");
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
  hipLaunchKernelGGL(seg_sum_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, dy, order, keys, n, c, lg, dy_grp,
                     (float*)ws);
  VC_CHECK_LAUNCH("seg_sum_kernel");
  if (n > kSegRows) {
    hipLaunchKernelGGL(seg_fixup_kernel, dim3((unsigned)(cdiv(n, kSegRows) - 1)), dim3(256), 0, st, keys, n, c, lg, (const float*)ws,
                       dy_grp);
  }
  VC_CHECK_LAUNCH("seg_fixup_kernel");
  return VC_OK;
}

}  // extern "C"
