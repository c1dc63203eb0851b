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
//   * seg_fixup_kernel   thread = (chunk whose last run began there and continues, channel): binary-searches the end of the run
//                        and adds the continuation partials of the following chunks in chunk order.
// No atomics, no absmax pass, no accumulator buffer, no convert pass; every group's additions happen in one fixed order, so
// the result is bit-stable run to run.  HBM-bound: one gathered read of dy (rows are 32-256 B contiguous) + 8 B of plan per row.
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

__global__ void __launch_bounds__(256) seg_fixup_kernel(const int32_t* __restrict__ keys, int64_t n, int c, int lg_c,
                                                        const float* __restrict__ part, float* __restrict__ grp) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t chunk = t >> lg_c;
  const int ch = (int)(t & (c - 1));
  const int64_t j0 = chunk * kSegRows;
  const int64_t j1 = j0 + kSegRows;
  if (j1 >= n) return;                        // the last chunk cannot have an open run
  const int key = keys[j1 - 1];
  if (keys[j1] != key) return;                // last run ends with the chunk
  if (keys[j0] == key && j0 > 0 && keys[j0 - 1] == key) return;   // the whole chunk is a continuation: not the run's first chunk
  // end of the run: first position e in (j1, n] with keys[e] != key (keys ascending)
  int64_t lo = j1, hi = n;                    // invariant: keys[lo] == key, (hi == n or keys[hi] != key)
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] == key) lo = mid; else hi = mid;
  }
  const int64_t c_end = (hi - 1) / kSegRows;  // last chunk holding rows of the run
  float total = part[(chunk * 2 + 1) * c + ch];
  int64_t q = chunk + 1;
  for (; q + 8 <= c_end + 1; q += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[((q + u) * 2 + 0) * c + ch];
#pragma unroll
    for (int u = 0; u < 8; ++u) total += v[u];
  }
  for (; q <= c_end; ++q) total += part[(q * 2 + 0) * c + ch];
  grp[(int64_t)key * c + ch] = total;
}

// keys[i] = rep[i] < 0 ? i : rep[i]  (what the plan sorts; rep = -1 marks "own representative" in some tables)
__global__ void __launch_bounds__(256) group_keys_kernel(const int32_t* __restrict__ rep, int64_t n, int32_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int g = rep[i];
  keys[i] = g < 0 ? (int32_t)i : g;
}

}  // namespace vc

using namespace vc;

extern "C" {

int vc_group_keys(const int32_t* rep, int64_t n, int32_t* keys, void* stream) {
  VC_REQUIRE(n >= 0 && (n == 0 || (rep && keys)), "vc_group_keys: null/invalid argument");
  if (n == 0) return VC_OK;
  hipLaunchKernelGGL(group_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rep, n, keys);
  VC_CHECK_LAUNCH("group_keys_kernel");
  return VC_OK;
}

size_t vc_group_sum_sorted_workspace_bytes(int64_t n, int c) {
  if (n < 0 || c < 1) return 0;
  return (size_t)cdiv(n, kSegRows) * 2 * c * sizeof(float) + 256;
}

int vc_group_sum_sorted(const float* dy, const int32_t* grp_plan, int64_t n, int c, float* dy_grp, void* ws, size_t ws_bytes,
                        void* stream) {
  VC_REQUIRE(n >= 0 && c >= 1 && (c & (c - 1)) == 0, "vc_group_sum_sorted: channel count must be a power of two (got %d)", c);
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
  hipLaunchKernelGGL(seg_fixup_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, keys, n, c, lg, (const float*)ws,
                     dy_grp);
  VC_CHECK_LAUNCH("seg_fixup_kernel");
  return VC_OK;
}

}  // extern "C"
