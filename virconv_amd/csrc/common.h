// Shared host/device helpers for libvirconv_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/virconv_hip.h"

namespace vc {

void set_error(const char* fmt, ...);

#define VC_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ::vc::set_error(__VA_ARGS__);      \
      return VC_EINVAL;                  \
    }                                    \
  } while (0)

#define VC_CHECK_LAUNCH(name)                                                      \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      ::vc::set_error("%s: HIP launch error: %s", name, hipGetErrorString(e__));   \
      return VC_EHIP;                                                              \
    }                                                                              \
  } while (0)

#define VC_CHECK_HIP(expr)                                                         \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) {                                                       \
      ::vc::set_error("%s: %s", #expr, hipGetErrorString(e__));                    \
      return VC_EHIP;                                                              \
    }                                                                              \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Internal 3-D view of a 2-D or 3-D index row: 2-D tensors are (D=1, H=S0, W=S1) with z = 0.
struct Dims {
  int ndim;  // 2 or 3
  int D, H, W;
};

static inline Dims make_dims(int ndim, const int32_t* shape) {
  Dims d;
  d.ndim = ndim;
  if (ndim == 3) {
    d.D = shape[0]; d.H = shape[1]; d.W = shape[2];
  } else {
    d.D = 1; d.H = shape[0]; d.W = shape[1];
  }
  return d;
}

// The image-space (2-D) branch of every block of a geometry plan in ONE launch per step (plan.hip): stage s covers blocks
// [block0[s], block0[s + 1]) of the fused grid.
struct Uv2dStage {
  const int32_t* coords;   // (n, 4) int32 [b, z, y, x]
  int32_t* uv;             // (n, 3) int32 [b, u, v]
  int32_t* img;            // (B, U, V) int32: 1 + highest row of the pixel, 0 = empty
  int32_t* pair;           // (kv, n) pair table of the 2-D SubM conv
  int32_t* rep;            // (n) representative row of each row's pixel
  int64_t n;
  int stride, U, V, SH, SW, ky, kx, dy, dx;
  float vs, minx, miny, minz;
  unsigned block0_mark;    // first block of the stage in the projection + mark grid (256 rows per block)
  unsigned block0_rule;    // ... in the rulebook grid (64 rows per block)
};
struct Uv2dArgs {
  Uv2dStage st[8];
  int n_stages;
  int B;
  const float* params;     // (B, 32) projection parameters (project_prepare_kernel)
};
int uv_mark_multi(const Uv2dArgs& a, unsigned total_blocks, hipStream_t st);   // index_kernels.hip (-ffp-contract=off)

// The group plans (rows sorted by representative, group_kernels.hip) of ALL image-space tables of a plan by ONE sort: the key of
// a row is (table index << key_bits) | representative, so the tables come out one after the other, each in the order its own
// stable sort gives (bit-identical to vc_group_plan per table).  plan[s] = [order (n_s)][sorted keys (n_s)] as vc_group_plan writes it.
struct GroupMulti {
  const int32_t* rep[8];
  int32_t* plan[8];
  int64_t n[8];
  int64_t off[8];       // first position of table s in the concatenated arrays
  unsigned block0[8];   // first block of table s (blocks of 256 rows, whole blocks per table)
  int n_tables;
  int key_bits;
};
size_t group_plan_multi_workspace_bytes(int64_t total_rows);
// fills off / block0 / key_bits from rep / plan / n / n_tables; VC_ECAPACITY when the keys do not fit 32 bits or the workspace is short
int group_plan_multi(GroupMulti& g, void* ws, size_t ws_bytes, hipStream_t st);
// While one is alive (geometry plan only) vc_spconv_mark_count* / vc_spconv_pairs skip the fill of their bitmap / forward pair table:
// the plan keeps those of all its convs in one zone each and fills the zone once.
extern thread_local bool t_sp_skip_clear;
struct SpSkipClear {
  bool prev;
  explicit SpSkipClear(bool on = true) : prev(t_sp_skip_clear) { if (on) t_sp_skip_clear = true; }
  ~SpSkipClear() { t_sp_skip_clear = prev; }
};

struct Kern3 {  // kernel geometry in the internal 3-D view
  int k[3], s[3], p[3], d[3];
  int kv;
};

static inline Kern3 make_kern(int ndim, const int32_t* ks, const int32_t* st, const int32_t* pd, const int32_t* dl) {
  Kern3 g;
  for (int a = 0; a < 3; ++a) { g.k[a] = 1; g.s[a] = 1; g.p[a] = 0; g.d[a] = 1; }
  int o = 3 - ndim;
  for (int a = 0; a < ndim; ++a) {
    g.k[o + a] = ks[a];
    if (st) g.s[o + a] = st[a];
    if (pd) g.p[o + a] = pd[a];
    if (dl) g.d[o + a] = dl[a];
  }
  g.kv = g.k[0] * g.k[1] * g.k[2];
  return g;
}

#ifdef __HIPCC__
__device__ __forceinline__ void load_coord(const int32_t* __restrict__ indices, int64_t i, int ndim, int& b, int& z,
                                           int& y, int& x) {
  const int32_t* r = indices + i * (ndim + 1);
  b = r[0];
  if (ndim == 3) { z = r[1]; y = r[2]; x = r[3]; }
  else { z = 0; y = r[1]; x = r[2]; }
}

__device__ __forceinline__ uint64_t mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

static constexpr uint64_t kEmptyKey = ~0ULL;

// Point-wise pseudo-random PERMUTATION of [0, n): 4-round Feistel network over 2*half_bits bits (2^(2h) >= n) + cycle walking
// (re-encrypt until the value is < n).  A bijection of [0, n) that can be evaluated at any index without sorting n keys.
// Used by the layer discard (vc_random_keep) and the input point discard (vc_input_discard).
__device__ __forceinline__ uint32_t feistel_round(uint32_t r, uint32_t key) {
  uint32_t v = r ^ key;
  v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
  return v;
}
__device__ __forceinline__ int feistel_half_bits(uint64_t n) {
  int half = 1;
  while (half < 31 && (1ULL << (2 * half)) < n) ++half;
  return half;
}
__device__ __forceinline__ uint64_t feistel_perm(uint64_t i, uint64_t n, int half_bits, uint64_t seed) {
  const uint32_t mask = (half_bits >= 32) ? 0xffffffffu : ((1u << half_bits) - 1u);
  uint32_t k[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) k[r] = (uint32_t)(mix64(seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(r + 1)) >> 16);
  uint64_t x = i;
  do {
    uint32_t l = (uint32_t)(x >> half_bits) & mask, rr = (uint32_t)x & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t t = l ^ (feistel_round(rr, k[r]) & mask);
      l = rr;
      rr = t;
    }
    x = ((uint64_t)l << half_bits) | rr;
  } while (x >= n);
  return x;
}

// Coordinate hash (K3), workspace = [keys: cap x u64][vals: cap x i32].  LOCALITY-PRESERVING: the key is the linear
// voxel index (x fastest); the 8 keys of an aligned x-octet map to 8 CONSECUTIVE slots (one 64-byte run of `keys`), only
// the octet id is scrambled, and a collision jumps a whole octet (slot + 8), which keeps the low bits.  The rulebook
// kernels probe x-1, x, x+1 of nine (z, y) lines for x-consecutive rows, so a wave's probes of one offset fall into a
// handful of cache lines instead of 64 random ones (PMC before: subm_rulebook fetched 274 MB per launch against 44 MB
// algorithmic).  An x-octet chain only ever holds keys with the same low 3 bits, i.e. at most one per octet, so the
// capacity is sized on octets (see the note below on how many).
// Round 3: octets = next_pow2(> n) (half the former next_pow2(>= 2n)).  A chain holds at most one key per octet and only keys of
// one residue class (x mod 8), so even the worst case -- every voxel alone in its octet, all in the same class -- leaves an
// empty slot in every chain (insert and lookup terminate); a real tensor has ~n/8 keys per class, i.e. a chain load <= 1/8.
// Halves the 0xFF fill in front of every hash build (100 -> 50 MB for the 310 k-row tensors) and the table's cache footprint.
__device__ __forceinline__ uint64_t coord_slot(uint64_t key, uint64_t mask) {
  return ((mix64(key >> 3) << 3) | (key & 7ULL)) & mask;
}
__device__ __forceinline__ uint64_t coord_next(uint64_t slot, uint64_t mask) { return (slot + 8) & mask; }

__device__ __forceinline__ int hash_lookup(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                           uint64_t mask, uint64_t key) {
  uint64_t slot = coord_slot(key, mask);
  for (;;) {
    uint64_t k = keys[slot];
    if (k == key) return vals[slot];
    if (k == kEmptyKey) return -1;
    slot = coord_next(slot, mask);
  }
}
#endif

static inline uint64_t hash_capacity(int64_t n) {  // point hash of the voxelizer: plain open addressing, load <= 1/2
  uint64_t cap = 1024;
  while (cap < (uint64_t)(2 * n)) cap <<= 1;
  return cap;
}

// BatchNorm sums out of conv-epilogue partial rows on the two-launch route (bn_partial_reduce_kernel + finalize): the rows are
// reduced in <= 256 groups of `bn_partial_rows_per_group` consecutive rows (a multiple of 8), then over the groups.  (The in-kernel
// finish of the gather-GEMM, conv_finish_ticket / _reduce, has its own three levels: conv_kernels.hip.)
static inline int64_t bn_partial_rows_per_group(int64_t nrows) { return (cdiv(nrows, 256) + 7) & ~(int64_t)7; }

// In-kernel finish of the BatchNorm sums (conv_kernels.hip): a caller arms a request, the next epilogue launch of this host
// thread takes it if its kernel supports the finish (the v2 gather-GEMM with > 512 partial rows), conv_finish_take() tells.
struct BnFinishRequest {
  int bwd;             // 0: statistics (mean, var, running statistics); 1: backward sums (dbeta, dgamma, sums[2][c])
  int64_t n;           // rows of the tensor the statistics are over
  float *o0, *o1;      // stats: mean, var;  bwd: dbeta, dgamma (may be null)
  float *r0, *r1;      // stats: running mean / var (may be null);  bwd: r0 = sums[2][c]
  long long* nbt;      // stats: num_batches_tracked (may be null)
  float momentum;
  double* dpartial;    // >= 256 x 2c doubles of scratch, private to the launch while it runs
};
void conv_finish_arm(const BnFinishRequest& r);
bool conv_finish_take(hipStream_t st);   // true: the sums are finished (by the launch since the last arm); always disarms
// the dx kernel of the BatchNorm(+ReLU) backward alone, the per-channel sums given (bn_kernels.hip)
int bn_bwd_dx_launch(const float* x, const float* dy, int dy_stride, int dy_col0, int64_t n, int c, const float* mean,
                     const float* var, const float* gamma, const float* beta, float eps, int relu, const float* sums, float* dx,
                     hipStream_t st);

// Cross-stream dependency without a marker packet.  hipEventRecord puts a barrier packet into the producer's queue; between two
// short kernels of the main stream that packet costs 5-8 us of queue time (profiles/r03_trace_gaps.txt: the gap in front of every
// backward-input conv whose unit forks its weight gradient onto the side stream).  hipExtLaunchKernelGGL can instead bind an event
// to the completion signal of the kernel it launches.  A caller arms the slot with an event right before calling an operator; the
// operator's LAST launch (VC_LAUNCH_WITH_STOP_EVENT) takes it; the caller then waits on the event from the other stream, or falls
// back to hipEventRecord when nothing took it.
struct StopEventSlot {
  hipEvent_t ev = nullptr;
  bool bound = false;
};
inline thread_local StopEventSlot t_stop_event;
#define VC_LAUNCH_WITH_STOP_EVENT(kernel, grid, block, lds, st, ...)                                              \
  do {                                                                                                            \
    if (::vc::t_stop_event.ev != nullptr && !::vc::t_stop_event.bound) {                                          \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, nullptr, ::vc::t_stop_event.ev, 0, __VA_ARGS__);         \
      ::vc::t_stop_event.bound = true;                                                                            \
    } else {                                                                                                      \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                              \
    }                                                                                                             \
  } while (0)

// kernel timing of a bandwidth-bound kernel (vc_trace_begin direction 3 = bn_bwd_dx_pow2_kernel, the largest of the step by time;
// conv_kernels.hip): start / stop events around the launch on its stream, record = {channels, rows}
int trace_open_aux(int dir, int c, hipStream_t st);
void trace_close_aux(int i, int dir, int c, int64_t n, hipStream_t st);

static inline uint64_t coord_hash_capacity(int64_t n) {
  uint64_t oct = 128;
  while (oct <= (uint64_t)n) oct <<= 1;
  return oct * 8;
}

}  // namespace vc
