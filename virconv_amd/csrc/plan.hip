// Geometry plan executor (include/virconv_hip.h, "geometry plan"): every index structure of a chain of NRConvBlocks
// (pcdet/models/backbones_3d/spconv_backbone.py:150-229; chained in VirConvL8x.forward :609-699 and the MM stream of
// VirConv8x.forward :444-535) from TWO calls around ONE host read.  In the reference this work is spconv's indice generation,
// run conv by conv with a device-to-host sync per strided conv; round 3 issued it from Python (~90 ctypes
// calls, ~60 allocations, four pipelined count reads per train step: 2.2 ms of host time and 0.84 ms of step time, measured with
// tools/whatif.py).
//
// What is new here besides the composition:
//   * device-side row counts through the whole chain: a strided conv marks from coordinates whose count is still on the device
//     (vc_spconv_mark_count_dev), the layer discard draws / gathers its kept rows for a device-side count, capacities bound the
//     buffers -- one host read for all levels, also WITH layer discard between the levels;
//   * subm_bitmap_rulebook_kernel: the rows of a strided conv's output are the set bits of its occupancy bitmap in ascending
//     order, so "coordinate -> row" is prefix[word] + popcount(word & below): the 3-D SubM rulebook of stages 2-4 probes that
//     bitmap (11.5 / 1.4 / 0.2 MB, cache resident) instead of building and probing a 25-50 MB hash table;
//   * image_mark / image_rulebook kernels: the pixel tensors of the 2-D branch index a dense per-sample image
//     (B x U x V int32, 13.4 MB at stride 1 ... 0.2 MB at stride 8) -- duplicate rule max row by atomicMax, neighbours by address;
//   * parity_order_kernel: the active-offset set of an INPUT row of a strided conv is a function of its coordinate's residue
//     modulo the stride (+ grid borders), so the backward row order is a stable counting sort on <= 16 residue classes per
//     2048-row window, read from the coordinates alone (vc_row_order reads the 27-row table: 45 us -> 6 us per table);
//   * sp_mark2_kernel (index_kernels.hip): output cells enumerated from the output side, 8 candidates instead of 27 offset tests.
// (A hand-written LDS radix sort for the group plans was built too and lost to rocPRIM on these keys: csrc/experiments/.)
// Tables are bit-identical to the stand-alone operators on the same coordinates (tests/test_plan_gpu.py); row orders are hints.
#include <algorithm>
#include <mutex>

#include "common.h"

namespace vc {

int g_plan_subm_bitmap = 1;   // vc_debug_set plan_subm_bitmap: 0 = hash build + vc_subm_rulebook for every 3-D SubM table (A/B)
int g_plan_group_multi = 1;   // vc_debug_set plan_group_multi: 0 = one sort per image-space table (vc_group_plan) instead of one for all (A/B)
int g_plan_image_2d = 1;      // vc_debug_set plan_image_2d:    0 = hash build + vc_subm_rulebook for the pixel tables (A/B)
int g_plan_parity_order = 1;  // vc_debug_set plan_parity_order: 0 = vc_row_order on the pair table (A/B)
// developer diagnostics of LOG.md A.15 (tools/det_check.py), both 0 on the product path:
int g_plan_params_pad = 0;    // vc_debug_set plan_params_pad: bytes of padding in front of the projection parameter block (moves it: does the
                              // damage follow the block or stay at the address?)
int g_plan_reprepare = 0;     // vc_debug_set plan_reprepare: 1 = write the parameter block again right in front of every projection of
                              // vc_plan_finish (damage to the block between the two calls is then repaired: is the block what is hit?)
int g_plan_uv_poison = 0;     // vc_debug_set plan_uv_poison: 1 = every uv buffer is filled with 0x7F bytes in front of the projection (diagnostics)
long long g_plan_uv_dbg = 0;  // vc_debug_set plan_uv_dbg: device address of a diagnostics buffer for the stand-alone projection (project_uv_kernel<MODE, true>: every row's intermediates)
int g_plan_uv_lds = 0;        // vc_debug_set plan_uv_lds: dynamic LDS bytes of the stand-alone projection launch (round 6 lab, tools/a17_lab.py)
int g_plan_uv_pad = 0;        // vc_debug_set plan_uv_pad: uv_mark_multi_kernel<PAD> (index_kernels.hip; LOG.md A.21) -- decided by tools/a17_lab.py stress
int g_plan_uv_mode = 0;       // vc_debug_set plan_uv_mode: project_uv_kernel<MODE> of the plan's projections (0 = product kernel)
int project_uv_debug(const int32_t* indices, int64_t n, const float* params, int batch_size, int stride, int32_t* uv, int32_t* dbg,
                     int dbg_records, int mode, int has_trans, hipStream_t st);   // index_kernels.hip

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------ device-side counts
// n_keep = int(n * (1 - rate)): the same double arithmetic as the reference's Python (spconv_backbone.py:139-141)
__global__ void keep_count_kernel(const int32_t* __restrict__ n_dev, double keep_frac, int32_t* __restrict__ n_keep_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *n_keep_dev = (int32_t)((double)(*n_dev) * keep_frac);
}

__global__ void set_count_kernel(int32_t* __restrict__ dst, int32_t v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}

__global__ void __launch_bounds__(256) random_keep_dev_kernel(const int32_t* __restrict__ n_dev, const int32_t* __restrict__ n_keep_dev,
                                                              uint64_t seed, int64_t* __restrict__ keep) {
  const int64_t n = *n_dev, n_keep = *n_keep_dev;
  const int half = feistel_half_bits((uint64_t)n);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_keep; i += (int64_t)gridDim.x * 256)   // bounded grid over a capacity
    keep[i] = (int64_t)feistel_perm((uint64_t)i, (uint64_t)n, half, seed);
}

// kept[j, :] = idx[keep[j], :] for j < *n_keep_dev (4 ints per row: one 16-byte load / store per thread)
__global__ void __launch_bounds__(256) gather_coords_dev_kernel(const int4* __restrict__ idx, const int64_t* __restrict__ keep,
                                                                const int32_t* __restrict__ n_keep_dev, int64_t n_keep_host,
                                                                int4* __restrict__ kept) {
  const int64_t n_keep = n_keep_dev ? (int64_t)*n_keep_dev : n_keep_host;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n_keep; j += (int64_t)gridDim.x * 256) kept[j] = idx[keep[j]];
}

// ------------------------------------------------------------------------------------------ SubM rulebook by bitmap rank
// Same block shape and probe batching as subm_rulebook_kernel (index_kernels.hip): 64 consecutive rows x 4 offset groups, eight
// probes in flight per thread.  A probe is one 8-byte bitmap word; a hit adds one 4-byte prefix load.
__global__ void __launch_bounds__(256) subm_bitmap_rulebook_kernel(const int32_t* __restrict__ indices, int64_t n, int D, int H, int W,
                                                                   int kz, int ky, int kx, int dz, int dy, int dx,
                                                                   const unsigned long long* __restrict__ bitmap,
                                                                   const uint32_t* __restrict__ prefix, int32_t* __restrict__ pair) {
  __shared__ int s_off[128];
  const int kv = kz * ky * kx;
  for (int k = threadIdx.x; k < kv; k += blockDim.x) s_off[k] = (k / (ky * kx)) | (((k / kx) % ky) << 8) | ((k % kx) << 16);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  if (i >= n) return;
  const int kg = threadIdx.x >> 6;
  const int centre = ((kz / 2) * ky + ky / 2) * kx + kx / 2;
  const int4 c = *reinterpret_cast<const int4*>(indices + i * 4);   // [b, z, y, x]
  for (int k0 = kg; k0 < kv; k0 += 32) {
    int64_t L[8];
    unsigned long long w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 4 * j;
      L[j] = -1;
      w[j] = 0ULL;
      if (k >= kv || k == centre) continue;
      const int off = s_off[k];
      const int nz = c.y + ((off & 255) - kz / 2) * dz, ny = c.z + (((off >> 8) & 255) - ky / 2) * dy, nx = c.w + ((off >> 16) - kx / 2) * dx;
      if (nz >= 0 && nz < D && ny >= 0 && ny < H && nx >= 0 && nx < W) {
        L[j] = (((int64_t)c.x * D + nz) * H + ny) * W + nx;
        w[j] = bitmap[L[j] >> 6];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 4 * j;
      if (k >= kv) continue;
      int r = -1;
      if (k == centre) r = (int)i;
      else if (L[j] >= 0 && ((w[j] >> (L[j] & 63)) & 1ULL))
        r = (int)prefix[L[j] >> 6] + __popcll(w[j] & ((1ULL << (L[j] & 63)) - 1ULL));
      pair[(int64_t)k * n + i] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------ pixel tables on a dense image
// img[(b * U + u) * V + v] = 1 + the highest row whose pixel is (b, u, v) (0: empty): the duplicate rule of the coordinate hash
// (hash_insert_kernel) by address.  Equal pixels sit in consecutive rows (voxels outside the camera frustum clamp onto border
// pixels by the thousand): runs of equal keys are folded in the wave first, one atomic per run.
__global__ void __launch_bounds__(256) image_mark_kernel(const int32_t* __restrict__ uv, int64_t n, int B, int U, int V,
                                                         int32_t* __restrict__ img) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool live = i < n;
  int64_t key = -2 - lane;   // dead lanes: distinct, never equal to a real key
  if (live) {
    const int b = uv[i * 3], u = uv[i * 3 + 1], v = uv[i * 3 + 2];
    if (b >= 0 && b < B && u >= 0 && u < U && v >= 0 && v < V) key = ((int64_t)b * U + u) * V + v;
    else live = false;
  }
  int row = live ? (int)i : -1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t ok = __shfl_up((long long)key, off, 64);
    const int orow = __shfl_up(row, off, 64);
    if (lane >= off && ok == key) row = max(row, orow);
  }
  const int64_t nk = __shfl_down((long long)key, 1, 64);
  if (!live || (lane != 63 && nk == key)) return;
  atomicMax(&img[key], row + 1);
}

// pair[k, i] = row of the pixel (u, v) + offset_k, rep[i] = row of the own pixel.  Block = 64 rows x 4 offset groups.
__global__ void __launch_bounds__(256) image_rulebook_kernel(const int32_t* __restrict__ uv, int64_t n, int B, int U, int V, int SH, int SW,
                                                             int ky, int kx, int dy, int dx, const int32_t* __restrict__ img,
                                                             int32_t* __restrict__ pair, int32_t* __restrict__ rep) {
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  if (i >= n) return;
  const int kg = threadIdx.x >> 6;
  const int kv = ky * kx, centre = (ky / 2) * kx + kx / 2;
  const int b = uv[i * 3], u = uv[i * 3 + 1], v = uv[i * 3 + 2];
  const bool inside = b >= 0 && b < B && u >= 0 && u < U && v >= 0 && v < V;
  for (int k = kg; k < kv; k += 4) {
    int r = -1;
    if (k == centre) {
      r = (int)i;
      if (rep) rep[i] = inside ? img[((int64_t)b * U + u) * V + v] - 1 : (int)i;
    } else {
      const int nu = u + (k / kx - ky / 2) * dy, nv = v + (k % kx - kx / 2) * dx;
      // SH x SW: the tensor's spatial shape (the bound the coordinate hash applies); U x V: the extent pixels can take
      if (b >= 0 && b < B && nu >= 0 && nu < SH && nv >= 0 && nv < SW && nu < U && nv < V) r = img[((int64_t)b * U + nu) * V + nv] - 1;
    }
    pair[(int64_t)k * n + i] = r;
  }
}

// ------------------------------------------------------------------------------------------ backward row order by residue class
// One block per window of WIN = 4 * THREADS consecutive input rows; class(row) = mixed-radix number of (coordinate + padding) mod
// stride over the three axes (NC = s0 * s1 * s2 <= 16 classes).  Stable counting sort: per 64-row chunk the lanes of a class take
// popcount-below as their rank (ballot), an exclusive scan over [class][chunk] gives the bases.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) parity_order_kernel(const int32_t* __restrict__ indices, int64_t n, int s0, int s1, int s2,
                                                               int p0, int p1, int p2, int32_t* __restrict__ order) {
  constexpr int WIN = 4 * THREADS, CH = WIN / 64;
  static_assert(CH * 16 <= THREADS, "one [class][chunk] counter per thread");
  __shared__ int cnt[16 * CH];
  __shared__ int s_wave[THREADS / 64];
  const int nc = s0 * s1 * s2;
  const int64_t base = (int64_t)blockIdx.x * WIN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < 16 * CH; j += THREADS) cnt[j] = 0;
  __syncthreads();
  int cls[4], rin[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t r = base + u * THREADS + threadIdx.x;
    const bool valid = r < n;
    cls[u] = -1;
    rin[u] = 0;
    if (valid) {
      const int4 c = *reinterpret_cast<const int4*>(indices + r * 4);
      cls[u] = (((c.y + p0) % s0) * s1 + ((c.z + p1) % s1)) * s2 + ((c.w + p2) % s2);
    }
    const int chunk = u * (THREADS / 64) + wave;
    for (int cc = 0; cc < nc; ++cc) {
      const unsigned long long mm = __ballot(cls[u] == cc);
      if (cls[u] == cc) rin[u] = __popcll(mm & ((1ULL << lane) - 1ULL));
      if (lane == 0) cnt[cc * CH + chunk] = __popcll(mm);
    }
  }
  __syncthreads();
  {  // exclusive scan of cnt[0 .. nc * CH): one entry per thread
    const int v = ((int)threadIdx.x < nc * CH) ? cnt[threadIdx.x] : 0;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w_ = 0; w_ < wave; ++w_) woff += s_wave[w_];
    if ((int)threadIdx.x < nc * CH) cnt[threadIdx.x] = woff + inc - v;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (cls[u] < 0) continue;
    const int chunk = u * (THREADS / 64) + wave;
    order[base + cnt[cls[u] * CH + chunk] + rin[u]] = (int32_t)(base + u * THREADS + threadIdx.x);
  }
}

// ------------------------------------------------------------------------------------------ host side
static constexpr uint64_t kPlanMagic = 0x56434e4c504c4e31ULL;
static constexpr int kMaxCounts = 64;

struct Conv {                 // one strided conv of the chain
  int present;
  int32_t in_shape[3], out_shape[3];
  int kv;
  int64_t cap_in, cap_out;
  int64_t ws_off;             // arena_a: bitmap | prefix | block sums
  size_t ws_bytes;
  int64_t out_idx_off;        // arena_a: (cap_out, 4) int32
  int cnt_out;                // slot of n_out in counts[]
  int64_t n_in, n_out;        // known after vc_plan_wait
};
struct Blk {
  Conv down;
  int64_t coords_off;         // arena_a offset of the block's coordinates; -1: the caller's indices (first block without strided conv / input discard)
  int64_t cap;                // row capacity of the block's tensor
  int cnt_n;                  // slot of the tensor's row count (-1: known at begin)
  int64_t n;
  int64_t keep_off, kept_off, cap_keep;   // layer discard after the block (arena_a), -1: none
  int cnt_keep;
  int64_t n_keep;
  int32_t shape[3];           // spatial shape of the block's tensor
  int early;                  // 1: the block's SubM / pixel tables are built by vc_plan_begin (row count known)
};
struct PlanState {
  uint64_t magic;
  int n_counts, ev_slot, waited, params_ready, finished;
  uint64_t ev_gen;            // generation of the event slot this plan recorded (vc_plan_wait refuses a slot that moved on)
  int32_t* host_counts;
  int64_t counts_off, params_off;
  int64_t in_keep_off, in_kept_off, cap_in_keep, n_in_keep;   // discard of the chain's input
  int cnt_in_keep;
  int64_t a_bytes;            // arena_a bytes of the chain part (early tables start here)
  int64_t ws_zone_off;        // arena_a: the bitmap workspaces of all strided convs, adjacent (ONE clear per plan)
  size_t ws_zone_bytes;
  Blk blk[VC_PLAN_MAX_BLOCKS];
  Conv tail;
  vc_plan_out early_out;      // views of the tables built at begin
};
static_assert(sizeof(PlanState) <= sizeof(vc_plan_state), "vc_plan_state too small");

struct Bump2 {
  size_t off = 0;
  int64_t take(size_t bytes) {
    const size_t o = off;
    off += al256(bytes);
    return (int64_t)o;
  }
};

// output cells one input row can reach: per axis the kernel's EXTENT (k - 1) * dilation + 1 over the stride, rounded up -- with
// gcd(dilation, stride) > 1 (k3 s2 d2) an input reaches 3 cells per axis, not 2 (sp_mark2 enumerates the same J)
static inline int64_t conv_reach(const vc_plan_conv& g) {
  int64_t r = 1;
  for (int a = 0; a < 3; ++a) r *= ((g.ksize[a] - 1) * g.dilation[a] + 1 + g.stride[a] - 1) / g.stride[a];
  return r;
}
static inline void conv_out_shape(const int32_t* in, const vc_plan_conv& g, int32_t* out) {
  for (int a = 0; a < 3; ++a) out[a] = (in[a] + 2 * g.padding[a] - g.dilation[a] * (g.ksize[a] - 1) - 1) / g.stride[a] + 1;
}

static int check_desc(const vc_plan_desc* d) {
  VC_REQUIRE(d && d->indices && d->n >= 1 && d->batch_size >= 1, "vc_plan: null/empty input");
  VC_REQUIRE(d->n_blocks >= 1 && d->n_blocks <= VC_PLAN_MAX_BLOCKS, "vc_plan: 1..%d blocks", VC_PLAN_MAX_BLOCKS);
  VC_REQUIRE(d->discard_rate >= 0.0 && d->discard_rate < 1.0, "vc_plan: discard_rate must be in [0, 1)");
  VC_REQUIRE(d->n < (1LL << 30), "vc_plan: too many rows");
  for (int a = 0; a < 3; ++a) VC_REQUIRE(d->spatial_shape[a] >= 1, "vc_plan: invalid spatial shape");
  bool any2d = false;
  for (int b = 0; b < d->n_blocks; ++b) {
    const vc_plan_block& B = d->blocks[b];
    for (int a = 0; a < 3; ++a) {
      VC_REQUIRE(B.subm_ksize[a] >= 1 && B.subm_ksize[a] % 2 == 1 && B.subm_dilation[a] >= 1, "vc_plan: block %d: SubM kernel sizes must be odd", b);
      if (B.has_down) VC_REQUIRE(B.down.ksize[a] >= 1 && B.down.stride[a] >= 1 && B.down.padding[a] >= 0 && B.down.dilation[a] >= 1, "vc_plan: block %d: invalid strided conv", b);
    }
    VC_REQUIRE(B.subm_ksize[0] * B.subm_ksize[1] * B.subm_ksize[2] <= 128, "vc_plan: block %d: SubM kernel volume > 128", b);
    if (B.has_2d) {
      any2d = true;
      VC_REQUIRE(B.uv_stride >= 1 && B.ksize2d[0] % 2 == 1 && B.ksize2d[1] % 2 == 1 && B.ksize2d[0] * B.ksize2d[1] <= 128 &&
                 B.dilation2d[0] >= 1 && B.dilation2d[1] >= 1, "vc_plan: block %d: invalid 2-D branch", b);
    }
    VC_REQUIRE(!B.discard || B.keep == nullptr || B.keep_rows >= 1, "vc_plan: block %d: empty injected keep", b);
  }
  VC_REQUIRE(!any2d || (d->calib && d->image_shape[0] >= 1 && d->image_shape[1] >= 1), "vc_plan: the 2-D branch needs calib and image_shape");
  return VC_OK;
}

// Layout of the chain part of arena_a: a pure function of the description (capacities, no counts)
static void layout_chain(const vc_plan_desc* d, PlanState& S, Bump2& bump) {
  int nc = 0;
  S.counts_off = bump.take(kMaxCounts * sizeof(int32_t));
  if (g_plan_params_pad > 0) (void)bump.take((size_t)g_plan_params_pad);
  S.params_off = bump.take((size_t)d->batch_size * 32 * sizeof(float));
  const double keep_frac = 1.0 - d->discard_rate;
  int64_t cur_cap = d->n;        // row capacity of the coordinates feeding the next thing
  int32_t shape[3] = {d->spatial_shape[0], d->spatial_shape[1], d->spatial_shape[2]};
  S.in_keep_off = S.in_kept_off = -1;
  S.cnt_in_keep = -1;
  S.n_in_keep = S.cap_in_keep = 0;
  if (d->input_discard) {
    S.cap_in_keep = d->input_keep ? d->input_keep_rows : (int64_t)((double)d->n * keep_frac);
    S.n_in_keep = S.cap_in_keep;     // the input's row count is the caller's: known now
    S.in_keep_off = bump.take((size_t)std::max<int64_t>(S.cap_in_keep, 1) * 8);
    S.in_kept_off = bump.take((size_t)std::max<int64_t>(S.cap_in_keep, 1) * 16);
    cur_cap = S.cap_in_keep;
  }
  bool counts_known = true;      // every row count so far is known on the host
  Bump2 wsb;                     // the convs' bitmap workspaces: one zone, cleared by one fill in vc_plan_begin
  auto plan_conv = [&](Conv& C, const vc_plan_conv& g) {
    C.present = 1;
    for (int a = 0; a < 3; ++a) C.in_shape[a] = shape[a];
    conv_out_shape(shape, g, C.out_shape);
    C.kv = g.ksize[0] * g.ksize[1] * g.ksize[2];
    C.cap_in = cur_cap;
    const int64_t cells = (int64_t)d->batch_size * C.out_shape[0] * C.out_shape[1] * C.out_shape[2];
    C.cap_out = std::min<int64_t>(cur_cap * conv_reach(g), cells);
    C.ws_bytes = vc_spconv_workspace_bytes(d->batch_size, 3, C.out_shape);
    C.ws_off = wsb.take(C.ws_bytes);     // relative to the zone, rebased below
    C.out_idx_off = bump.take((size_t)std::max<int64_t>(C.cap_out, 1) * 16);
    C.cnt_out = nc++;
    C.n_in = C.n_out = -1;
    cur_cap = C.cap_out;
    for (int a = 0; a < 3; ++a) shape[a] = C.out_shape[a];
    counts_known = false;
  };
  for (int b = 0; b < d->n_blocks; ++b) {
    const vc_plan_block& B = d->blocks[b];
    Blk& K = S.blk[b];
    K = Blk{};
    K.coords_off = -1;
    if (B.has_down) {
      plan_conv(K.down, B.down);
      K.coords_off = K.down.out_idx_off;
      K.cnt_n = K.down.cnt_out;
    } else {
      K.down.present = 0;
      K.cnt_n = (b == 0) ? -1 : S.blk[b - 1].cnt_keep >= 0 ? S.blk[b - 1].cnt_keep : S.blk[b - 1].cnt_n;
      if (b == 0) K.coords_off = d->input_discard ? S.in_kept_off : -1;
      else K.coords_off = S.blk[b - 1].kept_off >= 0 ? S.blk[b - 1].kept_off : S.blk[b - 1].coords_off;
    }
    K.cap = cur_cap;
    K.n = counts_known ? cur_cap : -1;
    K.early = (counts_known && !d->defer_early_tables) ? 1 : 0;
    for (int a = 0; a < 3; ++a) K.shape[a] = shape[a];
    K.keep_off = K.kept_off = -1;
    K.cnt_keep = -1;
    K.n_keep = -1;
    if (B.discard) {
      K.cap_keep = B.keep ? B.keep_rows : (int64_t)((double)cur_cap * keep_frac);
      K.keep_off = bump.take((size_t)std::max<int64_t>(K.cap_keep, 1) * 8);
      K.kept_off = bump.take((size_t)std::max<int64_t>(K.cap_keep, 1) * 16);
      if (counts_known) K.n_keep = K.cap_keep;
      else K.cnt_keep = nc++;
      cur_cap = K.cap_keep;
    }
  }
  S.tail = Conv{};
  if (d->has_tail) plan_conv(S.tail, d->tail);
  S.ws_zone_bytes = wsb.off;
  S.ws_zone_off = bump.take(wsb.off);
  for (int b = 0; b < d->n_blocks; ++b)
    if (S.blk[b].down.present) S.blk[b].down.ws_off += S.ws_zone_off;
  if (S.tail.present) S.tail.ws_off += S.ws_zone_off;
  S.n_counts = nc;
  S.a_bytes = (int64_t)bump.off;
}

// the SubM / pixel tables of block b (and, with_conv, the pair tables + row orders of its strided conv); dry = layout only
struct TableArena {
  char* base;       // null when dry
  int id;           // 0 / 1
  Bump2* bump;
};
static const vc_plan_view kAbsent = {-1, 0, 0, 0};

static int conv_tables(const vc_plan_desc* d, const PlanState& S, const Conv& C, const vc_plan_conv& g, const int32_t* in_coords,
                       const vc_plan_view& in_view, char* arena_a, TableArena& A, bool dry, vc_plan_table_out& T, hipStream_t st, int phase,
                       Bump2* pf_zone = nullptr) {
  // phase 0: what a forward pass reads (pair tables, forward row order); phase 1: the rest (backward row order).  Both phases walk
  // the same allocations, so that the offsets agree.
  const int64_t n_in = C.n_in, n_out = C.n_out;
  T = vc_plan_table_out{};
  T.present = 1;
  T.kv = C.kv;
  T.n_in = n_in;
  T.n_out = n_out;
  for (int a = 0; a < 3; ++a) T.out_shape[a] = C.out_shape[a];
  T.rep = T.grp_plan = T.order_fwd = T.order_bwd = kAbsent;
  T.in_indices = in_view;
  T.out_indices = vc_plan_view{0, 4, C.out_idx_off, n_out};
  const bool pf_prefilled = pf_zone != nullptr;
  const int64_t pf = pf_zone ? pf_zone->take((size_t)std::max<int64_t>(C.kv * n_out, 1) * 4)
                             : A.bump->take((size_t)std::max<int64_t>(C.kv * n_out, 1) * 4);
  const int64_t pb = A.bump->take((size_t)std::max<int64_t>(C.kv * n_in, 1) * 4);
  T.pair_fwd = vc_plan_view{A.id, (int32_t)n_out, pf, C.kv};
  T.pair_bwd = vc_plan_view{A.id, (int32_t)n_in, pb, C.kv};
  const bool want_bwd_order = d->need_grad && C.kv > 8 && C.kv <= 32;
  const bool want_fwd_order = d->row_order_fwd && C.kv > 8 && C.kv <= 32;
  int64_t ob = -1, of = -1;
  if (want_bwd_order) { ob = A.bump->take((size_t)std::max<int64_t>(n_in, 1) * 4); T.order_bwd = vc_plan_view{A.id, 1, ob, n_in}; }
  if (want_fwd_order) { of = A.bump->take((size_t)std::max<int64_t>(n_out, 1) * 4); T.order_fwd = vc_plan_view{A.id, 1, of, n_out}; }
  if (dry) return VC_OK;
  int rc = VC_OK;
  if (phase == 0) {
    // (the forward pair tables of a plan sit in one zone that finish_impl fills with -1 in one go: pf_zone)
    SpSkipClear skip_fill(pf_prefilled);
    rc = vc_spconv_pairs(in_coords, n_in, 3, d->batch_size, C.out_shape, g.ksize, g.stride, g.padding, g.dilation, arena_a + C.ws_off,
                         C.ws_bytes, n_out, (int32_t*)(A.base + pf), (int32_t*)(A.base + pb), st);
    if (rc != VC_OK) return rc;
  }
  if (phase == 1 && want_bwd_order && n_in > 0) {
    const int nc = g.stride[0] * g.stride[1] * g.stride[2];
    if (g_plan_parity_order && nc <= 16 && g.dilation[0] == 1 && g.dilation[1] == 1 && g.dilation[2] == 1) {
      hipLaunchKernelGGL((parity_order_kernel<512>), dim3((unsigned)cdiv(n_in, 2048)), dim3(512), 0, st, in_coords, n_in, g.stride[0],
                         g.stride[1], g.stride[2], g.padding[0], g.padding[1], g.padding[2], (int32_t*)(A.base + ob));
      VC_CHECK_LAUNCH("parity_order_kernel");
    } else {
      rc = vc_row_order((const int32_t*)(A.base + pb), n_in, C.kv, nullptr, -1, 2048, (int32_t*)(A.base + ob), st);
      if (rc != VC_OK) return rc;
    }
  }
  if (phase == 0 && want_fwd_order && n_out > 0) {
    rc = vc_row_order((const int32_t*)(A.base + pf), n_out, C.kv, nullptr, -1, 2048, (int32_t*)(A.base + of), st);
    if (rc != VC_OK) return rc;
  }
  (void)S;
  return VC_OK;
}

// pair[k, i] / rep[i] of EVERY block's pixel tensor in one launch: image_rulebook_kernel per stage of Uv2dArgs (64 rows x 4 offset
// groups per block of the fused grid)
__global__ void __launch_bounds__(256) image_rulebook_multi_kernel(Uv2dArgs a) {
  int s = 0;
#pragma unroll
  for (int t = 1; t < 8; ++t)
    if (t < a.n_stages && blockIdx.x >= a.st[t].block0_rule) s = t;
  const Uv2dStage& S = a.st[s];
  const int64_t i = (int64_t)(blockIdx.x - S.block0_rule) * 64 + (threadIdx.x & 63);
  if (i >= S.n) return;
  const int kg = threadIdx.x >> 6;
  const int kv = S.ky * S.kx, centre = (S.ky / 2) * S.kx + S.kx / 2;
  const int b = S.uv[i * 3], u = S.uv[i * 3 + 1], v = S.uv[i * 3 + 2];
  const bool inside = b >= 0 && b < a.B && u >= 0 && u < S.U && v >= 0 && v < S.V;
  for (int k = kg; k < kv; k += 4) {
    int r = -1;
    if (k == centre) {
      r = (int)i;
      S.rep[i] = inside ? S.img[((int64_t)b * S.U + u) * S.V + v] - 1 : (int)i;
    } else {
      const int nu = u + (k / S.kx - S.ky / 2) * S.dy, nv = v + (k % S.kx - S.kx / 2) * S.dx;
      if (b >= 0 && b < a.B && nu >= 0 && nu < S.SH && nv >= 0 && nv < S.SW && nu < S.U && nv < S.V) r = S.img[((int64_t)b * S.U + nu) * S.V + nv] - 1;
    }
    S.pair[(int64_t)k * S.n + i] = r;
  }
}

// The 3-D SubM table of block b (coordinate -> row by the bitmap of the strided conv that produced the rows, or by a hash);
// `launch` = false: layout only
static int block_tables_3d(const vc_plan_desc* d, const PlanState& S, int b, const int32_t* coords, const vc_plan_view& coords_view,
                           char* arena_a, TableArena& A, bool launch, vc_plan_block_out& O, hipStream_t st) {
  const vc_plan_block& B = d->blocks[b];
  const Blk& K = S.blk[b];
  const int64_t n = K.n;
  const int kv3 = B.subm_ksize[0] * B.subm_ksize[1] * B.subm_ksize[2];
  vc_plan_table_out& T3 = O.subm3d;
  T3 = vc_plan_table_out{};
  T3.present = 1;
  T3.kv = kv3;
  T3.n_in = T3.n_out = n;
  for (int a = 0; a < 3; ++a) T3.out_shape[a] = K.shape[a];
  T3.pair_bwd = T3.rep = T3.order_fwd = T3.order_bwd = T3.grp_plan = kAbsent;
  T3.in_indices = T3.out_indices = coords_view;
  const int64_t p3 = A.bump->take((size_t)std::max<int64_t>(kv3 * n, 1) * 4);
  T3.pair_fwd = vc_plan_view{A.id, (int32_t)n, p3, kv3};
  const bool by_bitmap = g_plan_subm_bitmap && K.down.present;
  const size_t hbytes = by_bitmap ? 0 : vc_hash_workspace_bytes(n);
  const int64_t hoff = by_bitmap ? -1 : A.bump->take(hbytes);
  if (!launch || n == 0) return VC_OK;
  int32_t* pair3 = (int32_t*)(A.base + p3);
  if (by_bitmap) {
    const unsigned long long* bitmap = (const unsigned long long*)(arena_a + K.down.ws_off);
    const int64_t nwords = cdiv((int64_t)d->batch_size * K.shape[0] * K.shape[1] * K.shape[2], 64);
    const uint32_t* prefix = (const uint32_t*)(bitmap + (nwords < 1 ? 1 : nwords));
    hipLaunchKernelGGL(subm_bitmap_rulebook_kernel, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, st, coords, n, K.shape[0], K.shape[1],
                       K.shape[2], B.subm_ksize[0], B.subm_ksize[1], B.subm_ksize[2], B.subm_dilation[0], B.subm_dilation[1],
                       B.subm_dilation[2], bitmap, prefix, pair3);
    VC_CHECK_LAUNCH("subm_bitmap_rulebook_kernel");
  } else {
    int rc = vc_hash_build(coords, n, 3, K.shape, A.base + hoff, hbytes, st);
    if (rc != VC_OK) return rc;
    rc = vc_subm_rulebook(coords, n, 3, K.shape, B.subm_ksize, B.subm_dilation, A.base + hoff, hbytes, pair3, nullptr, st);
    if (rc != VC_OK) return rc;
  }
  return VC_OK;
}

// The image-space branch of EVERY block (pixel coordinates, 2-D SubM pair tables, representatives, group plans), all in arena B:
//   phase 2: the projection (the plan's only floating-point kernel) + pixel marking of all blocks in ONE launch, the pair tables in
//            one more, behind one clear of all pixel images -- enqueued by vc_plan_finish BEHIND the caller's event (LOG.md A.15 /
//            A.17: this kernel, and only this one, computes wrong pixels beside the conv kernels);
//   phase 1: the group plans (sorts) -- vc_plan_finish_backward;    any other phase: layout only.
// coords[b] / n of each block come from the caller's walk.
static int tables_2d(const vc_plan_desc* d, const PlanState& S, const int32_t* const* coords, char* arena_a, TableArena& A, bool dry,
                     vc_plan_out& O, hipStream_t st, int phase) {
  struct L { int64_t uvo, p2, repo, gpo, gwo, h2o, imgo; size_t gw_bytes, h2_bytes, img_bytes; int kv2, U, V; };
  L lay[VC_PLAN_MAX_BLOCKS] = {};
  bool any = false;
  for (int b = 0; b < d->n_blocks; ++b) {
    const vc_plan_block& B = d->blocks[b];
    vc_plan_block_out& BO = O.blocks[b];
    BO.uv = kAbsent;
    BO.subm2d = vc_plan_table_out{};
    if (!B.has_2d) continue;
    any = true;
    const int64_t n = S.blk[b].n;
    L& l = lay[b];
    l.kv2 = B.ksize2d[0] * B.ksize2d[1];
    l.uvo = A.bump->take((size_t)std::max<int64_t>(n, 1) * 12);
    l.p2 = A.bump->take((size_t)std::max<int64_t>(l.kv2 * n, 1) * 4);
    l.repo = A.bump->take((size_t)std::max<int64_t>(n, 1) * 4);
    BO.uv = vc_plan_view{A.id, 3, l.uvo, n};
    vc_plan_table_out& T2 = BO.subm2d;
    T2.present = 1;
    T2.kv = l.kv2;
    T2.n_in = T2.n_out = n;
    T2.out_shape[0] = d->image_shape[0]; T2.out_shape[1] = d->image_shape[1]; T2.out_shape[2] = 0;
    T2.pair_bwd = T2.order_fwd = T2.order_bwd = T2.grp_plan = kAbsent;
    T2.in_indices = T2.out_indices = BO.uv;
    T2.pair_fwd = vc_plan_view{A.id, (int32_t)n, l.p2, l.kv2};
    T2.rep = vc_plan_view{A.id, 1, l.repo, n};
    // pixels are clamped to [0, 1399] x [0, 599] and divided by the stride (vc_project_uv, spconv_backbone.py:76-81)
    l.U = std::min<int>(d->image_shape[0], (1400 - 1) / B.uv_stride + 1);
    l.V = std::min<int>(d->image_shape[1], (600 - 1) / B.uv_stride + 1);
    l.h2o = l.gpo = l.gwo = l.imgo = -1;
    if (!g_plan_image_2d) {
      l.h2_bytes = vc_hash_workspace_bytes(n);
      l.h2o = A.bump->take(l.h2_bytes);
    }
    if (d->need_grad) {
      l.gpo = A.bump->take((size_t)std::max<int64_t>(2 * n, 1) * 4);
      T2.grp_plan = vc_plan_view{A.id, (int32_t)n, l.gpo, 2};
      l.gw_bytes = vc_group_plan_workspace_bytes(n);
      l.gwo = A.bump->take(l.gw_bytes);
    }
  }
  // the group plans of all blocks by one sort (common.h GroupMulti): one shared workspace
  int64_t gmo = -1;
  size_t gm_bytes = 0;
  if (any && d->need_grad && g_plan_group_multi) {
    int64_t total = 0;
    for (int b = 0; b < d->n_blocks; ++b)
      if (d->blocks[b].has_2d) total += S.blk[b].n;
    gm_bytes = group_plan_multi_workspace_bytes(total);
    gmo = A.bump->take(gm_bytes);
  }
  // one zone for the pixel images of all blocks: one clear
  int64_t zone = -1;
  size_t zone_bytes = 0;
  if (any && g_plan_image_2d) {
    for (int b = 0; b < d->n_blocks; ++b)
      if (d->blocks[b].has_2d) {
        lay[b].img_bytes = al256((size_t)d->batch_size * lay[b].U * lay[b].V * 4);
        zone_bytes += lay[b].img_bytes;
      }
    zone = A.bump->take(zone_bytes);
    size_t o = 0;
    for (int b = 0; b < d->n_blocks; ++b)
      if (d->blocks[b].has_2d) { lay[b].imgo = zone + (int64_t)o; o += lay[b].img_bytes; }
  }
  if (dry || !any) return VC_OK;
  int rc;
  if (phase == 1) {
    if (gmo >= 0) {
      GroupMulti g{};
      for (int b = 0; b < d->n_blocks; ++b)
        if (d->blocks[b].has_2d && d->need_grad && S.blk[b].n > 0 && g.n_tables < 8) {
          g.rep[g.n_tables] = (const int32_t*)(A.base + lay[b].repo);
          g.plan[g.n_tables] = (int32_t*)(A.base + lay[b].gpo);
          g.n[g.n_tables++] = S.blk[b].n;
        }
      int n2d = 0;
      for (int b = 0; b < d->n_blocks; ++b) n2d += d->blocks[b].has_2d && S.blk[b].n > 0;
      if (g.n_tables == n2d) {   // (more than 8 image-space tables: the per-table route below)
        if (g.n_tables == 0) return VC_OK;
        rc = group_plan_multi(g, A.base + gmo, gm_bytes, st);
        if (rc != VC_ECAPACITY) return rc;
      }
    }
    for (int b = 0; b < d->n_blocks; ++b)
      if (d->blocks[b].has_2d && d->need_grad && S.blk[b].n > 0) {
        rc = vc_group_plan((const int32_t*)(A.base + lay[b].repo), S.blk[b].n, (int32_t*)(A.base + lay[b].gpo), A.base + lay[b].gwo,
                           lay[b].gw_bytes, st);
        if (rc != VC_OK) return rc;
      }
    return VC_OK;
  }
  if (phase != 2) return VC_OK;
  const float* params = (const float*)(arena_a + S.params_off);
  const bool diagnostics = d->debug_buf != nullptr || g_plan_uv_mode != 0;
  if (g_plan_image_2d && !diagnostics) {
    VC_CHECK_HIP(hipMemsetAsync(A.base + zone, 0, zone_bytes, st));
    Uv2dArgs a{};
    a.B = d->batch_size;
    a.params = params;
    unsigned bm = 0, br = 0;
    for (int b = 0; b < d->n_blocks; ++b) {
      const vc_plan_block& B = d->blocks[b];
      const int64_t n = S.blk[b].n;
      if (!B.has_2d || n == 0) continue;
      Uv2dStage& T = a.st[a.n_stages++];
      const L& l = lay[b];
      T.coords = coords[b];
      T.uv = (int32_t*)(A.base + l.uvo);
      T.img = (int32_t*)(A.base + l.imgo);
      T.pair = (int32_t*)(A.base + l.p2);
      T.rep = (int32_t*)(A.base + l.repo);
      T.n = n;
      T.stride = B.uv_stride; T.U = l.U; T.V = l.V; T.SH = d->image_shape[0]; T.SW = d->image_shape[1];
      T.ky = B.ksize2d[0]; T.kx = B.ksize2d[1]; T.dy = B.dilation2d[0]; T.dx = B.dilation2d[1];
      // hard-coded range / voxel size of the reference (spconv_backbone.py:8): python floats (fp64) rounded to fp32 on use
      const double vs = 0.05 * B.uv_stride;
      T.vs = (float)vs; T.minx = (float)(0.0 + vs / 2); T.miny = (float)(-40.0 + vs / 2); T.minz = (float)(-3.0 + vs / 2);
      T.block0_mark = bm; T.block0_rule = br;
      bm += (unsigned)cdiv(n, 256); br += (unsigned)cdiv(n, 64);
    }
    if (a.n_stages == 0) return VC_OK;
    if (g_plan_uv_poison)
      for (int t = 0; t < a.n_stages; ++t) VC_CHECK_HIP(hipMemsetAsync(a.st[t].uv, 0x7F, (size_t)a.st[t].n * 12, st));
    rc = uv_mark_multi(a, bm, st);
    if (rc != VC_OK) return rc;
    hipLaunchKernelGGL(image_rulebook_multi_kernel, dim3(br), dim3(256), 0, st, a);
    VC_CHECK_LAUNCH("image_rulebook_multi_kernel");
    return VC_OK;
  }
  // block by block: the coordinate-hash form (vc_debug_set plan_image_2d = 0: A/B) and the diagnostics forms of the projection
  for (int b = 0; b < d->n_blocks; ++b) {
    const vc_plan_block& B = d->blocks[b];
    const int64_t n = S.blk[b].n;
    if (!B.has_2d || n == 0) continue;
    const L& l = lay[b];
    int32_t* uv = (int32_t*)(A.base + l.uvo);
    if (g_plan_reprepare) {   // diagnostics: the parameter block written again right in front of its reader
      rc = vc_project_prepare(d->calib, d->trans, d->batch_size, (float*)(arena_a + S.params_off), st);
      if (rc != VC_OK) return rc;
    }
    if (diagnostics)
      rc = project_uv_debug(coords[b], n, params, d->batch_size, B.uv_stride, uv, (int32_t*)d->debug_buf, d->debug_buf ? 4096 : 0,
                            g_plan_uv_mode, d->trans != nullptr, st);
    else
      rc = vc_project_uv(coords[b], n, params, d->batch_size, B.uv_stride, uv, nullptr, st);
    if (rc != VC_OK) return rc;
    int32_t* pair2 = (int32_t*)(A.base + l.p2);
    int32_t* rep = (int32_t*)(A.base + l.repo);
    if (g_plan_image_2d) {
      int32_t* img = (int32_t*)(A.base + l.imgo);
      VC_CHECK_HIP(hipMemsetAsync(img, 0, l.img_bytes, st));
      hipLaunchKernelGGL(image_mark_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const int32_t*)uv, n, d->batch_size, l.U, l.V, img);
      VC_CHECK_LAUNCH("image_mark_kernel");
      hipLaunchKernelGGL(image_rulebook_kernel, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, st, (const int32_t*)uv, n, d->batch_size, l.U, l.V,
                         d->image_shape[0], d->image_shape[1], B.ksize2d[0], B.ksize2d[1], B.dilation2d[0], B.dilation2d[1],
                         (const int32_t*)img, pair2, rep);
      VC_CHECK_LAUNCH("image_rulebook_kernel");
    } else {
      rc = vc_hash_build(uv, n, 2, d->image_shape, A.base + l.h2o, l.h2_bytes, st);
      if (rc != VC_OK) return rc;
      rc = vc_subm_rulebook(uv, n, 2, d->image_shape, B.ksize2d, B.dilation2d, A.base + l.h2o, l.h2_bytes, pair2, rep, st);
      if (rc != VC_OK) return rc;
    }
  }
  return VC_OK;
}

// Events of the counts' trip to the host: a process-wide ring (a plan may be begun on one host thread and waited for on another:
// autograd worker, loader thread), each slot stamped with the generation of the plan that last recorded it.  vc_plan_wait
// refuses a slot whose stamp moved on (more than kPlanEvents plans between their begin and their wait) instead of reading counts
// that belong to another plan.
static constexpr int kPlanEvents = 64;
static constexpr int kPlanDevices = 16;
// One ring PER DEVICE, created on first use of that device and never destroyed (round 6, ADVICE r5: the single ring of round 5 destroyed its
// events when another device became current -- a process driving two GPUs, or a plan begun on one device and waited for while another is
// current, lost in-flight plans with a misleading error).  S.ev_slot = device * kPlanEvents + slot.
static std::mutex g_plan_ev_mutex;
static hipEvent_t g_plan_ev[kPlanDevices][kPlanEvents] = {};
static uint64_t g_plan_ev_gen[kPlanDevices][kPlanEvents] = {};
static uint64_t g_plan_ev_next[kPlanDevices] = {};
static bool g_plan_ev_made[kPlanDevices] = {};

// -> slot (>= 0) with S.ev_gen set, the event recorded on `st`; < 0: HIP error
static int plan_event_record(PlanState& S, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_plan_ev_mutex);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kPlanDevices) return -1;
  if (!g_plan_ev_made[dev]) {   // events belong to the device they were created on
    for (int i = 0; i < kPlanEvents; ++i)
      if (hipEventCreateWithFlags(&g_plan_ev[dev][i], hipEventDisableTiming) != hipSuccess) return -1;
    g_plan_ev_made[dev] = true;
  }
  const uint64_t gen = ++g_plan_ev_next[dev];
  const int slot = (int)(gen % kPlanEvents);
  if (hipEventRecord(g_plan_ev[dev][slot], st) != hipSuccess) return -1;
  g_plan_ev_gen[dev][slot] = gen;
  S.ev_slot = dev * kPlanEvents + slot;
  S.ev_gen = gen;
  return S.ev_slot;
}
// the plan's event, or nullptr when its slot has been recorded again by a later plan of the same device
static hipEvent_t plan_event_of(const PlanState& S) {
  std::lock_guard<std::mutex> lock(g_plan_ev_mutex);
  if (S.ev_slot < 0 || S.ev_slot >= kPlanDevices * kPlanEvents) return nullptr;
  const int dev = S.ev_slot / kPlanEvents, slot = S.ev_slot % kPlanEvents;
  if (!g_plan_ev_made[dev] || g_plan_ev_gen[dev][slot] != S.ev_gen) return nullptr;
  return g_plan_ev[dev][slot];
}

}  // namespace vc

using namespace vc;

extern "C" {

size_t vc_plan_begin_arena_bytes(const vc_plan_desc* d) {
  if (check_desc(d) != VC_OK) return 0;
  PlanState S{};
  Bump2 bump;
  layout_chain(d, S, bump);
  TableArena A{nullptr, 0, &bump};
  for (int b = 0; b < d->n_blocks; ++b)
    if (S.blk[b].early) {
      vc_plan_block_out O{};
      if (block_tables_3d(d, S, b, nullptr, kAbsent, nullptr, A, false, O, nullptr) != VC_OK) return 0;
    }
  return bump.off + 256;
}

int vc_plan_begin(const vc_plan_desc* d, void* arena_a, size_t arena_a_bytes, int32_t* host_counts, vc_plan_state* state,
                  void* stream) {
  int rc = check_desc(d);
  if (rc != VC_OK) return rc;
  VC_REQUIRE(arena_a && host_counts && state, "vc_plan_begin: null argument");
  PlanState& S = *reinterpret_cast<PlanState*>(state);
  S = PlanState{};
  Bump2 bump;
  layout_chain(d, S, bump);
  VC_REQUIRE(S.n_counts <= kMaxCounts, "vc_plan_begin: too many row counts");
  const size_t need = vc_plan_begin_arena_bytes(d);
  if (need == 0 || arena_a_bytes < need) { set_error("vc_plan_begin: arena_a too small (%zu < %zu)", arena_a_bytes, need); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  char* A = (char*)arena_a;
  int32_t* counts = (int32_t*)(A + S.counts_off);
  S.host_counts = host_counts;
  const double keep_frac = 1.0 - d->discard_rate;

  if (d->calib) {
    rc = vc_project_prepare(d->calib, d->trans, d->batch_size, (float*)(A + S.params_off), st);
    if (rc != VC_OK) return rc;
    if (d->debug_buf) {   // diagnostics: [2] = the plan has augmentation parameters (the caller zeroed the buffer)
      VC_REQUIRE(d->debug_bytes >= 256 + 128 * 4096, "vc_plan_begin: debug_buf too small");
      hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(64), 0, st, (int32_t*)d->debug_buf + 2, (int32_t)(d->trans != nullptr));
      VC_CHECK_LAUNCH("set_count_kernel");
    }
  }
  // ---- the chain: coordinates, keeps and counts of every level, nothing read back yet
  const int32_t* cur = d->indices;      // coordinates feeding the next strided conv / block
  int64_t cur_n = d->n;                 // their row count when known on the host (cur_cnt < 0)
  int cur_cnt = -1;                     // ... or its slot in counts[]
  int64_t cur_cap = d->n;
  if (d->input_discard) {
    int64_t* keep = (int64_t*)(A + S.in_keep_off);
    if (d->input_keep) {
      VC_REQUIRE(d->input_keep_rows == (int64_t)((double)d->n * keep_frac), "vc_plan_begin: injected input keep has %lld rows, expected int(%lld * (1 - rate))",
                 (long long)d->input_keep_rows, (long long)d->n);
      VC_CHECK_HIP(hipMemcpyAsync(keep, d->input_keep, (size_t)d->input_keep_rows * 8, hipMemcpyDeviceToDevice, st));
    } else {
      rc = vc_random_keep(d->n, S.n_in_keep, d->input_keep_seed, keep, st);
      if (rc != VC_OK) return rc;
    }
    if (S.n_in_keep > 0) {
      hipLaunchKernelGGL(gather_coords_dev_kernel, dim3((unsigned)cdiv(S.n_in_keep, 256)), dim3(256), 0, st, (const int4*)cur, (const int64_t*)keep,
                         (const int32_t*)nullptr, S.n_in_keep, (int4*)(A + S.in_kept_off));
      VC_CHECK_LAUNCH("gather_coords_dev_kernel");
    }
    cur = (const int32_t*)(A + S.in_kept_off);
    cur_n = cur_cap = S.n_in_keep;
  }
  if (S.ws_zone_bytes > 0) VC_CHECK_HIP(hipMemsetAsync(A + S.ws_zone_off, 0, S.ws_zone_bytes, st));   // every conv's bitmap, once
  auto run_conv = [&](Conv& C, const vc_plan_conv& g) -> int {
    int r;
    SpSkipClear skip_clear;   // (the zone is clear: vc_spconv_mark_count* skip their own fill)
    if (cur_cnt < 0)
      r = vc_spconv_mark_count(cur, cur_n, 3, d->batch_size, C.out_shape, g.ksize, g.stride, g.padding, g.dilation, A + C.ws_off, C.ws_bytes,
                               counts + C.cnt_out, st);
    else
      r = vc_spconv_mark_count_dev(cur, cur_cap, counts + cur_cnt, 3, d->batch_size, C.out_shape, g.ksize, g.stride, g.padding, g.dilation,
                                   A + C.ws_off, C.ws_bytes, counts + C.cnt_out, st);
    if (r != VC_OK) return r;
    r = vc_spconv_emit_indices(3, d->batch_size, C.out_shape, A + C.ws_off, C.ws_bytes, C.cap_out, (int32_t*)(A + C.out_idx_off), st);
    if (r != VC_OK) return r;
    if (cur_cnt < 0) C.n_in = cur_n;
    cur = (const int32_t*)(A + C.out_idx_off);
    cur_cnt = C.cnt_out;
    cur_n = -1;
    cur_cap = C.cap_out;
    return VC_OK;
  };
  for (int b = 0; b < d->n_blocks; ++b) {
    const vc_plan_block& B = d->blocks[b];
    Blk& K = S.blk[b];
    if (B.has_down) {
      rc = run_conv(K.down, B.down);
      if (rc != VC_OK) return rc;
    }
    if (B.discard) {
      int64_t* keep = (int64_t*)(A + K.keep_off);
      if (cur_cnt < 0) {   // row count known: plain launches
        if (B.keep) {
          VC_REQUIRE(B.keep_rows == K.n_keep, "vc_plan_begin: block %d: injected keep has %lld rows, expected %lld", b, (long long)B.keep_rows,
                     (long long)K.n_keep);
          VC_CHECK_HIP(hipMemcpyAsync(keep, B.keep, (size_t)B.keep_rows * 8, hipMemcpyDeviceToDevice, st));
        } else {
          rc = vc_random_keep(cur_n, K.n_keep, B.keep_seed, keep, st);
          if (rc != VC_OK) return rc;
        }
        if (K.n_keep > 0) {
          hipLaunchKernelGGL(gather_coords_dev_kernel, dim3((unsigned)cdiv(K.n_keep, 256)), dim3(256), 0, st, (const int4*)cur,
                             (const int64_t*)keep, (const int32_t*)nullptr, K.n_keep, (int4*)(A + K.kept_off));
          VC_CHECK_LAUNCH("gather_coords_dev_kernel");
        }
        cur_n = cur_cap = K.n_keep;
      } else {             // row count on the device
        if (B.keep) {
          VC_CHECK_HIP(hipMemcpyAsync(keep, B.keep, (size_t)B.keep_rows * 8, hipMemcpyDeviceToDevice, st));
          // the row count the injected keep implies; checked against int(n * (1 - rate)) in vc_plan_wait
          hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(64), 0, st, counts + K.cnt_keep, (int32_t)B.keep_rows);
          VC_CHECK_LAUNCH("set_count_kernel");
        } else {
          hipLaunchKernelGGL(keep_count_kernel, dim3(1), dim3(64), 0, st, (const int32_t*)(counts + cur_cnt), keep_frac, counts + K.cnt_keep);
          VC_CHECK_LAUNCH("keep_count_kernel");
          if (K.cap_keep > 0) {
            hipLaunchKernelGGL(random_keep_dev_kernel, dim3((unsigned)std::min<int64_t>(cdiv(K.cap_keep, 256), 2048)), dim3(256), 0, st, (const int32_t*)(counts + cur_cnt),
                               (const int32_t*)(counts + K.cnt_keep), B.keep_seed, keep);
            VC_CHECK_LAUNCH("random_keep_dev_kernel");
          }
        }
        if (K.cap_keep > 0) {
          hipLaunchKernelGGL(gather_coords_dev_kernel, dim3((unsigned)std::min<int64_t>(cdiv(K.cap_keep, 256), 2048)), dim3(256), 0, st,
                             (const int4*)cur, (const int64_t*)keep, (const int32_t*)(counts + K.cnt_keep), (int64_t)0,
                             (int4*)(A + K.kept_off));
          VC_CHECK_LAUNCH("gather_coords_dev_kernel");
        }
        cur_cnt = K.cnt_keep;
        cur_cap = K.cap_keep;
      }
      cur = (const int32_t*)(A + K.kept_off);
    }
  }
  if (d->has_tail) {
    rc = run_conv(S.tail, d->tail);
    if (rc != VC_OK) return rc;
  }
  // ---- the counts' trip to the host starts here; the tables below run underneath it
  if (S.n_counts > 0) VC_CHECK_HIP(hipMemcpyAsync(host_counts, counts, (size_t)S.n_counts * 4, hipMemcpyDeviceToHost, st));
  VC_REQUIRE(plan_event_record(S, st) >= 0, "vc_plan_begin: cannot create / record the counts' event");
  // ---- the 3-D SubM tables of the blocks whose row count is already known (the first block of VirConvL8x): integer kernels, under the
  // counts' trip.  Every image-space table (the projection is the plan's only floating-point kernel) belongs to vc_plan_finish.
  S.early_out = vc_plan_out{};
  TableArena TA{A, 0, &bump};
  for (int b = 0; b < d->n_blocks; ++b) {
    Blk& K = S.blk[b];
    if (!K.early) continue;
    const int32_t* coords = K.coords_off < 0 ? d->indices : (const int32_t*)(A + K.coords_off);
    const vc_plan_view cv = K.coords_off < 0 ? kAbsent : vc_plan_view{0, 4, K.coords_off, K.n};
    rc = block_tables_3d(d, S, b, coords, cv, A, TA, true, S.early_out.blocks[b], st);
    if (rc != VC_OK) return rc;
  }
  S.magic = kPlanMagic;
  return VC_OK;
}

int vc_plan_wait(const vc_plan_desc* d, vc_plan_state* state) {
  VC_REQUIRE(d && state, "vc_plan_wait: null argument");
  PlanState& S = *reinterpret_cast<PlanState*>(state);
  VC_REQUIRE(S.magic == kPlanMagic, "vc_plan_wait: state was not written by vc_plan_begin");
  hipEvent_t ev = plan_event_of(S);
  VC_REQUIRE(ev != nullptr, "vc_plan_wait: this plan's event slot was recorded again by a later plan (more than %d plans between "
             "vc_plan_begin and vc_plan_wait): its counts cannot be trusted", kPlanEvents);
  // poll (a blocking hipEventSynchronize parks the thread in the kernel driver: slower by the wake-up latency, and much slower with
  // an RCCL communicator alive in the process -- DESIGN.md 5)
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) { set_error("vc_plan_wait: %s", hipGetErrorString(e)); return VC_EHIP; }
  }
  // the slot may have been recorded again WHILE this thread polled it (another thread began 64 plans): what completed is then a later plan's
  // event, not this one's
  VC_REQUIRE(plan_event_of(S) == ev, "vc_plan_wait: this plan's event slot was recorded again by a later plan while it was being waited for");
  const int32_t* hc = S.host_counts;
  const double keep_frac = 1.0 - d->discard_rate;
  int64_t cur_n = d->input_discard ? S.n_in_keep : d->n;
  for (int b = 0; b < d->n_blocks; ++b) {
    const vc_plan_block& B = d->blocks[b];
    Blk& K = S.blk[b];
    if (B.has_down) {
      K.down.n_in = cur_n;
      K.down.n_out = hc[K.down.cnt_out];
      VC_REQUIRE(K.down.n_out >= 0 && K.down.n_out <= K.down.cap_out, "vc_plan_wait: block %d: strided conv emits %lld rows, capacity %lld", b,
                 (long long)K.down.n_out, (long long)K.down.cap_out);
      cur_n = K.down.n_out;
    }
    K.n = cur_n;
    if (B.discard) {
      const int64_t want = (int64_t)((double)cur_n * keep_frac);
      if (K.cnt_keep >= 0) K.n_keep = hc[K.cnt_keep];
      VC_REQUIRE(K.n_keep == want, "vc_plan_wait: block %d keeps %lld rows, expected int(%lld * (1 - rate)) = %lld", b, (long long)K.n_keep,
                 (long long)cur_n, (long long)want);
      VC_REQUIRE(K.n_keep <= K.cap_keep, "vc_plan_wait: block %d keeps %lld rows, capacity %lld", b, (long long)K.n_keep, (long long)K.cap_keep);
      cur_n = K.n_keep;
    }
  }
  if (d->has_tail) {
    S.tail.n_in = cur_n;
    S.tail.n_out = hc[S.tail.cnt_out];
    VC_REQUIRE(S.tail.n_out >= 0 && S.tail.n_out <= S.tail.cap_out, "vc_plan_wait: the tail conv emits %lld rows, capacity %lld",
               (long long)S.tail.n_out, (long long)S.tail.cap_out);
  }
  S.waited = 1;
  return VC_OK;
}

static int finish_impl(const vc_plan_desc* d, PlanState& S, char* arena_a, char* arena_b, size_t arena_b_bytes, vc_plan_out* out, bool dry,
                       size_t* need, hipStream_t st, int phase) {
  Bump2 bump;
  TableArena TB{arena_b, 1, &bump};
  vc_plan_out O = S.early_out;
  O.input_keep = O.input_kept_indices = kAbsent;
  O.n_input_kept = d->input_discard ? S.n_in_keep : 0;
  if (d->input_discard) {
    O.input_keep = vc_plan_view{0, 1, S.in_keep_off, S.n_in_keep};
    O.input_kept_indices = vc_plan_view{0, 4, S.in_kept_off, S.n_in_keep};
  }
  const int32_t* cur = d->input_discard ? (const int32_t*)(arena_a + S.in_kept_off) : d->indices;
  vc_plan_view cur_view = d->input_discard ? O.input_kept_indices : kAbsent;
  const int32_t* coords2d[VC_PLAN_MAX_BLOCKS] = {};   // each block's coordinates, for the image-space branch below
  // the forward pair tables of all strided convs: one zone at the head of arena B, filled with -1 by ONE launch
  size_t pf_total = 0;
  for (int b = 0; b < d->n_blocks; ++b)
    if (d->blocks[b].has_down) pf_total += al256((size_t)std::max<int64_t>(S.blk[b].down.kv * S.blk[b].down.n_out, 1) * 4);
  if (d->has_tail) pf_total += al256((size_t)std::max<int64_t>(S.tail.kv * S.tail.n_out, 1) * 4);
  Bump2 pfz;
  pfz.off = (size_t)bump.take(pf_total);
  if (!dry && phase == 0 && pf_total > 0) VC_CHECK_HIP(hipMemsetAsync(arena_b + pfz.off, 0xFF, pf_total, st));
  for (int b = 0; b < d->n_blocks; ++b) {
    const vc_plan_block& B = d->blocks[b];
    Blk& K = S.blk[b];
    vc_plan_block_out& BO = O.blocks[b];
    int rc;
    if (B.has_down) {
      rc = conv_tables(d, S, K.down, B.down, cur, cur_view, arena_a, TB, dry, BO.down, st, phase, &pfz);
      if (rc != VC_OK) return rc;
      cur = (const int32_t*)(arena_a + K.down.out_idx_off);
      cur_view = vc_plan_view{0, 4, K.down.out_idx_off, K.down.n_out};
    } else {
      BO.down = vc_plan_table_out{};
    }
    if (!K.early) {
      rc = block_tables_3d(d, S, b, cur, cur_view, arena_a, TB, !dry && phase == 0, BO, st);
      if (rc != VC_OK) return rc;
    }
    coords2d[b] = cur;
    BO.n = K.n;
    BO.n_keep = B.discard ? K.n_keep : 0;
    BO.keep = BO.kept_indices = kAbsent;
    if (B.discard) {
      BO.keep = vc_plan_view{0, 1, K.keep_off, K.n_keep};
      BO.kept_indices = vc_plan_view{0, 4, K.kept_off, K.n_keep};
      cur = (const int32_t*)(arena_a + K.kept_off);
      cur_view = BO.kept_indices;
    }
  }
  O.tail = vc_plan_table_out{};
  if (d->has_tail) {
    const int rc = conv_tables(d, S, S.tail, d->tail, cur, cur_view, arena_a, TB, dry, O.tail, st, phase, &pfz);
    if (rc != VC_OK) return rc;
  }
  {
    const int rc = tables_2d(d, S, coords2d, arena_a, TB, dry, O, st, phase);
    if (rc != VC_OK) return rc;
  }
  if (need) *need = bump.off + 256;
  if (!dry) {
    if (bump.off > arena_b_bytes) { set_error("vc_plan_finish: arena_b too small"); return VC_ECAPACITY; }
    if (out) *out = O;
  }
  return VC_OK;
}

size_t vc_plan_finish_arena_bytes(const vc_plan_desc* d, const vc_plan_state* state) {
  if (check_desc(d) != VC_OK || !state) return 0;
  PlanState S = *reinterpret_cast<const PlanState*>(state);
  if (S.magic != kPlanMagic || !S.waited) { set_error("vc_plan_finish_arena_bytes: call vc_plan_wait first"); return 0; }
  size_t need = 0;
  if (finish_impl(d, S, nullptr, nullptr, 0, nullptr, true, &need, nullptr, 0) != VC_OK) return 0;
  return need;
}

int vc_plan_finish(const vc_plan_desc* d, vc_plan_state* state, void* arena_a, void* arena_b, size_t arena_b_bytes, vc_plan_out* out,
                   void* stream) {
  int rc = check_desc(d);
  if (rc != VC_OK) return rc;
  VC_REQUIRE(state && arena_a && arena_b && out, "vc_plan_finish: null argument");
  PlanState& S = *reinterpret_cast<PlanState*>(state);
  {
    // LOG.md A.17: the pixel projection computes wrong values in lanes 48-63 of some waves when its waves share a compute unit with
    // the bf16-split gather-GEMM / weight-gradient waves (round 6: disjoint CU masks or a projection block that owns its CU's LDS
    // remove it, profiles/r06_a17_cu_mask.md).  An unfenced image-space branch must be asked for, not got by leaving a field zero.
    bool has2d = false;
    for (int b = 0; b < d->n_blocks; ++b) has2d = has2d || d->blocks[b].has_2d != 0;
    if (has2d && !d->tables_wait_event && !d->allow_unfenced_projection) {
      set_error("vc_plan_finish: this plan has an image-space branch (pixel projection) and tables_wait_event is NULL: record an event on the "
                "stream of the feature passes behind the previous pass and name it here, or set allow_unfenced_projection = 1 if no conv "
                "kernel of this library can be running while the plan's tables are built (LOG.md A.17)");
      return VC_EINVAL;
    }
  }
  VC_REQUIRE(S.magic == kPlanMagic && S.waited, "vc_plan_finish: call vc_plan_begin and vc_plan_wait first");
  size_t need = 0;
  {
    PlanState T = S;
    rc = finish_impl(d, T, nullptr, nullptr, 0, nullptr, true, &need, nullptr, 0);
    if (rc != VC_OK) return rc;
  }
  if (arena_b_bytes < need) { set_error("vc_plan_finish: arena_b too small (%zu < %zu)", arena_b_bytes, need); return VC_ECAPACITY; }
  // the integer tables (pair tables of the strided convs, 3-D SubM tables) right away: they may run beside anything
  rc = finish_impl(d, S, (char*)arena_a, (char*)arena_b, arena_b_bytes, out, false, nullptr, (hipStream_t)stream, 0);
  if (rc != VC_OK) return rc;
  // ... the image-space branch -- the projection, the plan's only floating-point kernel -- behind the caller's event: beside the conv
  // kernels of a running feature pass it computes wrong pixels in lanes 48-63 of some waves (LOG.md A.15 / A.17)
  bool any2d = false;
  for (int b = 0; b < d->n_blocks; ++b) any2d = any2d || d->blocks[b].has_2d != 0;
  // (checked before anything is enqueued: see the top of this function)
  if (d->tables_wait_event && any2d) VC_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)d->tables_wait_event, 0));
  if (any2d) {
    PlanState T = S;
    rc = finish_impl(d, T, (char*)arena_a, (char*)arena_b, arena_b_bytes, nullptr, false, nullptr, (hipStream_t)stream, 2);
  }
  if (rc == VC_OK) S.finished = 1;
  return rc;
}

int vc_plan_finish_backward(const vc_plan_desc* d, vc_plan_state* state, void* arena_a, void* arena_b, size_t arena_b_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc != VC_OK) return rc;
  VC_REQUIRE(state && arena_a && arena_b, "vc_plan_finish_backward: null argument");
  PlanState& S = *reinterpret_cast<PlanState*>(state);
  VC_REQUIRE(S.magic == kPlanMagic && S.finished, "vc_plan_finish_backward: call vc_plan_finish first");
  if (!d->need_grad) return VC_OK;
  hipStream_t st = (hipStream_t)stream;
  PlanState T = S;
  return finish_impl(d, T, (char*)arena_a, (char*)arena_b, arena_b_bytes, nullptr, false, nullptr, st, 1);
}

}  // extern "C"
