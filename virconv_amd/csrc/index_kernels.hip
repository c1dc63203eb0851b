// Index-side kernels of the VirConv hot path for gfx950 (HBM/L2-bound integer work; no MFMA here):
//   K3 coordinate hash, K4 submanifold rulebook, K5 strided rulebook (bitmap + popcount scan),
//   K9 voxel->pixel projection, K2 row gather/scatter (layer discard), K10 dense scatter/gather,
//   K1 first-touch voxelisation with fused MeanVFE.
// Compiled with -ffp-contract=off: the projection and voxeliser float math must round once per operation so that
// integer outputs are bit-identical to oracle/geometry.py.
#include <stdarg.h>

#include "common.h"

namespace vc {

static thread_local char g_err[512] = "";
thread_local bool t_sp_skip_clear = false;   // common.h, SpSkipClear
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------ block scan helper
// exclusive scan of one int per thread over a 256-thread block (4 waves of 64); returns block total in *total.
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* total, int* lds /* >= 4 ints */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// Kernel-offset decode table of a block: entry k = oz | oy << 8 | ox << 16.  These index kernels were bound by integer
// division (ten runtime div/mod per candidate, 27 candidates per row); the decode now costs one LDS read and the stride
// division a shift.  Must be called by every thread of the block (contains a barrier).
__device__ __forceinline__ void fill_offset_table(int* tbl, int kz, int ky, int kx) {
  const int kv = kz * ky * kx;
  for (int k = threadIdx.x; k < kv; k += blockDim.x) tbl[k] = (k / (ky * kx)) | (((k / kx) % ky) << 8) | ((k % kx) << 16);
  __syncthreads();
}

// ------------------------------------------------------------------------------------------ K3 hash build
// Rows of the image-space (2-D) tensors repeat coordinates massively (every voxel outside the camera frustum clamps onto a
// border pixel), and equal keys sit in consecutive rows: a segmented max-scan over the wave folds each run of equal keys
// into its last lane, which does the one CAS + atomicMax for the run (same-address atomics serialise in L2).
__global__ void __launch_bounds__(256) hash_insert_kernel(const int32_t* __restrict__ indices, int64_t n, int ndim,
                                                          int D, int H, int W, uint64_t* __restrict__ keys,
                                                          int32_t* __restrict__ vals, uint64_t mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = i < n;
  uint64_t key = kEmptyKey - 1 - (uint64_t)lane;  // dead lanes: distinct, never equal to a real key
  if (live) {
    int b, z, y, x;
    load_coord(indices, i, ndim, b, z, y, x);
    key = (((uint64_t)b * D + z) * H + y) * W + x;
  }
  int row = live ? (int)i : -1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t ok = __shfl_up((unsigned long long)key, off, 64);
    const int orow = __shfl_up(row, off, 64);
    if (lane >= off && ok == key) row = max(row, orow);
  }
  const uint64_t nk = __shfl_down((unsigned long long)key, 1, 64);
  if (!live || (lane != 63 && nk == key)) return;  // not the tail of its run
  uint64_t slot = coord_slot(key, mask);
  for (;;) {
    unsigned long long prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmptyKey,
                                        (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) {
      atomicMax(&vals[slot], row);  // duplicate rule: highest row wins
      return;
    }
    slot = coord_next(slot, mask);
  }
}

// ------------------------------------------------------------------------------------------ K4 subm rulebook
// Block = 64 consecutive rows x 4 offset groups: thread (r, kg) probes offsets kg, kg+4, ... of row r.  All offsets of a row
// range are handled by ONE block (the old grid had the offset as blockIdx.y, i.e. the whole coordinate list was streamed
// from HBM once per offset and neighbouring offsets of a row never shared a cache); each table write is 64 consecutive ints.
__global__ void __launch_bounds__(256) subm_rulebook_kernel(const int32_t* __restrict__ indices, int64_t n, int ndim,
                                                            int D, int H, int W, int kz, int ky, int kx, int dz,
                                                            int dy, int dx, const uint64_t* __restrict__ keys,
                                                            const int32_t* __restrict__ vals, uint64_t mask,
                                                            int32_t* __restrict__ pair, int32_t* __restrict__ rep) {
  __shared__ int s_off[128];
  fill_offset_table(s_off, kz, ky, kx);
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  if (i >= n) return;
  const int kg = threadIdx.x >> 6;
  const int kv = kz * ky * kx;
  const int centre = ((kz / 2) * ky + ky / 2) * kx + kx / 2;
  int b, z, y, x;
  load_coord(indices, i, ndim, b, z, y, x);
  // Offsets in batches of 8 per thread: the first probe of every offset of the batch is issued before any is resolved (most
  // lookups end at their first slot, hit or empty), so a thread has up to 8 independent loads in flight instead of one chain.
  for (int k0 = kg; k0 < kv; k0 += 32) {
    uint64_t key[8], slot[8], found[8];
    bool ok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 4 * j;
      ok[j] = false;
      key[j] = 0; slot[j] = 0; found[j] = kEmptyKey;
      if (k >= kv || k == centre) continue;
      const int off = s_off[k];
      const int oz = off & 255, oy = (off >> 8) & 255, ox = off >> 16;
      const int nz = z + (oz - kz / 2) * dz, ny = y + (oy - ky / 2) * dy, nx = x + (ox - kx / 2) * dx;
      if (nz >= 0 && nz < D && ny >= 0 && ny < H && nx >= 0 && nx < W) {
        ok[j] = true;
        key[j] = (((uint64_t)b * D + nz) * H + ny) * W + nx;
        slot[j] = coord_slot(key[j], mask);
        found[j] = keys[slot[j]];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 4 * j;
      if (k >= kv) continue;
      if (k == centre) {
        pair[(int64_t)k * n + i] = (int32_t)i;
        if (rep) {
          const uint64_t kk = (((uint64_t)b * D + z) * H + y) * W + x;
          rep[i] = hash_lookup(keys, vals, mask, kk);
        }
        continue;
      }
      int r = -1;
      if (ok[j]) {
        uint64_t sl = slot[j], f = found[j];
        while (f != key[j] && f != kEmptyKey) {  // rare: collision chain
          sl = coord_next(sl, mask);
          f = keys[sl];
        }
        if (f == key[j]) r = vals[sl];
      }
      pair[(int64_t)k * n + i] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------ K5 strided rulebook
int g_sp_mark_variant = 2;   // vc_debug_set sp_mark_variant: 1 = sp_mark_kernel (27 offset tests + segmented scans), 2 = sp_mark2_kernel
static constexpr int kWordsPerThread = 8;
static constexpr int kWordsPerBlock = 256 * kWordsPerThread;

struct SpGeom {
  int Do, Ho, Wo;
  int k[3], s[3], p[3], d[3];
  int sh[3];  // log2(stride) when the stride is a power of two (always, in this model), else -1
};

__device__ __forceinline__ bool sp_div(int t, int s, int sh, int& q) {
  if (sh >= 0) {
    if (t & (s - 1)) return false;
    q = t >> sh;
    return true;
  }
  if (t % s) return false;
  q = t / s;
  return true;
}

// candidate output cell (linear, 64-bit) of input row coordinate (b,z,y,x) through kernel offset `off` (table entry), or -1
__device__ __forceinline__ int64_t sp_candidate(const SpGeom& g, int b, int z, int y, int x, int off) {
  const int oz = off & 255, oy = (off >> 8) & 255, ox = off >> 16;
  const int tz = z + g.p[0] - oz * g.d[0], ty = y + g.p[1] - oy * g.d[1], tx = x + g.p[2] - ox * g.d[2];
  if (tz < 0 || ty < 0 || tx < 0) return -1;
  int qz, qy, qx;
  if (!sp_div(tz, g.s[0], g.sh[0], qz) || !sp_div(ty, g.s[1], g.sh[1], qy) || !sp_div(tx, g.s[2], g.sh[2], qx)) return -1;
  if (qz >= g.Do || qy >= g.Ho || qx >= g.Wo) return -1;
  return (((int64_t)b * g.Do + qz) * g.Ho + qy) * g.Wo + qx;
}

// Same block shape as subm_rulebook_kernel (64 rows x 4 offset groups).  For one offset the 64 lanes of a wave are 64
// x-consecutive rows whose candidate cells mostly fall into the SAME 64-cell bitmap word: a segmented OR-scan over the wave
// merges them and only the last lane of each run issues the atomic (same-address L2 atomics serialise).
// n_dev (optional): the number of rows lives on the device (a strided conv chained behind another one whose output count the host
// has not read yet): n is then the row CAPACITY; the launch takes a bounded grid and strides over the real rows.
__global__ void __launch_bounds__(256) sp_mark_kernel(const int32_t* __restrict__ indices, int64_t n, int ndim,
                                                      SpGeom g, unsigned long long* __restrict__ bitmap,
                                                      const int32_t* __restrict__ n_dev) {
  if (n_dev != nullptr) {
    const int64_t real = *n_dev;
    if (real < n) n = real;
  }
  __shared__ int s_off[128];
  fill_offset_table(s_off, g.k[0], g.k[1], g.k[2]);
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int kv = g.k[0] * g.k[1] * g.k[2];
  // grid-stride over 64-row groups: a launch over a row CAPACITY (n_dev) takes a bounded grid instead of one block per 64 rows
  // of capacity (the capacities of a chain of strided convs compound: 97 k blocks for 2.7 k real ones at stage 4)
  for (int64_t rb = blockIdx.x; rb * 64 < n; rb += gridDim.x) {
    const int64_t i = rb * 64 + lane;
    const bool live = i < n;
    int b = 0, z = 0, y = 0, x = 0;
    if (live) load_coord(indices, i, ndim, b, z, y, x);
    for (int k = kg; k < kv; k += 4) {
      const int64_t L = live ? sp_candidate(g, b, z, y, x, s_off[k]) : -1;
      const bool valid = L >= 0;
      // lanes without a candidate (for stride 2 every other x) adopt the word of the nearest valid lane below them, so that
      // the lanes of one word form ONE contiguous run whose last lane flushes it
      const unsigned long long vm = __ballot(valid);
      const unsigned long long below = vm & ((lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL));
      const int src = below ? 63 - __clzll((long long)below) : lane;
      const long long w0 = valid ? (long long)(L >> 6) : (long long)(-1 - lane);
      const long long w = __shfl(w0, src, 64);
      unsigned long long bits = valid ? (1ULL << (L & 63)) : 0ULL;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const long long ow = __shfl_up(w, off, 64);
        const unsigned long long ob = __shfl_up(bits, off, 64);
        if (lane >= off && ow == w) bits |= ob;
      }
      const long long nw = __shfl_down(w, 1, 64);
      if (bits != 0ULL && (lane == 63 || nw != w)) atomicOr(&bitmap[w], bits);
    }
  }
}

// sp_mark2_kernel (round 4): the same marking with the candidates enumerated from the OUTPUT side.  Along one axis an input
// coordinate p reaches the output cells q = q0 - j, q0 = floor((p + pad) / stride), j = 0 .. J-1 with J = ceil(((k-1) d + 1) / s)
// (<= 2 for the model's k = 3, s = 2 convs: 4 (z, y) combinations x 2 adjacent x cells instead of 27 offset tests, 3.4 of them
// valid); the x cells of one combination are adjacent, so they are ONE word-and-mask (plus, rarely, a second one across a word
// border).  Thread = row (256 rows per block).  Same-word atomics of a wave are merged where there is something to merge: the
// lanes sharing the first pending lane's word OR their masks through a butterfly and one lane issues the atomic, twice (a wave of
// x-consecutive rows spans one or two words); when the first lane shares its word with nobody -- rows in permuted order, the
// training-time input of stages 2-4 after the layer discard -- every lane issues its own atomic at once.  The kernel above
// folds runs with a 6-step segmented scan per offset, 27 times per row: 55 us per launch against the ~15 us of this one.
static constexpr int kMarkJ = 4;   // candidates per axis this kernel serves (callers fall back to sp_mark_kernel beyond)
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off, 64);
  return v;
}
struct MarkAxis { int q0, valid; };   // valid: bit j set = cell q0 - j is a candidate
__device__ __forceinline__ MarkAxis mark_axis(int p, int pad, int k, int s, int sh, int d, int out, int J) {
  const int t = p + pad;                        // >= 0
  MarkAxis a;
  a.q0 = (sh >= 0) ? (t >> sh) : (t / s);
  a.valid = 0;
  for (int j = 0; j < J; ++j) {
    const int q = a.q0 - j, o = t - q * s;      // o = kernel offset * dilation
    if (q >= 0 && q < out && o % d == 0 && o / d < k) a.valid |= 1 << j;
  }
  return a;
}
__global__ void __launch_bounds__(256) sp_mark2_kernel(const int32_t* __restrict__ indices, int64_t n, int ndim, SpGeom g, int Jz, int Jy,
                                                       int Jx, unsigned long long* __restrict__ bitmap, const int32_t* __restrict__ n_dev) {
  if (n_dev != nullptr) {
    const int64_t real = *n_dev;
    if (real < n) n = real;
  }
  const int lane = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < n; base += (int64_t)gridDim.x * 256) {
    const int64_t i = base + threadIdx.x;
    const bool live = i < n;
    int b = 0, z = 0, y = 0, x = 0;
    if (live) load_coord(indices, i, ndim, b, z, y, x);
    const MarkAxis az = mark_axis(z, g.p[0], g.k[0], g.s[0], g.sh[0], g.d[0], g.Do, Jz);
    const MarkAxis ay = mark_axis(y, g.p[1], g.k[1], g.s[1], g.sh[1], g.d[1], g.Ho, Jy);
    const MarkAxis ax = mark_axis(x, g.p[2], g.k[2], g.s[2], g.sh[2], g.d[2], g.Wo, Jx);
    for (int jz = 0; jz < Jz; ++jz)
      for (int jy = 0; jy < Jy; ++jy) {
        const bool on = live && ((az.valid >> jz) & 1) && ((ay.valid >> jy) & 1) && ax.valid != 0;
        // cells (b, q0z - jz, q0y - jy, q0x - jx): the highest one decides the word, lower ones across a word border go to `lo`
        const int64_t Lhi = (((int64_t)b * g.Do + (az.q0 - jz)) * g.Ho + (ay.q0 - jy)) * g.Wo + ax.q0;
        const int64_t w = on ? (Lhi >> 6) : -1;
        unsigned long long hi = 0ULL, lo = 0ULL;
        if (on) {
          const int bit = (int)(Lhi & 63);
          for (int jx = 0; jx < Jx; ++jx)
            if ((ax.valid >> jx) & 1) {
              if (bit - jx >= 0) hi |= 1ULL << (bit - jx);
              else lo |= 1ULL << (64 + bit - jx);
            }
        }
        bool pending = hi != 0ULL;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const unsigned long long pm = __ballot(pending);
          if (pm == 0ULL) break;
          const int lead = __ffsll((long long)pm) - 1;
          const long long wl = __shfl((long long)w, lead, 64);
          const bool mine = pending && w == wl;
          if (__popcll(__ballot(mine)) <= 1) break;      // nothing to merge: permuted rows
          const unsigned long long all = wave_or64(mine ? hi : 0ULL);
          if (lane == lead) atomicOr(&bitmap[wl], all);
          if (mine) pending = false;
        }
        if (pending) atomicOr(&bitmap[w], hi);
        if (lo != 0ULL) atomicOr(&bitmap[w - 1], lo);
      }
  }
}

__global__ void __launch_bounds__(256) sp_blocksum_kernel(const unsigned long long* __restrict__ bitmap,
                                                          int64_t nwords, int32_t* __restrict__ blocksum) {
  __shared__ int lds[4];
  int64_t w0 = (int64_t)blockIdx.x * kWordsPerBlock + (int64_t)threadIdx.x * kWordsPerThread;
  int c = 0;
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j)
    if (w0 + j < nwords) c += __popcll(bitmap[w0 + j]);
  int tot;
  block_exclusive_scan_256(c, &tot, lds);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of blocksum[0..nb); total -> *total_out (and total_out2 if non-null)
__global__ void __launch_bounds__(256) scan_blocksums_kernel(int32_t* __restrict__ blocksum, int64_t nb,
                                                             int32_t* __restrict__ total_out,
                                                             int32_t* __restrict__ total_out2) {
  __shared__ int lds[4];
  int carry = 0;
  for (int64_t base = 0; base < nb; base += 256) {
    int64_t j = base + threadIdx.x;
    int v = (j < nb) ? blocksum[j] : 0;
    int tot;
    int ex = block_exclusive_scan_256(v, &tot, lds);
    if (j < nb) blocksum[j] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    *total_out = carry;
    if (total_out2) *total_out2 = carry;
  }
}

// prefix[w] = number of set bits before word w (block-local scan + scanned block sums)
__global__ void __launch_bounds__(256) sp_prefix_kernel(const unsigned long long* __restrict__ bitmap, int64_t nwords,
                                                        const int32_t* __restrict__ blocksum,
                                                        uint32_t* __restrict__ prefix) {
  __shared__ int lds[4];
  int64_t w0 = (int64_t)blockIdx.x * kWordsPerBlock + (int64_t)threadIdx.x * kWordsPerThread;
  int pc[kWordsPerThread];
  int c = 0;
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j) {
    pc[j] = (w0 + j < nwords) ? __popcll(bitmap[w0 + j]) : 0;
    c += pc[j];
  }
  int tot;
  int ex = block_exclusive_scan_256(c, &tot, lds) + blocksum[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kWordsPerThread; ++j) {
    if (w0 + j < nwords) prefix[w0 + j] = (uint32_t)ex;
    ex += pc[j];
  }
}

// one thread per bitmap word: occupied cells are clustered, so the emission is spread as thinly as possible; the
// coordinate of bit 0 is decoded once (one 64-bit division) and then walked with carries.
__global__ void __launch_bounds__(256) sp_emit_kernel(const unsigned long long* __restrict__ bitmap, int64_t nwords,
                                                      const uint32_t* __restrict__ prefix, int ndim, int Do, int Ho,
                                                      int Wo, int64_t n_out, int32_t* __restrict__ out_indices) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  unsigned long long m = bitmap[w];
  if (m == 0ULL) return;
  int64_t row = prefix[w];
  const int64_t cells = (int64_t)Do * Ho * Wo;
  const int hw = Ho * Wo;
  const int64_t L0 = w * 64;
  int b = (int)(L0 / cells);
  int rem = (int)(L0 - (int64_t)b * cells);
  int z = rem / hw;
  rem -= z * hw;
  int y = rem / Wo, x = rem - y * Wo;
  int prev = 0;
  while (m) {
    const int bit = __ffsll((long long)m) - 1;
    m &= m - 1;
    x += bit - prev;
    prev = bit;
    while (x >= Wo) {
      x -= Wo;
      if (++y >= Ho) {
        y = 0;
        if (++z >= Do) { z = 0; ++b; }
      }
    }
    if (row < n_out) {
      int32_t* o = out_indices + row * (ndim + 1);
      o[0] = b;
      if (ndim == 3) { o[1] = z; o[2] = y; o[3] = x; }
      else { o[1] = y; o[2] = x; }
    }
    ++row;
  }
}

__global__ void __launch_bounds__(256) sp_pairs_kernel(const int32_t* __restrict__ indices, int64_t n, int ndim,
                                                       SpGeom g, const unsigned long long* __restrict__ bitmap,
                                                       const uint32_t* __restrict__ prefix, int64_t n_out,
                                                       int32_t* __restrict__ pair_fwd,
                                                       int32_t* __restrict__ pair_bwd) {
  __shared__ int s_off[128];
  fill_offset_table(s_off, g.k[0], g.k[1], g.k[2]);
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);  // 64 rows x 4 offset groups, as sp_mark_kernel
  if (i >= n) return;
  const int kg = threadIdx.x >> 6;
  int b, z, y, x;
  load_coord(indices, i, ndim, b, z, y, x);
  const int kv = g.k[0] * g.k[1] * g.k[2];
  for (int k = kg; k < kv; k += 4) {
    const int64_t L = sp_candidate(g, b, z, y, x, s_off[k]);
    int o = -1;
    if (L >= 0) {
      unsigned long long w = bitmap[L >> 6];
      o = (int)prefix[L >> 6] + __popcll(w & ((1ULL << (L & 63)) - 1ULL));
      atomicMax(&pair_fwd[(int64_t)k * n_out + o], (int)i);  // duplicate inputs: highest row wins
    }
    pair_bwd[(int64_t)k * n + i] = o;
  }
}

// ------------------------------------------------------------------------------------------ K9 projection
// params per sample (32 floats): [0..11] M1 = V2C^T @ R0^T (4x3 row-major), [12..23] P2T (4x3), [24] cos(-rot),
// [25] sin(-rot), [26] flip (0/1), [27] scale, [28] has_trans (0/1)
__global__ void project_prepare_kernel(const float* __restrict__ calib, const float* __restrict__ trans, int B,
                                       float* __restrict__ params) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* v2c = calib + b * 33;  // 3x4
  const float* r0 = v2c + 12;         // 3x3
  const float* p2 = v2c + 21;         // 3x4
  float* P = params + b * 32;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) {
      float acc = __fmul_rn(v2c[0 * 4 + r], r0[c * 3 + 0]);
      acc = __fadd_rn(acc, __fmul_rn(v2c[1 * 4 + r], r0[c * 3 + 1]));
      acc = __fadd_rn(acc, __fmul_rn(v2c[2 * 4 + r], r0[c * 3 + 2]));
      P[r * 3 + c] = acc;
    }
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) P[12 + r * 3 + c] = p2[c * 4 + r];
  if (trans) {
    const float* t = trans + b * 3;
    double a = -(double)t[0];
    P[24] = (float)cos(a);
    P[25] = (float)sin(a);
    P[26] = (t[1] != 0.0f) ? 1.0f : 0.0f;
    P[27] = t[2];
    P[28] = 1.0f;
  } else {
    P[24] = 1.0f; P[25] = 0.0f; P[26] = 0.0f; P[27] = 1.0f; P[28] = 0.0f;
  }
  P[29] = P[30] = P[31] = 0.0f;
}

__device__ __forceinline__ int sat_int(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (int)0x80000000;
  return (int)f;  // truncation toward zero
}

extern int g_plan_uv_mode, g_plan_uv_lds, g_plan_uv_pad;   // plan.hip
extern long long g_plan_uv_dbg;
__device__ unsigned long long g_a17_mismatch = 0;   // MODE 6: threads whose two computations of the inverse augmentation disagreed (vc_debug_get a17_mismatch)

// MODE / DBG: developer diagnostics of LOG.md A.15 / A.17 (the product path is <0, false>).
//   MODE 4 / 5 / 6 (round 6, tools/a17_lab.py): 4 = MODE 0 at s_setprio 3; 5 = every arithmetic step of the inverse augmentation
//   fenced by 16 wait states; 6 = the block computed twice from opaque copies and compared (mismatches counted in g_a17_mismatch).
//   MODE 0: the flag word P[28] decides a divergent branch (v_cmp -> s_and_saveexec) around the inverse augmentation.
//   MODE 1: no branch on loaded data: the inverse augmentation is always computed and selected per lane (v_cndmask).
//   MODE 2: `has_trans` comes as a kernel argument (wave-uniform branch); the flag word in memory is not read at all.
//   MODE 3: MODE 0 with the three divisions by the scale as multiplications by v_rcp_f32 (no v_div_scale / v_div_fmas in the block).
// DBG: `dbg` = [0] record count, [2] 1 when the plan has augmentation parameters, [64 ..) 32-int records.  A thread whose flag word
// reads "no augmentation" although the plan has one logs the bits it saw, the ballot of its wave, and re-reads the word.
template <int MODE, bool DBG>
__global__ void __launch_bounds__(256) project_uv_kernel(const int32_t* __restrict__ indices, int64_t n,
                                                         const float* __restrict__ params, int B, int stride,
                                                         float vsx, float vsy, float vsz, float minx, float miny,
                                                         float minz, int32_t* __restrict__ uv,
                                                         float* __restrict__ depth, int32_t* dbg, int dbg_records, int has_trans_arg) {
  if constexpr (MODE == 4) __builtin_amdgcn_s_setprio(3);   // round 6 lab: this wave wins every issue arbitration on its SIMD
  if constexpr (MODE >= 200 && MODE < 300) {   // round 6 lab: MODE 0 with its code moved by 4 (MODE - 200) bytes (s_nop 0 at the entry: code placement, not timing)
#define VC_A17_SHIFT(n) asm volatile(".rept " #n "\n\ts_nop 0\n\t.endr")
    if constexpr (MODE == 201) VC_A17_SHIFT(1); else if constexpr (MODE == 202) VC_A17_SHIFT(2); else if constexpr (MODE == 204) VC_A17_SHIFT(4);
    else if constexpr (MODE == 208) VC_A17_SHIFT(8); else if constexpr (MODE == 212) VC_A17_SHIFT(12); else if constexpr (MODE == 216) VC_A17_SHIFT(16);
    else if constexpr (MODE == 224) VC_A17_SHIFT(24); else if constexpr (MODE == 232) VC_A17_SHIFT(32);
#undef VC_A17_SHIFT
  }
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 r = *reinterpret_cast<const int4*>(indices + i * 4);  // [b, z, y, x]
  const int b = r.x;
  int u = 0, v = 0;
  float dep = 0.0f;
  if (b >= 0 && b < B) {
    const float* P = params + b * 32;
    float X = __fadd_rn(__fmul_rn((float)r.w, vsx), minx);
    float Y = __fadd_rn(__fmul_rn((float)r.z, vsy), miny);
    float Z = __fadd_rn(__fmul_rn((float)r.y, vsz), minz);
    const float flag = (MODE == 2) ? (has_trans_arg ? 1.0f : 0.0f) : P[28];
    const bool has = flag != 0.0f;
    if constexpr (DBG) {
      if (!has && dbg[2] != 0) {
        const unsigned long long bal = __ballot(has);
        const int slot = atomicAdd(dbg, 1);
        if (slot < dbg_records) {
          int32_t* R = dbg + 64 + slot * 32;
          R[0] = (int32_t)i; R[1] = b; R[2] = (int32_t)blockIdx.x; R[3] = (int32_t)(threadIdx.x & 63);
          R[4] = __float_as_int(flag);
          R[5] = (int32_t)__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID
          R[6] = (int32_t)(bal & 0xffffffffu); R[7] = (int32_t)(bal >> 32);
          const volatile int32_t* Pv = reinterpret_cast<const volatile int32_t*>(P);
          R[8] = Pv[28];                                            // the same word again, at once
          __builtin_amdgcn_s_sleep(64);
          R[9] = Pv[28];                                            // ... and a moment later
          for (int j = 0; j < 5; ++j) R[10 + j] = Pv[24 + j];
          R[15] = stride; R[16] = MODE; R[31] = 0x600DF00D;
        }
      }
    }
    if constexpr (MODE == 1) {
      const float sc = P[27], fl = P[26], ca = P[24], sa = P[25], nsa = -P[25];
      float Xs = __fdiv_rn(X, sc), Ys = __fdiv_rn(Y, sc), Zs = __fdiv_rn(Z, sc);
      if (fl != 0.0f) Ys = -Ys;
      const float X2 = __fadd_rn(__fmul_rn(Xs, ca), __fmul_rn(Ys, nsa));
      const float Y2 = __fadd_rn(__fmul_rn(Xs, sa), __fmul_rn(Ys, ca));
      X = has ? X2 : X; Y = has ? Y2 : Y; Z = has ? Zs : Z;
    } else if (has) {
      const float sc = P[27];
      if constexpr (MODE == 5 || (MODE >= 100 && MODE < 200)) {   // round 6 lab: steps of the block fenced by 16 wait states (values pinned in VGPRs at each fence)
        // MODE 5: every fence; MODE 100 + mask: 1 behind the parameter loads, 2 behind the divisions, 4 behind the flip, 8 behind the four
        // products, 16 behind the first sum, 32 behind the second
        constexpr int PM = (MODE == 5) ? 63 : MODE - 100;
#define VC_A17_PAD(bit) do { if constexpr ((PM & (bit)) != 0) asm volatile("s_nop 15" : "+v"(X), "+v"(Y), "+v"(Z)); } while (0)
        float fl = P[26], ca = P[24], sa = P[25], scv = sc;
        if constexpr ((PM & 1) != 0) asm volatile("s_nop 15" : "+v"(fl), "+v"(ca), "+v"(sa), "+v"(scv), "+v"(X), "+v"(Y), "+v"(Z));
        X = __fdiv_rn(X, scv); Y = __fdiv_rn(Y, scv); Z = __fdiv_rn(Z, scv); VC_A17_PAD(2);
        if (fl != 0.0f) Y = -Y;
        VC_A17_PAD(4);
        float a = __fmul_rn(X, ca), b = __fmul_rn(Y, -sa), c = __fmul_rn(X, sa), e = __fmul_rn(Y, ca);
        if constexpr ((PM & 8) != 0) asm volatile("s_nop 15" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
        X = __fadd_rn(a, b);
        if constexpr ((PM & 16) != 0) asm volatile("s_nop 15" : "+v"(X), "+v"(c), "+v"(e));
        Y = __fadd_rn(c, e);
        VC_A17_PAD(32);
#undef VC_A17_PAD
      } else if constexpr (MODE == 6) {   // round 6 lab: the block computed twice from opaque copies of its inputs and compared
        float Xa = X, Ya = Y, Za = Z, Xb = X, Yb = Y, Zb = Z;
        asm volatile("" : "+v"(Xb), "+v"(Yb), "+v"(Zb));
        const float fl = P[26], ca = P[24], sa = P[25], nsa = -P[25];
        {
          Xa = __fdiv_rn(Xa, sc); Ya = __fdiv_rn(Ya, sc); Za = __fdiv_rn(Za, sc);
          if (fl != 0.0f) Ya = -Ya;
          const float X2 = __fadd_rn(__fmul_rn(Xa, ca), __fmul_rn(Ya, nsa));
          const float Y2 = __fadd_rn(__fmul_rn(Xa, sa), __fmul_rn(Ya, ca));
          Xa = X2; Ya = Y2;
        }
        asm volatile("" : "+v"(Xa), "+v"(Ya), "+v"(Za));
        {
          Xb = __fdiv_rn(Xb, sc); Yb = __fdiv_rn(Yb, sc); Zb = __fdiv_rn(Zb, sc);
          if (fl != 0.0f) Yb = -Yb;
          const float X2 = __fadd_rn(__fmul_rn(Xb, ca), __fmul_rn(Yb, nsa));
          const float Y2 = __fadd_rn(__fmul_rn(Xb, sa), __fmul_rn(Yb, ca));
          Xb = X2; Yb = Y2;
        }
        asm volatile("" : "+v"(Xb), "+v"(Yb), "+v"(Zb));
        if (__float_as_int(Xa) != __float_as_int(Xb) || __float_as_int(Ya) != __float_as_int(Yb) || __float_as_int(Za) != __float_as_int(Zb))
          atomicAdd(&g_a17_mismatch, 1ull);
        X = Xa; Y = Ya; Z = Za;
      } else if constexpr (MODE >= 300 && MODE < 400) {   // round 6 lab: PARTS of the block (its own idle-GPU result is the reference): which part must be there?
        if constexpr (MODE == 302 || MODE == 305) { X = __fdiv_rn(X, sc); Y = __fdiv_rn(Y, sc); Z = __fdiv_rn(Z, sc); }   // 302 scale only
        if constexpr (MODE == 304 || MODE == 305) { if (P[26] != 0.0f) Y = -Y; }                                           // 304 flip only; 305 scale + flip
        if constexpr (MODE == 303) {                                                                                       // 303 rotation only
          const float ca = P[24], sa = P[25], nsa = -P[25];
          const float X2 = __fadd_rn(__fmul_rn(X, ca), __fmul_rn(Y, nsa));
          const float Y2 = __fadd_rn(__fmul_rn(X, sa), __fmul_rn(Y, ca));
          X = X2; Y = Y2;
        }
        // 301: nothing (the divergent branch on the flag word and an empty block)
      } else {
      if constexpr (MODE == 3) {   // diagnostics only (not the reference's rounding): no IEEE division sequence in the block
        const float rs = __builtin_amdgcn_rcpf(sc);
        X = __fmul_rn(X, rs); Y = __fmul_rn(Y, rs); Z = __fmul_rn(Z, rs);
      } else {
        X = __fdiv_rn(X, sc); Y = __fdiv_rn(Y, sc); Z = __fdiv_rn(Z, sc);
      }
      if (P[26] != 0.0f) Y = -Y;
      const float ca = P[24], sa = P[25], nsa = -P[25];
      const float X2 = __fadd_rn(__fmul_rn(X, ca), __fmul_rn(Y, nsa));
      const float Y2 = __fadd_rn(__fmul_rn(X, sa), __fmul_rn(Y, ca));
      X = X2; Y = Y2;
      }
    }
    float rect[3], hom[3];
    if constexpr (MODE == 701 || MODE == 702) {
      // round 6 lab: is it the L1?  The six 16-byte loads of the matrix words written out by hand, waited for before any use -- 701 with
      // sc0 sc1 (the requests go past the CU's vector L1 to L2), 702 plain (control: the same code through the L1)
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 q0, q1, q2, q3, q4, q5;
      if constexpr (MODE == 701)
        asm volatile("global_load_dwordx4 %0, %6, off sc0 sc1\n\tglobal_load_dwordx4 %1, %6, off offset:16 sc0 sc1\n\t"
                     "global_load_dwordx4 %2, %6, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %3, %6, off offset:48 sc0 sc1\n\t"
                     "global_load_dwordx4 %4, %6, off offset:64 sc0 sc1\n\tglobal_load_dwordx4 %5, %6, off offset:80 sc0 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5) : "v"(P) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %6, off offset:16\n\t"
                     "global_load_dwordx4 %2, %6, off offset:32\n\tglobal_load_dwordx4 %3, %6, off offset:48\n\t"
                     "global_load_dwordx4 %4, %6, off offset:64\n\tglobal_load_dwordx4 %5, %6, off offset:80\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5) : "v"(P) : "memory");
      const float p[24] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3],
                           q3[0], q3[1], q3[2], q3[3], q4[0], q4[1], q4[2], q4[3], q5[0], q5[1], q5[2], q5[3]};
#pragma unroll
      for (int c = 0; c < 3; ++c)
        rect[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(X, p[0 + c]), __fmul_rn(Y, p[3 + c])), __fmul_rn(Z, p[6 + c])), p[9 + c]);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        hom[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(rect[0], p[12 + c]), __fmul_rn(rect[1], p[15 + c])), __fmul_rn(rect[2], p[18 + c])),
                           p[21 + c]);
      u = sat_int(__fdiv_rn(hom[0], rect[2]));
      v = sat_int(__fdiv_rn(hom[1], rect[2]));
      dep = __fsub_rn(hom[2], p[21 + 2]);
    } else
    if constexpr (MODE == 401 || MODE == 402 || MODE == 403) {
      // round 6 lab: is it the LOADS?  The 24 matrix words are loaded into named registers, all of them waited for (vmcnt(0)), and only then
      // used -- 401: 16 idle cycles between the wait and the first use; 402: no idle cycles (control); 403: the wait, then every register
      // read once more by a v_mov (one more pass over the loaded registers before the arithmetic)
      float p[24];
#pragma unroll
      for (int k = 0; k < 24; ++k) p[k] = P[k];
      if constexpr (MODE == 401)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),
                     "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]), "+v"(p[16]), "+v"(p[17]),
                     "+v"(p[18]), "+v"(p[19]), "+v"(p[20]), "+v"(p[21]), "+v"(p[22]), "+v"(p[23]));
      else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),
                     "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]), "+v"(p[16]), "+v"(p[17]),
                     "+v"(p[18]), "+v"(p[19]), "+v"(p[20]), "+v"(p[21]), "+v"(p[22]), "+v"(p[23]));
      if constexpr (MODE == 403) {
#pragma unroll
        for (int k = 0; k < 24; ++k) asm volatile("v_mov_b32 %0, %0" : "+v"(p[k]));
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
        rect[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(X, p[0 + c]), __fmul_rn(Y, p[3 + c])), __fmul_rn(Z, p[6 + c])), p[9 + c]);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        hom[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(rect[0], p[12 + c]), __fmul_rn(rect[1], p[15 + c])), __fmul_rn(rect[2], p[18 + c])),
                           p[21 + c]);
      u = sat_int(__fdiv_rn(hom[0], rect[2]));
      v = sat_int(__fdiv_rn(hom[1], rect[2]));
      dep = __fsub_rn(hom[2], p[21 + 2]);
    } else {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      rect[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(X, P[0 + c]), __fmul_rn(Y, P[3 + c])), __fmul_rn(Z, P[6 + c])),
                          P[9 + c]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      hom[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(rect[0], P[12 + c]), __fmul_rn(rect[1], P[15 + c])),
                                   __fmul_rn(rect[2], P[18 + c])),
                         P[21 + c]);
    u = sat_int(__fdiv_rn(hom[0], rect[2]));
    v = sat_int(__fdiv_rn(hom[1], rect[2]));
    dep = __fsub_rn(hom[2], P[21 + 2]);
    }
    if constexpr (DBG) {   // every row's intermediates, per stage (int 3 of the header: rows of capacity per stage region)
      const int cap = dbg[3];
      if (cap > 0 && i < cap) {
        int lg = 0;
        while ((1 << lg) < stride) ++lg;
        float* S8 = reinterpret_cast<float*>(dbg + 64 + 32 * dbg_records) + ((int64_t)lg * cap + i) * 8;
        S8[0] = X; S8[1] = Y; S8[2] = Z; S8[3] = rect[0]; S8[4] = rect[1]; S8[5] = rect[2]; S8[6] = hom[0]; S8[7] = hom[1];
      }
    }
  }
  u = min(max(u, 0), 1400 - 1) / stride;
  v = min(max(v, 0), 600 - 1) / stride;
  int32_t* o = uv + i * 3;
  if constexpr (MODE == 602) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(u), "+v"(v));   // round 6 lab: idle IN FRONT of the output store
  o[0] = b; o[1] = u; o[2] = v;
  if (depth) depth[i] = dep;
  if constexpr (MODE == 601) asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // ... the store acknowledged + idle BEHIND it, before s_endpgm
  if constexpr (MODE == 603) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // ... idle behind the store without waiting for it
  if constexpr (MODE == 501) {   // round 6 lab: were the COORDINATES this thread computed with the ones in memory?  (read again, at the very end)
    const volatile int32_t* iv = indices + i * 4;
    const int b2 = iv[0], z2 = iv[1], y2 = iv[2], x2 = iv[3];
    if (b2 != r.x || z2 != r.y || y2 != r.z || x2 != r.w) atomicAdd(&g_a17_mismatch, 1ull);
  }
}

// ---- the image-space branch of every block of a geometry plan in one launch (plan.hip; Uv2dArgs, common.h): projection
// (the arithmetic of project_uv_kernel<0>, op for op: spconv_backbone.py:54-83) fused with the pixel marking of image_mark_kernel --
// img[(b * U + u) * V + v] = 1 + the highest row of the pixel, runs of equal pixels in consecutive rows folded in the wave first.
// PAD (LOG.md A.17 / A.21, profiles/r06_a17_cu_mask.md section 3): 16 idle cycles behind the parameter words, behind the inverse
// augmentation and in front of the two divisions.  Same instructions, same results; in the stand-alone kernel ONE such pause anywhere in front of
// the matrix loads took the wrong pixels beside dense bf16 MFMA kernels from ~430 of 800 launches to 0.  Not a fix (the cause is not known): a
// second layer under the event fence, which stays the guarantee.  vc_debug_set plan_uv_pad = 0 selects the unpadded form (A/B, the lab).
template <bool PAD>
__device__ __forceinline__ void project_point(const int4 r, const float* __restrict__ params, int B, int stride, float vs, float minx,
                                              float miny, float minz, int& u, int& v) {
  const int b = r.x;
  u = 0; v = 0;
  if (b >= 0 && b < B) {
    const float* P = params + b * 32;
    float X = __fadd_rn(__fmul_rn((float)r.w, vs), minx);
    float Y = __fadd_rn(__fmul_rn((float)r.z, vs), miny);
    float Z = __fadd_rn(__fmul_rn((float)r.y, vs), minz);
    if (P[28] != 0.0f) {
      float sc = P[27], ca = P[24], sa = P[25];
      const float fl = P[26];
      if constexpr (PAD) asm volatile("s_nop 15" : "+v"(sc), "+v"(ca), "+v"(sa));
      X = __fdiv_rn(X, sc); Y = __fdiv_rn(Y, sc); Z = __fdiv_rn(Z, sc);
      if (fl != 0.0f) Y = -Y;
      const float nsa = -sa;
      const float X2 = __fadd_rn(__fmul_rn(X, ca), __fmul_rn(Y, nsa));
      const float Y2 = __fadd_rn(__fmul_rn(X, sa), __fmul_rn(Y, ca));
      X = X2; Y = Y2;
    }
    if constexpr (PAD) asm volatile("s_nop 15" : "+v"(X), "+v"(Y), "+v"(Z));
    float rect[3], hom[2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      rect[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(X, P[0 + c]), __fmul_rn(Y, P[3 + c])), __fmul_rn(Z, P[6 + c])),
                          P[9 + c]);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      hom[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(rect[0], P[12 + c]), __fmul_rn(rect[1], P[15 + c])),
                                   __fmul_rn(rect[2], P[18 + c])),
                         P[21 + c]);
    if constexpr (PAD) asm volatile("s_nop 15" : "+v"(hom[0]), "+v"(hom[1]), "+v"(rect[2]));
    u = sat_int(__fdiv_rn(hom[0], rect[2]));
    v = sat_int(__fdiv_rn(hom[1], rect[2]));
  }
  u = min(max(u, 0), 1400 - 1) / stride;
  v = min(max(v, 0), 600 - 1) / stride;
}

template <bool PAD>
__global__ void __launch_bounds__(256) uv_mark_multi_kernel(Uv2dArgs a) {
  int s = 0;
#pragma unroll
  for (int t = 1; t < 8; ++t)
    if (t < a.n_stages && blockIdx.x >= a.st[t].block0_mark) s = t;
  const Uv2dStage& S = a.st[s];
  const int64_t i = (int64_t)(blockIdx.x - S.block0_mark) * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool live = i < S.n;
  int64_t key = -2 - lane;   // dead lanes: distinct, never equal to a real key
  if (live) {
    const int4 r = *reinterpret_cast<const int4*>(S.coords + i * 4);  // [b, z, y, x]
    int u, v;
    project_point<PAD>(r, a.params, a.B, S.stride, S.vs, S.minx, S.miny, S.minz, u, v);
    int32_t* o = S.uv + i * 3;
    o[0] = r.x; o[1] = u; o[2] = v;
    if (r.x >= 0 && r.x < a.B && u >= 0 && u < S.U && v >= 0 && v < S.V) key = ((int64_t)r.x * S.U + u) * S.V + v;
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
  atomicMax(&S.img[key], row + 1);
}

int uv_mark_multi(const Uv2dArgs& a, unsigned total_blocks, hipStream_t st) {
  if (total_blocks == 0) return VC_OK;
  if (g_plan_uv_pad) hipLaunchKernelGGL(uv_mark_multi_kernel<true>, dim3(total_blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(uv_mark_multi_kernel<false>, dim3(total_blocks), dim3(256), 0, st, a);
  VC_CHECK_LAUNCH("uv_mark_multi_kernel");
  return VC_OK;
}

// ------------------------------------------------------------------------------------------ K2 gather / scatter rows
// one thread per float4 (or scalar tail) of an output row; C is a multiple of 4 on the hot path
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ feat,
                                                          const int32_t* __restrict__ idx, int c, int icols,
                                                          const int64_t* __restrict__ keep, int64_t n_keep,
                                                          float* __restrict__ feat_out,
                                                          int32_t* __restrict__ idx_out) {
  const int c4 = c >> 2;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_keep * c4) return;
  // c4 is a power of two for every layer of this model: shift instead of a 64-bit division per float4
  int64_t j = ((c4 & (c4 - 1)) == 0) ? (t >> (__ffs(c4) - 1)) : (t / c4);
  int q = (int)(t - j * c4);
  int64_t src = keep[j];
  reinterpret_cast<float4*>(feat_out)[j * c4 + q] = reinterpret_cast<const float4*>(feat)[src * c4 + q];
  if (q == 0 && idx_out)
    for (int a = 0; a < icols; ++a) idx_out[j * icols + a] = idx[src * icols + a];
}

// index-only variant (geometry plan: the coordinates of the kept rows are needed before any feature exists)
__global__ void __launch_bounds__(256) gather_index_rows_kernel(const int32_t* __restrict__ idx, int icols,
                                                                const int64_t* __restrict__ keep, int64_t n_keep,
                                                                int32_t* __restrict__ idx_out) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_keep * icols) return;
  int64_t j = t / icols;
  int a = (int)(t - j * icols);
  idx_out[t] = idx[keep[j] * icols + a];
}

__global__ void __launch_bounds__(256) scatter_rows_kernel(const float* __restrict__ gout, int c,
                                                           const int64_t* __restrict__ keep, int64_t n_keep,
                                                           float* __restrict__ gin) {
  const int c4 = c >> 2;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_keep * c4) return;
  int64_t j = ((c4 & (c4 - 1)) == 0) ? (t >> (__ffs(c4) - 1)) : (t / c4);
  int q = (int)(t - j * c4);
  reinterpret_cast<float4*>(gin)[keep[j] * c4 + q] = reinterpret_cast<const float4*>(gout)[j * c4 + q];
}

// ------------------------------------------------------------------------------------------ K2 random keep (layer discard)
// layer_voxel_discard keeps perm[:n_keep] of a random permutation of the N rows (spconv_backbone.py:134-147).  torch.randperm
// on the device is a radix sort of N random keys (9 launches, 0.1 ms for 3e5 rows, three times per step); a pseudo-random
// PERMUTATION can be evaluated point-wise instead: a 4-round Feistel network over 2h bits (2^(2h) >= N, < 4N) is a bijection
// of [0, 2^(2h)), and cycle-walking (re-encrypt until the value is < N) restricts it to a bijection of [0, N).  keep[i] is
// that permutation at i: distinct rows, any prefix length, one thread per kept row, no sort.
__global__ void __launch_bounds__(256) random_keep_kernel(int64_t n, int64_t n_keep, uint64_t seed, int half_bits,
                                                          int64_t* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_keep) return;
  keep[i] = (int64_t)feistel_perm((uint64_t)i, (uint64_t)n, half_bits, seed);
}

// ------------------------------------------------------------------------------------------ K10 dense
// Block = 64 rows x all channels.  The rows are read (written) as whole coalesced rows through an LDS tile, each row's
// dense base offset is computed once, and every wave instruction touches 64 rows of ONE channel plane (x-adjacent rows are
// adjacent there).  The first version used a thread per (channel, row): a 64-bit division per element and one 4-byte
// access per 256-byte feature row -- 444 MB fetched per launch for 16 MB of features (PMC).
template <bool TO_DENSE>
__global__ void __launch_bounds__(256) dense_kernel(float* __restrict__ feat, const int32_t* __restrict__ indices,
                                                    int64_t n, int c, int ndim, int D, int H, int W,
                                                    float* __restrict__ dense, int ph = 0, int pw = 0) {
  // ph / pw: the dense volume carries a border of ph rows / pw columns around every (H, W) plane (vc_to_dense_fill_padded)
  H += 2 * ph;
  W += 2 * pw;
  extern __shared__ float d_tile[];                       // [64][c + 1] floats, then 64 int64 base offsets
  const int ld = c + 1;
  int64_t* s_base = reinterpret_cast<int64_t*>(d_tile + 64 * ld + ((64 * ld) & 1));
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int rows = (int)min((int64_t)64, n - row0);
  const int64_t plane = (int64_t)D * H * W;
  if (threadIdx.x < rows) {
    int b, z, y, x;
    load_coord(indices, row0 + threadIdx.x, ndim, b, z, y, x);
    s_base[threadIdx.x] = (((int64_t)b * c * D + z) * H + y + ph) * W + x + pw;   // offset of channel 0
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (TO_DENSE) {
    for (int r = wave; r < rows; r += 4)
      for (int col = lane; col < c; col += 64) d_tile[r * ld + col] = feat[(row0 + r) * c + col];
  }
  __syncthreads();
  if (lane < rows) {
    const int64_t base = s_base[lane];
    for (int ch = wave; ch < c; ch += 4) {
      if (TO_DENSE) dense[base + ch * plane] = d_tile[lane * ld + ch];
      else d_tile[lane * ld + ch] = dense[base + ch * plane];
    }
  }
  if (!TO_DENSE) {
    __syncthreads();
    for (int r = wave; r < rows; r += 4)
      for (int col = lane; col < c; col += 64) feat[(row0 + r) * c + col] = d_tile[r * ld + col];
  }
}

// ------------------------------------------------------------------------------------------ K10 dense, write-once (f3)
// HeightCompression (height_compression.py:27-31) = .dense() + a free view (B, C, D, H, W) -> (B, C*D, H, W).  The scatter form
// above needs the whole dense tensor zero-filled first (36 MB per frame at (64, 4, 200, 176), written twice where a voxel
// lands).  Write-once form: (1) a row-id volume (B*D*H*W int32, 0.56 MB per frame) takes row+1 at every active cell (atomicMax:
// the highest row wins on duplicate coordinates, the "last write wins" of spconv's scatter); (2) one block per 64 x-consecutive
// cells of a (b, z, y) line gathers the rows of its occupied cells through LDS and writes all C channel planes of the tile,
// zeros where nothing is active: every dense element is written exactly once, coalesced along x, and no memset of the output.
__global__ void __launch_bounds__(256) dense_rowid_kernel(const int32_t* __restrict__ indices, int64_t n, int ndim, int D,
                                                          int H, int W, int32_t* __restrict__ rowid) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  int b, z, y, x;
  load_coord(indices, r, ndim, b, z, y, x);
  atomicMax(&rowid[(((int64_t)b * D + z) * H + y) * W + x], (int)(r + 1));
}

__global__ void __launch_bounds__(256) dense_fill_kernel(const float* __restrict__ feat, const int32_t* __restrict__ rowid,
                                                         int c, int D, int H, int W, int tiles_per_line,
                                                         float* __restrict__ dense, int ph, int pw) {
  // Output planes are (H + 2 ph) x (W + 2 pw): the border of the FIRST BEV conv (ZeroPad2d(1) in front of Conv2d(k3, p0),
  // base_bev_backbone.py:31-36) is written here, once, as zeros -- the 36 MB-per-frame pad copy in front of that conv disappears.
  // A block = one padded line (b, z, yp) x one 64-wide tile of padded columns.
  extern __shared__ float f_tile[];                 // [64][c + 1]
  __shared__ int s_rid[64];
  __shared__ int s_any;
  const int ld = c + 1;
  const int Hp = H + 2 * ph, Wp = W + 2 * pw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t line = blockIdx.x / tiles_per_line;  // (b*D + z)*Hp + yp
  const int x0 = (int)(blockIdx.x - line * tiles_per_line) * 64;   // padded column of the tile's first cell
  const int cells = min(64, Wp - x0);
  const int yp = (int)(line % Hp);
  const int64_t bz = line / Hp;                      // b*D + z
  const int y = yp - ph;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int x = x0 + (int)threadIdx.x - pw;
    const bool inside = threadIdx.x < cells && y >= 0 && y < H && x >= 0 && x < W;
    const int rid = inside ? rowid[(bz * H + y) * W + x] : 0;
    s_rid[threadIdx.x] = rid;
    if (rid != 0) s_any = 1;
  }
  __syncthreads();
  const int64_t b = bz / D, z = bz - b * D;
  const int64_t plane = (int64_t)D * Hp * Wp;
  float* out = dense + (b * c * D + z) * (int64_t)Hp * Wp + (int64_t)yp * Wp + x0;   // channel 0 of this tile; channel ch at + ch * plane
  if (s_any == 0) {  // most tiles: nothing active
    if (lane < cells)
      for (int ch = wave; ch < c; ch += 4) out[ch * plane + lane] = 0.f;
    return;
  }
  for (int cell = wave; cell < cells; cell += 4) {   // a wave copies one row (coalesced) per trip
    const int rid = s_rid[cell];
    for (int col = lane; col < c; col += 64) f_tile[cell * ld + col] = rid ? feat[(int64_t)(rid - 1) * c + col] : 0.f;
  }
  __syncthreads();
  if (lane < cells)
    for (int ch = wave; ch < c; ch += 4) out[ch * plane + lane] = f_tile[lane * ld + ch];
}

// ------------------------------------------------------------------------------------------ f3: first BEV conv on the sparse rows
// HeightCompression folds the height axis into the channels (height_compression.py:27-31: (B, C, D, H, W) -> (B, C*D, H, W)) and the
// first block of BaseBEVBackbone runs ZeroPad2d(1) + Conv2d(C*D -> 64, k3) over that map (base_bev_backbone.py:31-38) although
// ~70 % of its cells hold no voxel.  Seen from the sparse tensor that conv is a sparse conv from the (b, z, y, x) rows onto the
// (b, y, x) cells with kernel (D, 3, 3): offset (z, ky, kx) of cell (y, x) reads the voxel at height z (absolute) of cell
// (y + ky - 1, x + kx - 1).  bev_pairs_kernel writes its pair table for EVERY cell of the BEV grid in dense (b, y, x) order --
// no compaction, no count read: the gather-GEMM's output rows are then the NHWC dense map itself (blocks whose 64 cells see no
// voxel return at once), and nhwc_to_nchw_kernel turns it into the NCHW map the rest of the 2-D backbone reads, with the
// BatchNorm (+ReLU) that follows the conv folded into the same pass.
__global__ void __launch_bounds__(256) bev_pairs_kernel(const int32_t* __restrict__ rowid, int B, int D, int H, int W, int ky, int kx,
                                                        int32_t* __restrict__ pair) {
  const int64_t cells = (int64_t)B * H * W;
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= cells) return;
  const int x = (int)(cell % W);
  const int64_t t = cell / W;
  const int y = (int)(t % H), b = (int)(t / H);
  int k = 0;
  for (int z = 0; z < D; ++z)
    for (int a = 0; a < ky; ++a)
      for (int c = 0; c < kx; ++c, ++k) {
        const int yy = y + a - ky / 2, xx = x + c - kx / 2;
        int r = 0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) r = rowid[(((int64_t)b * D + z) * H + yy) * W + xx];
        pair[(int64_t)k * cells + cell] = r - 1;
      }
}

// The transposed table of bev_pairs_kernel, for the backward pass of the sparse BEV stem: pair_bwd[k][i] = the BEV cell that voxel row i
// feeds through offset k = (kz, a, c) -- cell (b, y - (a - ky / 2), x - (c - kx / 2)) when kz is the row's own height z (the conv's
// z "offset" is absolute: one weight slab per height, height_compression.py:30) and the cell is inside the map, -1 otherwise.
__global__ void __launch_bounds__(256) bev_pairs_bwd_kernel(const int32_t* __restrict__ indices, int64_t n, int B, int D, int H, int W, int ky,
                                                            int kx, int32_t* __restrict__ pair_bwd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 r = *reinterpret_cast<const int4*>(indices + i * 4);   // [b, z, y, x]
  const bool ok = r.x >= 0 && r.x < B && r.y >= 0 && r.y < D && r.z >= 0 && r.z < H && r.w >= 0 && r.w < W;
  int k = 0;
  for (int z = 0; z < D; ++z)
    for (int a = 0; a < ky; ++a)
      for (int c = 0; c < kx; ++c, ++k) {
        const int yy = r.z - (a - ky / 2), xx = r.w - (c - kx / 2);
        int cell = -1;
        if (ok && z == r.y && yy >= 0 && yy < H && xx >= 0 && xx < W) cell = (int)(((int64_t)r.x * H + yy) * W + xx);
        pair_bwd[(int64_t)k * n + i] = cell;
      }
}

// out[b][ch][p] = act(x[b * HW + p][ch] * scale[ch] + shift[ch]); x is (B * HW, C) row-major (NHWC), out (B, C, HW) (NCHW).
// Block = 64 consecutive cells x all channels through an LDS tile: rows are read coalesced, every store instruction writes 64
// consecutive cells of one channel plane.
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ x, int64_t hw, int c, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, float* __restrict__ out) {
  extern __shared__ float t_tile[];   // [64][c + 1]
  const int ld = c + 1;
  const int64_t tiles_per_sample = (hw + 63) / 64;
  const int64_t b = blockIdx.x / tiles_per_sample;
  const int64_t p0 = (blockIdx.x - b * tiles_per_sample) * 64;
  const int cells = (int)min((int64_t)64, hw - p0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < cells; r += 4)
    for (int col = lane; col < c; col += 64) t_tile[r * ld + col] = x[(b * hw + p0 + r) * c + col];
  __syncthreads();
  if (lane < cells)
    for (int ch = wave; ch < c; ch += 4) {
      float v = t_tile[lane * ld + ch];
      if (scale != nullptr) v = v * scale[ch] + shift[ch];
      if (relu) v = fmaxf(v, 0.f);
      out[(b * c + ch) * hw + p0 + lane] = v;
    }
}

// ------------------------------------------------------------------------------------------ generic int scan (flags)
__global__ void __launch_bounds__(256) flag_blocksum_kernel(const int32_t* __restrict__ flags, int64_t n,
                                                            int32_t* __restrict__ blocksum) {
  __shared__ int lds[4];
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int v = (i < n) ? flags[i] : 0;
  int tot;
  block_exclusive_scan_256(v, &tot, lds);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------ representative-first row order
// Backward-input of a duplicate-pixel SubM conv (2-D image-space branch): only representative rows (rep[r] == r) own
// non-centre taps, every other row takes the centre tap alone -- and only 19-53 % of the rows are representatives.  In natural
// order a 16-row tile mixes both kinds, so it issues the union (up to 9 offsets) for rows that need one.  A stable partition
// "representatives first" makes the tiles homogeneous: the tiles of the second part issue ONE offset.  (vc_row_order's mask
// sort gives the same split but is a per-window LDS sort; this is one flag scan.)
__global__ void __launch_bounds__(256) rep_flag_blocksum_kernel(const int32_t* __restrict__ rep, int64_t n,
                                                                int32_t* __restrict__ blocksum) {
  __shared__ int lds[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int v = (i < n && rep[i] == (int32_t)i) ? 1 : 0;
  int tot;
  block_exclusive_scan_256(v, &tot, lds);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256) rep_partition_kernel(const int32_t* __restrict__ rep, int64_t n,
                                                            const int32_t* __restrict__ blocksum, int64_t nb,
                                                            int32_t* __restrict__ order) {
  __shared__ int lds[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int v = (i < n && rep[i] == (int32_t)i) ? 1 : 0;
  int tot;
  const int rank = block_exclusive_scan_256(v, &tot, lds) + blocksum[blockIdx.x];  // representatives before row i
  if (i >= n) return;
  const int n_rep = blocksum[nb];
  order[v ? rank : n_rep + (int)(i - rank)] = (int32_t)i;
}

// ------------------------------------------------------------------------------------------ K1 voxelise + MeanVFE
struct VoxGeom {
  float minx, miny, minz, vx, vy, vz;
  int gx, gy, gz;
};

__global__ void __launch_bounds__(256) vox_init_kernel(uint64_t* keys, int32_t* first, int32_t* cellvid, int32_t* slots,
                                                       int64_t cap, int maxp) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  keys[i] = kEmptyKey;
  first[i] = 0x7fffffff;
  cellvid[i] = -1;
  for (int j = 0; j < maxp; ++j) slots[i * maxp + j] = 0x7fffffff;
}

__global__ void __launch_bounds__(256) vox_insert_kernel(const float* __restrict__ points, int64_t p, int f, VoxGeom g,
                                                         uint64_t* keys, int32_t* first, int32_t* slots, uint64_t mask,
                                                         int maxp, int32_t* __restrict__ pslot) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p) return;
  const float* pt = points + i * f;
  // float32: subtract, divide, floor -- one rounding each
  const float fx = floorf(__fdiv_rn(__fsub_rn(pt[0], g.minx), g.vx));
  const float fy = floorf(__fdiv_rn(__fsub_rn(pt[1], g.miny), g.vy));
  const float fz = floorf(__fdiv_rn(__fsub_rn(pt[2], g.minz), g.vz));
  if (!(fx >= 0.0f && fx < (float)g.gx && fy >= 0.0f && fy < (float)g.gy && fz >= 0.0f && fz < (float)g.gz)) {
    pslot[i] = -1;
    return;
  }
  const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
  const uint64_t key = ((uint64_t)cz * g.gy + cy) * g.gx + cx;
  uint64_t slot = mix64(key) & mask;
  for (;;) {
    unsigned long long prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmptyKey,
                                        (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) break;
    slot = (slot + 1) & mask;
  }
  pslot[i] = (int)slot;
  atomicMin(&first[slot], (int)i);
  // keep the maxp smallest point indices of the cell, sorted: atomic bubble-insert
  int v = (int)i;
  int32_t* s = slots + slot * maxp;
  for (int j = 0; j < maxp; ++j) {
    int old = atomicMin(&s[j], v);
    if (old == 0x7fffffff) break;
    if (old > v) v = old;  // displaced the larger value, carry it on
  }
}

__global__ void __launch_bounds__(256) vox_creator_kernel(const int32_t* __restrict__ pslot,
                                                          const int32_t* __restrict__ first, int64_t p,
                                                          int32_t* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p) return;
  int s = pslot[i];
  flag[i] = (s >= 0 && first[s] == (int)i) ? 1 : 0;
}

__global__ void __launch_bounds__(256) vox_assign_kernel(const float* __restrict__ points, int f, VoxGeom g,
                                                         const int32_t* __restrict__ pslot,
                                                         const int32_t* __restrict__ flag,
                                                         const int32_t* __restrict__ blocksum, int64_t p,
                                                         int max_voxels, int32_t* __restrict__ cellvid,
                                                         int32_t* __restrict__ coords) {
  __shared__ int lds[4];
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int v = (i < p) ? flag[i] : 0;
  int tot;
  int rank = block_exclusive_scan_256(v, &tot, lds) + blocksum[blockIdx.x];
  if (i < p && v && rank < max_voxels) {
    cellvid[pslot[i]] = rank;
    const float* pt = points + i * f;
    coords[rank * 3 + 0] = (int)floorf(__fdiv_rn(__fsub_rn(pt[2], g.minz), g.vz));
    coords[rank * 3 + 1] = (int)floorf(__fdiv_rn(__fsub_rn(pt[1], g.miny), g.vy));
    coords[rank * 3 + 2] = (int)floorf(__fdiv_rn(__fsub_rn(pt[0], g.minx), g.vx));
  }
}

__global__ void __launch_bounds__(256) vox_reduce_kernel(const float* __restrict__ points, int f, int maxp,
                                                         const int32_t* __restrict__ cellvid,
                                                         const int32_t* __restrict__ slots, int64_t cap,
                                                         int vfe_max_last, float* __restrict__ features,
                                                         int32_t* __restrict__ num_points) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= cap * f) return;
  int64_t s = t / f;
  int ch = (int)(t - s * f);
  int vid = cellvid[s];
  if (vid < 0) return;
  const int32_t* sl = slots + s * maxp;
  float sum = 0.0f;
  int cnt = 0;
  for (int j = 0; j < maxp; ++j) {
    int pi = sl[j];
    if (pi == 0x7fffffff) break;
    float val = points[(int64_t)pi * f + ch];
    sum = (j == 0) ? val : __fadd_rn(sum, val);
    ++cnt;
  }
  float out = __fdiv_rn(sum, (float)max(cnt, 1));
  if (vfe_max_last && ch == f - 1) {
    // max over ALL maxp slots incl. zero padding, exactly as voxels.max(dim=1)
    float m2 = (cnt < maxp) ? 0.0f : -3.402823466e+38f;
    for (int j = 0; j < cnt; ++j) m2 = fmaxf(m2, points[(int64_t)sl[j] * f + ch]);
    out = m2;
  }
  features[(int64_t)vid * f + ch] = out;
  if (ch == 0) num_points[vid] = cnt;
}

// Un-fused variant: the zero-padded (M, maxp, F) voxel tensor of Point2VoxelCPU3d.point_to_voxel (data_processor.py:53-58);
// slot j of voxel v holds the j-th point (input order) that fell into the cell, unused slots are zeros.
__global__ void __launch_bounds__(256) vox_fill_kernel(const float* __restrict__ points, int f, int maxp,
                                                       const int32_t* __restrict__ cellvid,
                                                       const int32_t* __restrict__ slots, int64_t cap,
                                                       float* __restrict__ voxels, int32_t* __restrict__ num_points) {
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= cap * f) return;
  int64_t s = t / f;
  int ch = (int)(t - s * f);
  int vid = cellvid[s];
  if (vid < 0) return;
  const int32_t* sl = slots + s * maxp;
  int cnt = 0;
  for (int j = 0; j < maxp; ++j) {
    int pi = sl[j];
    float val = 0.0f;
    if (pi != 0x7fffffff) { val = points[(int64_t)pi * f + ch]; ++cnt; }
    voxels[((int64_t)vid * maxp + j) * f + ch] = val;
  }
  if (ch == 0) num_points[vid] = cnt;
}

__global__ void vox_count_kernel(const int32_t* total, int max_voxels, int32_t* n_voxels) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *n_voxels = min(*total, max_voxels);
}

// --------------------------------------------------------------------------------------------- row order (tile homogeneity)
// The gather-GEMM issues, per 16-row tile, every kernel offset that is active for ANY row of the tile.  Rows in natural
// (ascending linear index) order have very different active sets inside a tile -- measured on KITTI-shaped scenes the tile
// union is 21 offsets against 15 per row for the stage-3 SubM convs, 18 against 5 for the strided convs and 27 against
// 3.4 for their backward.  This kernel sorts the rows of each window of WIN consecutive rows by their active-offset
// bit mask (stable: ties keep ascending row order), in LDS, one block per window; windows keep the gathers L2-local.
//
// Two paths, chosen per window.  COUNTING path (the common one: the strided backward tables have a handful of distinct masks
// per window -- parity classes of the input coordinate): the distinct masks go into a 128-slot LDS hash set, each gets its
// rank by value, every 64-row chunk counts its rows per class with ballots (the lead lane of a class broadcasts it, the
// lanes of the class take popcount-below as their rank inside the chunk), an exclusive scan over [class][chunk] gives the
// bases, and every row writes itself to base + rank: a stable counting sort with ~10 block barriers.  BITONIC path (more
// than 128 distinct masks in the window, or KV = 32): the 66-stage LDS bitonic network on (mask << 32 | row) keys.
static constexpr int kOrdSlots = 128;
static constexpr unsigned kOrdEmpty = 0xffffffffu;

template <int WIN, int THREADS>
__global__ void __launch_bounds__(THREADS) row_order_kernel(const int32_t* __restrict__ tbl, int64_t n, int kv,
                                                            const int32_t* __restrict__ rep, int centre,
                                                            int32_t* __restrict__ order) {
  static_assert(WIN == 4 * THREADS && THREADS % 64 == 0, "4 rows per thread");
  constexpr int CH = WIN / 64;                        // 64-row chunks per window
  constexpr int NCNT = kOrdSlots * CH;                // [class][chunk] counters; == 8 * THREADS
  __shared__ unsigned long long raw[WIN];             // bitonic keys, or the counters of the counting path (NCNT ints)
  __shared__ unsigned s_hash[kOrdSlots];
  __shared__ int s_rank[kOrdSlots];
  __shared__ int s_wave[THREADS / 64];
  __shared__ int s_overflow;
  int* cnt = reinterpret_cast<int*>(raw);
  const int64_t base = (int64_t)blockIdx.x * WIN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  unsigned m[4];
  bool valid[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t r = base + u * THREADS + threadIdx.x;
    valid[u] = r < n;
    m[u] = 0u;
    if (valid[u]) {
      if (rep != nullptr && rep[r] != (int32_t)r) {
        m[u] = (centre >= 0) ? (1u << centre) : 0u;   // duplicate-pixel rows only ever use the centre offset
      } else {
        for (int k0 = 0; k0 < kv; k0 += 9) {          // 9 table entries in flight per trip (one dependent load per trip made
          int v[9];                                   // the mask build the longest part of this kernel)
#pragma unroll
          for (int j = 0; j < 9; ++j) v[j] = (k0 + j < kv) ? tbl[(int64_t)(k0 + j) * n + r] : -1;
#pragma unroll
          for (int j = 0; j < 9; ++j) m[u] |= (v[j] >= 0 ? 1u : 0u) << ((k0 + j) & 31);
        }
      }
    }
  }
  for (int j = threadIdx.x; j < kOrdSlots; j += THREADS) s_hash[j] = kOrdEmpty;
  if (threadIdx.x == 0) s_overflow = (kv >= 32) ? 1 : 0;   // a full 32-bit mask would collide with the empty marker
  __syncthreads();
  // insert the distinct masks: per 64-row chunk the lanes holding the same mask elect a leader (ballot loop), only the leader
  // does the LDS CAS and broadcasts the slot -- thousands of same-address LDS atomics per window were the longest part
  int slot[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    slot[u] = -1;
    if (kv >= 32) continue;
    unsigned long long remaining = __ballot(valid[u]);
    while (remaining) {
      // the set is full (> 128 distinct masks: a strided FORWARD table, a SubM table): every further insertion would probe all
      // 128 slots with serialised LDS atomics -- 1.4-2.8 ms per table measured -- and the window takes the bitonic path anyway
      if (*reinterpret_cast<volatile int*>(&s_overflow)) break;
      const int lead = __ffsll((long long)remaining) - 1;
      const unsigned mv = (unsigned)__shfl((int)m[u], lead, 64);
      const unsigned long long same = __ballot(valid[u] && m[u] == mv);
      int found = -1;
      if (lane == lead) {
        unsigned h = (mv * 0x9e3779b1u) >> 25;        // 7 bits
        for (int probe = 0; probe < kOrdSlots; ++probe) {
          const unsigned old = atomicCAS(&s_hash[h], kOrdEmpty, mv);
          if (old == kOrdEmpty || old == mv) { found = (int)h; break; }
          h = (h + 1) & (kOrdSlots - 1);
        }
        if (found < 0) s_overflow = 1;                // benign race: every writer stores 1
      }
      found = __shfl(found, lead, 64);
      if (valid[u] && m[u] == mv) slot[u] = found;
      remaining &= ~same;
    }
  }
  __syncthreads();

  if (s_overflow) {  // ---------------------------------------------------------------- bitonic path (block-uniform)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = u * THREADS + threadIdx.x;
      raw[j] = valid[u] ? (((unsigned long long)m[u] << 32) | (unsigned)j) : ~0ULL;   // padding sorts last
    }
    __syncthreads();
    for (int size = 2; size <= WIN; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
        for (int t = threadIdx.x; t < WIN / 2; t += THREADS) {
          const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
          const int hi = lo | stride;
          const bool up = (lo & size) == 0;
          const unsigned long long a = raw[lo], b = raw[hi];
          if ((a > b) == up) { raw[lo] = b; raw[hi] = a; }
        }
        __syncthreads();
      }
    }
    for (int j = threadIdx.x; j < WIN; j += THREADS) {
      const int64_t r = base + j;
      if (r < n) order[r] = (int32_t)(base + (int64_t)(raw[j] & 0xffffffffULL));
    }
    return;
  }

  // ---------------------------------------------------------------------------------- counting path
  for (int j = threadIdx.x; j < kOrdSlots; j += THREADS) {   // rank of every distinct mask by value
    const unsigned v = s_hash[j];
    int rk = -1;
    if (v != kOrdEmpty) {
      rk = 0;
      for (int t = 0; t < kOrdSlots; ++t) rk += (s_hash[t] < v) ? 1 : 0;   // empty slots are 0xffffffff: never smaller
    }
    s_rank[j] = rk;
  }
  for (int j = threadIdx.x; j < NCNT; j += THREADS) cnt[j] = 0;
  __syncthreads();
  int cls[4], rin[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    cls[u] = valid[u] ? s_rank[slot[u]] : -1;
    rin[u] = 0;
    const int chunk = u * (THREADS / 64) + wave;      // rows u*THREADS + wave*64 .. +63: 64 consecutive rows
    unsigned long long remaining = __ballot(valid[u]);
    while (remaining) {
      const int lead = __ffsll((long long)remaining) - 1;
      const int c = __shfl(cls[u], lead, 64);
      const unsigned long long mm = __ballot(valid[u] && cls[u] == c);
      if (valid[u] && cls[u] == c) rin[u] = __popcll(mm & ((1ULL << lane) - 1ULL));
      if (lane == lead) cnt[c * CH + chunk] = __popcll(mm);
      remaining &= ~mm;
    }
  }
  __syncthreads();
  {  // exclusive scan of cnt[0 .. NCNT) in place: 8 consecutive entries per thread, wave scan, cross-wave offsets
    const int e0 = threadIdx.x * 8;
    int v[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = cnt[e0 + j]; sum += v[j]; }
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    int run = woff + inc - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) { cnt[e0 + j] = run; run += v[j]; }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (!valid[u]) continue;
    const int chunk = u * (THREADS / 64) + wave;
    const int pos = cnt[cls[u] * CH + chunk] + rin[u];
    order[base + pos] = (int32_t)(base + u * THREADS + threadIdx.x);
  }
}

int64_t a17_mismatch_read() {   // vc_debug_get a17_mismatch: synchronises the device
  unsigned long long v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_a17_mismatch), sizeof(v)) != hipSuccess) return -1;
  return (int64_t)v;
}

// diagnostics form of vc_project_uv for the geometry plan (plan.hip, vc_plan_desc.debug_buf / vc_debug_set plan_uv_mode)
int project_uv_debug(const int32_t* indices, int64_t n, const float* params, int batch_size, int stride, int32_t* uv, int32_t* dbg,
                     int dbg_records, int mode, int has_trans, hipStream_t st) {
  if (n == 0) return VC_OK;
  const double vs = 0.05 * stride;
  const float vsf = (float)vs;
  const float minx = (float)(0.0 + vs / 2), miny = (float)(-40.0 + vs / 2), minz = (float)(-3.0 + vs / 2);
  const int lds = g_plan_uv_lds;   // round 6 lab: dynamic LDS bytes of the launch (a block that takes a CU's whole LDS shares it with no LDS-using block)
#define VC_UV_LAUNCH(M, D)                                                                                                         \
  do {                                                                                                                             \
    if (lds > 65536) VC_CHECK_HIP(hipFuncSetAttribute((const void*)project_uv_kernel<M, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL((project_uv_kernel<M, D>), dim3((unsigned)cdiv(n, 256)), dim3(256), (size_t)lds, st, indices, n, params, batch_size, stride, \
                       vsf, vsf, vsf, minx, miny, minz, uv, (float*)nullptr, dbg, dbg_records, has_trans);                           \
  } while (0)
  if (dbg) {
    if (mode == 1) VC_UV_LAUNCH(1, true); else if (mode == 2) VC_UV_LAUNCH(2, true); else if (mode == 3) VC_UV_LAUNCH(3, true); else VC_UV_LAUNCH(0, true);
  } else {
    if (mode == 1) VC_UV_LAUNCH(1, false); else if (mode == 2) VC_UV_LAUNCH(2, false); else if (mode == 3) VC_UV_LAUNCH(3, false);
    else if (mode == 4) VC_UV_LAUNCH(4, false); else if (mode == 5) VC_UV_LAUNCH(5, false); else if (mode == 6) VC_UV_LAUNCH(6, false);
    else if (mode == 100) VC_UV_LAUNCH(100, false); else if (mode == 101) VC_UV_LAUNCH(101, false); else if (mode == 102) VC_UV_LAUNCH(102, false);
    else if (mode == 104) VC_UV_LAUNCH(104, false); else if (mode == 108) VC_UV_LAUNCH(108, false); else if (mode == 116) VC_UV_LAUNCH(116, false);
    else if (mode == 132) VC_UV_LAUNCH(132, false); else if (mode == 162) VC_UV_LAUNCH(162, false);
    else if (mode == 201) VC_UV_LAUNCH(201, false); else if (mode == 202) VC_UV_LAUNCH(202, false); else if (mode == 204) VC_UV_LAUNCH(204, false);
    else if (mode == 208) VC_UV_LAUNCH(208, false); else if (mode == 212) VC_UV_LAUNCH(212, false); else if (mode == 216) VC_UV_LAUNCH(216, false);
    else if (mode == 224) VC_UV_LAUNCH(224, false); else if (mode == 232) VC_UV_LAUNCH(232, false);
    else if (mode == 301) VC_UV_LAUNCH(301, false); else if (mode == 302) VC_UV_LAUNCH(302, false); else if (mode == 303) VC_UV_LAUNCH(303, false);
    else if (mode == 304) VC_UV_LAUNCH(304, false); else if (mode == 305) VC_UV_LAUNCH(305, false);
    else if (mode == 501) VC_UV_LAUNCH(501, false); else if (mode == 601) VC_UV_LAUNCH(601, false); else if (mode == 602) VC_UV_LAUNCH(602, false);
    else if (mode == 603) VC_UV_LAUNCH(603, false); else if (mode == 701) VC_UV_LAUNCH(701, false); else if (mode == 702) VC_UV_LAUNCH(702, false);
    else if (mode == 401) VC_UV_LAUNCH(401, false); else if (mode == 402) VC_UV_LAUNCH(402, false); else if (mode == 403) VC_UV_LAUNCH(403, false);
    else VC_UV_LAUNCH(0, false);
  }
#undef VC_UV_LAUNCH
  VC_CHECK_LAUNCH("project_uv_kernel<diagnostics>");
  return VC_OK;
}

}  // namespace vc

using namespace vc;

// ============================================================================================== C ABI
extern "C" {

const char* vc_version(void) { return "virconv_hip 0.1 (gfx950)"; }
int vc_abi_version(void) { return VC_ABI_VERSION; }
const char* vc_last_error(void) { return g_err; }

size_t vc_hash_workspace_bytes(int64_t n) { return (size_t)coord_hash_capacity(n < 0 ? 0 : n) * 12; }

int vc_hash_build(const int32_t* indices, int64_t n, int ndim, const int32_t* shape, void* ws, size_t ws_bytes,
                  void* stream) {
  VC_REQUIRE(ndim == 2 || ndim == 3, "vc_hash_build: ndim must be 2 or 3 (got %d)", ndim);
  VC_REQUIRE(n >= 0 && ws && shape && (indices || n == 0), "vc_hash_build: null argument");
  const uint64_t cap = coord_hash_capacity(n);
  if (ws_bytes < cap * 12) { set_error("vc_hash_build: workspace %zu < %llu", ws_bytes, (unsigned long long)cap * 12); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  VC_CHECK_HIP(hipMemsetAsync(ws, 0xFF, cap * 12, st));
  if (n == 0) return VC_OK;
  Dims d = make_dims(ndim, shape);
  uint64_t* keys = (uint64_t*)ws;
  int32_t* vals = (int32_t*)(keys + cap);
  hipLaunchKernelGGL(hash_insert_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, indices, n, ndim, d.D, d.H,
                     d.W, keys, vals, cap - 1);
  VC_CHECK_LAUNCH("hash_insert_kernel");
  return VC_OK;
}

int vc_subm_rulebook(const int32_t* indices, int64_t n, int ndim, const int32_t* shape, const int32_t* ksize,
                     const int32_t* dilation, const void* ws, size_t ws_bytes, int32_t* pair_fwd, int32_t* rep_out,
                     void* stream) {
  VC_REQUIRE(ndim == 2 || ndim == 3, "vc_subm_rulebook: ndim must be 2 or 3");
  VC_REQUIRE(n >= 0 && ws && shape && ksize && (n == 0 || (indices && pair_fwd)), "vc_subm_rulebook: null argument");
  if (n == 0) return VC_OK;
  const uint64_t cap = coord_hash_capacity(n);
  if (ws_bytes < cap * 12) { set_error("vc_subm_rulebook: workspace too small"); return VC_ECAPACITY; }
  Dims d = make_dims(ndim, shape);
  Kern3 g = make_kern(ndim, ksize, nullptr, nullptr, dilation);
  VC_REQUIRE(g.kv <= 128, "vc_subm_rulebook: kernel volume %d > 128", g.kv);
  for (int a = 0; a < 3; ++a) VC_REQUIRE(g.k[a] % 2 == 1, "vc_subm_rulebook: kernel sizes must be odd");
  const uint64_t* keys = (const uint64_t*)ws;
  const int32_t* vals = (const int32_t*)(keys + cap);
  hipLaunchKernelGGL(subm_rulebook_kernel, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream,
                     indices, n, ndim, d.D, d.H, d.W, g.k[0], g.k[1], g.k[2], g.d[0], g.d[1], g.d[2], keys, vals,
                     cap - 1, pair_fwd, rep_out);
  VC_CHECK_LAUNCH("subm_rulebook_kernel");
  return VC_OK;
}

// spconv workspace: [bitmap u64 x nwords][prefix u32 x nwords][blocksum i32 x nb][total i32]
static inline void sp_layout(int batch, int ndim, const int32_t* out_shape, int64_t& nwords, int64_t& nb) {
  Dims d = make_dims(ndim, out_shape);
  int64_t cells = (int64_t)batch * d.D * d.H * d.W;
  nwords = cdiv(cells, 64);
  if (nwords < 1) nwords = 1;
  nb = cdiv(nwords, kWordsPerBlock);
}

size_t vc_spconv_workspace_bytes(int batch_size, int ndim, const int32_t* out_shape) {
  if (!out_shape || (ndim != 2 && ndim != 3) || batch_size < 1) return 0;
  int64_t nwords, nb;
  sp_layout(batch_size, ndim, out_shape, nwords, nb);
  return (size_t)(nwords * 12 + (nb + 1) * 4 + 64);
}

static inline SpGeom make_spgeom(const Dims& o, const Kern3& k) {
  SpGeom g;
  g.Do = o.D; g.Ho = o.H; g.Wo = o.W;
  for (int a = 0; a < 3; ++a) {
    g.k[a] = k.k[a]; g.s[a] = k.s[a]; g.p[a] = k.p[a]; g.d[a] = k.d[a];
    g.sh[a] = -1;
    if (k.s[a] > 0 && (k.s[a] & (k.s[a] - 1)) == 0) {
      int sh = 0;
      while ((1 << sh) < k.s[a]) ++sh;
      g.sh[a] = sh;
    }
  }
  return g;
}

static int spconv_mark_count(const int32_t* indices, int64_t n, const int32_t* n_dev, int ndim, int batch_size,
                             const int32_t* out_shape, const int32_t* ksize, const int32_t* stride_, const int32_t* padding,
                             const int32_t* dilation, void* ws, size_t ws_bytes, int32_t* n_out_dev, void* stream);

int vc_spconv_mark_count(const int32_t* indices, int64_t n, int ndim, int batch_size, const int32_t* out_shape,
                         const int32_t* ksize, const int32_t* stride_, const int32_t* padding, const int32_t* dilation,
                         void* ws, size_t ws_bytes, int32_t* n_out_dev, void* stream) {
  return spconv_mark_count(indices, n, nullptr, ndim, batch_size, out_shape, ksize, stride_, padding, dilation, ws, ws_bytes,
                           n_out_dev, stream);
}

int vc_spconv_mark_count_dev(const int32_t* indices, int64_t n_capacity, const int32_t* n_dev, int ndim, int batch_size,
                             const int32_t* out_shape, const int32_t* ksize, const int32_t* stride_, const int32_t* padding,
                             const int32_t* dilation, void* ws, size_t ws_bytes, int32_t* n_out_dev, void* stream) {
  VC_REQUIRE(n_dev != nullptr, "vc_spconv_mark_count_dev: null device row count");
  return spconv_mark_count(indices, n_capacity, n_dev, ndim, batch_size, out_shape, ksize, stride_, padding, dilation, ws,
                           ws_bytes, n_out_dev, stream);
}

static int spconv_mark_count(const int32_t* indices, int64_t n, const int32_t* n_dev, int ndim, int batch_size,
                             const int32_t* out_shape, const int32_t* ksize, const int32_t* stride_, const int32_t* padding,
                             const int32_t* dilation, void* ws, size_t ws_bytes, int32_t* n_out_dev, void* stream) {
  VC_REQUIRE(ndim == 2 || ndim == 3, "vc_spconv_mark_count: ndim must be 2 or 3");
  VC_REQUIRE(n >= 0 && batch_size >= 1 && out_shape && ksize && stride_ && padding && ws && n_out_dev &&
                 (indices || n == 0), "vc_spconv_mark_count: null/invalid argument");
  Dims o = make_dims(ndim, out_shape);
  VC_REQUIRE((int64_t)o.D * o.H * o.W < (1LL << 31), "vc_spconv_mark_count: output grid per sample must be < 2^31 cells");
  int64_t nwords, nb;
  sp_layout(batch_size, ndim, out_shape, nwords, nb);
  if (ws_bytes < vc_spconv_workspace_bytes(batch_size, ndim, out_shape)) { set_error("vc_spconv_mark_count: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* bitmap = (unsigned long long*)ws;
  uint32_t* prefix = (uint32_t*)(bitmap + nwords);
  int32_t* blocksum = (int32_t*)(prefix + nwords);
  if (!t_sp_skip_clear) VC_CHECK_HIP(hipMemsetAsync(bitmap, 0, nwords * 8, st));
  Kern3 k = make_kern(ndim, ksize, stride_, padding, dilation);
  VC_REQUIRE(k.kv <= 128, "strided rulebook: kernel volume %d > 128", k.kv);
  SpGeom g = make_spgeom(o, k);
  if (n > 0) {
    int J[3];
    bool v2 = g_sp_mark_variant == 2;
    for (int a = 0; a < 3; ++a) {
      J[a] = ((k.k[a] - 1) * k.d[a] + 1 + k.s[a] - 1) / k.s[a];
      if (J[a] > kMarkJ) v2 = false;
    }
    if (v2) {
      int64_t mark_blocks = cdiv(n, 256);
      if (n_dev != nullptr && mark_blocks > 4096) mark_blocks = 4096;   // capacity launch: grid-stride inside the kernel
      hipLaunchKernelGGL(sp_mark2_kernel, dim3((unsigned)mark_blocks), dim3(256), 0, st, indices, n, ndim, g, J[0], J[1], J[2], bitmap, n_dev);
      VC_CHECK_LAUNCH("sp_mark2_kernel");
    } else {
      int64_t mark_blocks = cdiv(n, 64);
      if (n_dev != nullptr && mark_blocks > 8192) mark_blocks = 8192;   // capacity launch: grid-stride inside the kernel
      hipLaunchKernelGGL(sp_mark_kernel, dim3((unsigned)mark_blocks), dim3(256), 0, st, indices, n, ndim, g, bitmap, n_dev);
      VC_CHECK_LAUNCH("sp_mark_kernel");
    }
  }
  hipLaunchKernelGGL(sp_blocksum_kernel, dim3((unsigned)nb), dim3(256), 0, st, bitmap, nwords, blocksum);
  VC_CHECK_LAUNCH("sp_blocksum_kernel");
  hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(256), 0, st, blocksum, nb, n_out_dev, blocksum + nb);
  VC_CHECK_LAUNCH("scan_blocksums_kernel");
  return VC_OK;
}

// Second stage in two halves, so that a chain of strided convs can run stage 1 + the coordinate emission of every level before
// the host has read a single count (the emission takes a row CAPACITY), and build the pair tables once the counts are known.
int vc_spconv_emit_indices(int ndim, int batch_size, const int32_t* out_shape, void* ws, size_t ws_bytes, int64_t capacity,
                           int32_t* out_indices, void* stream) {
  VC_REQUIRE(ndim == 2 || ndim == 3, "vc_spconv_emit_indices: ndim must be 2 or 3");
  VC_REQUIRE(capacity >= 0 && out_shape && ws && (out_indices || capacity == 0), "vc_spconv_emit_indices: null argument");
  Dims o = make_dims(ndim, out_shape);
  int64_t nwords, nb;
  sp_layout(batch_size, ndim, out_shape, nwords, nb);
  if (ws_bytes < vc_spconv_workspace_bytes(batch_size, ndim, out_shape)) { set_error("vc_spconv_emit_indices: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  const unsigned long long* bitmap = (const unsigned long long*)ws;
  uint32_t* prefix = (uint32_t*)(bitmap + nwords);
  const int32_t* blocksum = (const int32_t*)(prefix + nwords);
  hipLaunchKernelGGL(sp_prefix_kernel, dim3((unsigned)nb), dim3(256), 0, st, bitmap, nwords, blocksum, prefix);
  VC_CHECK_LAUNCH("sp_prefix_kernel");
  hipLaunchKernelGGL(sp_emit_kernel, dim3((unsigned)cdiv(nwords, 256)), dim3(256), 0, st, bitmap, nwords, prefix, ndim,
                     o.D, o.H, o.W, capacity, out_indices);
  VC_CHECK_LAUNCH("sp_emit_kernel");
  return VC_OK;
}

static int spconv_pairs(const int32_t* indices, int64_t n, int ndim, int batch_size, const int32_t* out_shape,
                        const int32_t* ksize, const int32_t* stride_, const int32_t* padding, const int32_t* dilation,
                        const void* ws, size_t ws_bytes, int64_t n_out, int32_t* pair_fwd, int32_t* pair_bwd, void* stream);

int vc_spconv_pairs(const int32_t* indices, int64_t n, int ndim, int batch_size, const int32_t* out_shape,
                    const int32_t* ksize, const int32_t* stride_, const int32_t* padding, const int32_t* dilation,
                    const void* ws, size_t ws_bytes, int64_t n_out, int32_t* pair_fwd, int32_t* pair_bwd, void* stream) {
  VC_REQUIRE(ndim == 2 || ndim == 3, "vc_spconv_pairs: ndim must be 2 or 3");
  VC_REQUIRE(n >= 0 && n_out >= 0 && out_shape && ksize && stride_ && padding && ws, "vc_spconv_pairs: null argument");
  VC_REQUIRE((n_out == 0 || pair_fwd) && (n == 0 || (indices && pair_bwd)), "vc_spconv_pairs: null tables");
  if (ws_bytes < vc_spconv_workspace_bytes(batch_size, ndim, out_shape)) { set_error("vc_spconv_pairs: workspace too small"); return VC_ECAPACITY; }
  return spconv_pairs(indices, n, ndim, batch_size, out_shape, ksize, stride_, padding, dilation, ws, ws_bytes, n_out, pair_fwd,
                      pair_bwd, stream);
}

int vc_spconv_emit_pairs(const int32_t* indices, int64_t n, int ndim, int batch_size, const int32_t* out_shape,
                         const int32_t* ksize, const int32_t* stride_, const int32_t* padding, const int32_t* dilation,
                         const void* ws, size_t ws_bytes, int64_t n_out, int32_t* out_indices, int32_t* pair_fwd,
                         int32_t* pair_bwd, void* stream) {
  VC_REQUIRE(ndim == 2 || ndim == 3, "vc_spconv_emit_pairs: ndim must be 2 or 3");
  VC_REQUIRE(n >= 0 && n_out >= 0 && out_shape && ksize && stride_ && padding && ws, "vc_spconv_emit_pairs: null argument");
  VC_REQUIRE(n_out == 0 || (out_indices && pair_fwd), "vc_spconv_emit_pairs: null outputs");
  VC_REQUIRE(n == 0 || (indices && pair_bwd), "vc_spconv_emit_pairs: null inputs");
  const int rc = vc_spconv_emit_indices(ndim, batch_size, out_shape, const_cast<void*>(ws), ws_bytes, n_out, out_indices, stream);
  if (rc != VC_OK) return rc;
  return spconv_pairs(indices, n, ndim, batch_size, out_shape, ksize, stride_, padding, dilation, ws, ws_bytes, n_out, pair_fwd,
                      pair_bwd, stream);
}

static int spconv_pairs(const int32_t* indices, int64_t n, int ndim, int batch_size, const int32_t* out_shape,
                        const int32_t* ksize, const int32_t* stride_, const int32_t* padding, const int32_t* dilation,
                        const void* ws, size_t ws_bytes, int64_t n_out, int32_t* pair_fwd, int32_t* pair_bwd, void* stream) {
  (void)ws_bytes;
  Dims o = make_dims(ndim, out_shape);
  int64_t nwords, nb;
  sp_layout(batch_size, ndim, out_shape, nwords, nb);
  hipStream_t st = (hipStream_t)stream;
  const unsigned long long* bitmap = (const unsigned long long*)ws;
  const uint32_t* prefix = (const uint32_t*)(bitmap + nwords);
  Kern3 k = make_kern(ndim, ksize, stride_, padding, dilation);
  VC_REQUIRE(k.kv <= 128, "strided rulebook: kernel volume %d > 128", k.kv);
  SpGeom g = make_spgeom(o, k);
  if (n_out > 0 && !t_sp_skip_clear) VC_CHECK_HIP(hipMemsetAsync(pair_fwd, 0xFF, (size_t)k.kv * n_out * 4, st));
  if (n > 0) {
    hipLaunchKernelGGL(sp_pairs_kernel, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, st, indices, n, ndim, g,
                       bitmap, prefix, n_out, pair_fwd, pair_bwd);
    VC_CHECK_LAUNCH("sp_pairs_kernel");
  }
  return VC_OK;
}

int vc_project_prepare(const float* calib, const float* trans, int batch_size, float* params, void* stream) {
  VC_REQUIRE(calib && params && batch_size >= 1, "vc_project_prepare: null/invalid argument");
  // 32 threads per block: this floating-point kernel runs at the head of the geometry plan, beside the previous step's conv kernels,
  // and what LOG.md A.17 saw go wrong there were lanes 48-63 of a wave -- no sample's parameters are ever computed in those lanes
  hipLaunchKernelGGL(project_prepare_kernel, dim3((unsigned)cdiv(batch_size, 32)), dim3(32), 0, (hipStream_t)stream,
                     calib, trans, batch_size, params);
  VC_CHECK_LAUNCH("project_prepare_kernel");
  return VC_OK;
}

int vc_project_uv(const int32_t* indices, int64_t n, const float* params, int batch_size, int stride, int32_t* uv,
                  float* depth, void* stream) {
  VC_REQUIRE(n >= 0 && params && stride >= 1 && (n == 0 || (indices && uv)), "vc_project_uv: null/invalid argument");
  if (n == 0) return VC_OK;
  if ((g_plan_uv_mode != 0 || g_plan_uv_lds != 0 || g_plan_uv_dbg != 0) && !depth)   // developer diagnostics (vc_debug_set plan_uv_mode / plan_uv_lds / plan_uv_dbg; tools/a17_lab.py)
    return project_uv_debug(indices, n, params, batch_size, stride, uv, (int32_t*)(uintptr_t)g_plan_uv_dbg, 0, g_plan_uv_mode, 1, (hipStream_t)stream);
  // hard-coded range / voxel size of the reference (spconv_backbone.py:8): python floats (fp64) rounded to fp32 on use
  const double vs = 0.05 * stride;
  const float vsf = (float)vs;
  const float minx = (float)(0.0 + vs / 2), miny = (float)(-40.0 + vs / 2), minz = (float)(-3.0 + vs / 2);
  hipLaunchKernelGGL((project_uv_kernel<0, false>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, indices, n,
                     params, batch_size, stride, vsf, vsf, vsf, minx, miny, minz, uv, depth, (int32_t*)nullptr, 0, 0);
  VC_CHECK_LAUNCH("project_uv_kernel");
  return VC_OK;
}


int vc_gather_rows(const float* features, const int32_t* indices, int c, int icols, const int64_t* keep, int64_t n_keep,
                   float* features_out, int32_t* indices_out, void* stream) {
  VC_REQUIRE(!indices_out || (indices && icols > 0), "vc_gather_rows: indices_out without indices");
  if (!features && !features_out && indices_out) {  // index-only gather
    VC_REQUIRE(n_keep >= 0 && (n_keep == 0 || keep), "vc_gather_rows: null argument");
    if (n_keep == 0) return VC_OK;
    hipLaunchKernelGGL(gather_index_rows_kernel, dim3((unsigned)cdiv(n_keep * icols, 256)), dim3(256), 0,
                       (hipStream_t)stream, indices, icols, keep, n_keep, indices_out);
    VC_CHECK_LAUNCH("gather_index_rows_kernel");
    return VC_OK;
  }
  VC_REQUIRE(c > 0 && c % 4 == 0, "vc_gather_rows: channel count must be a positive multiple of 4 (got %d)", c);
  VC_REQUIRE(n_keep >= 0 && (n_keep == 0 || (features && keep && features_out)), "vc_gather_rows: null argument");
  if (n_keep == 0) return VC_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv(n_keep * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     features, indices, c, icols, keep, n_keep, features_out, indices_out);
  VC_CHECK_LAUNCH("gather_rows_kernel");
  return VC_OK;
}

int vc_scatter_rows(const float* grad_out, int c, const int64_t* keep, int64_t n_keep, int64_t n_in, float* grad_in,
                    void* stream) {
  VC_REQUIRE(c > 0 && c % 4 == 0, "vc_scatter_rows: channel count must be a positive multiple of 4");
  VC_REQUIRE(n_keep >= 0 && n_in >= 0 && (n_in == 0 || grad_in), "vc_scatter_rows: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (n_in > 0) VC_CHECK_HIP(hipMemsetAsync(grad_in, 0, (size_t)n_in * c * 4, st));
  if (n_keep == 0) return VC_OK;
  VC_REQUIRE(grad_out && keep, "vc_scatter_rows: null argument");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)cdiv(n_keep * (c / 4), 256)), dim3(256), 0, st, grad_out, c,
                     keep, n_keep, grad_in);
  VC_CHECK_LAUNCH("scatter_rows_kernel");
  return VC_OK;
}

int vc_to_dense(const float* features, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                const int32_t* shape, float* dense, void* stream) {
  VC_REQUIRE((ndim == 2 || ndim == 3) && shape && c > 0 && batch_size >= 1, "vc_to_dense: invalid argument");
  if (n == 0) return VC_OK;
  VC_REQUIRE(features && indices && dense, "vc_to_dense: null argument");
  Dims d = make_dims(ndim, shape);
  const size_t lds = (size_t)64 * (c + 1) * sizeof(float) + 64 * sizeof(int64_t) + 8;
  VC_REQUIRE(lds <= 64 * 1024, "vc_to_dense: channel count %d too large", c);
  hipLaunchKernelGGL(dense_kernel<true>, dim3((unsigned)cdiv(n, 64)), dim3(256), lds, (hipStream_t)stream,
                     const_cast<float*>(features), indices, n, c, ndim, d.D, d.H, d.W, dense);
  VC_CHECK_LAUNCH("dense_kernel<to>");
  return VC_OK;
}

size_t vc_to_dense_fill_workspace_bytes(int batch_size, int ndim, const int32_t* shape) {
  if ((ndim != 2 && ndim != 3) || !shape || batch_size < 1) return 0;
  Dims d = make_dims(ndim, shape);
  return (size_t)batch_size * d.D * d.H * d.W * sizeof(int32_t);
}

int vc_to_dense_fill(const float* features, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                     const int32_t* shape, float* dense, void* ws, size_t ws_bytes, void* stream) {
  return vc_to_dense_fill_padded(features, indices, n, c, ndim, batch_size, shape, 0, 0, dense, ws, ws_bytes, stream);
}

int vc_to_dense_fill_padded(const float* features, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                            const int32_t* shape, int pad_h, int pad_w, float* dense, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE((ndim == 2 || ndim == 3) && shape && c > 0 && batch_size >= 1 && dense && ws, "vc_to_dense_fill: invalid argument");
  VC_REQUIRE(pad_h >= 0 && pad_w >= 0 && pad_h <= 64 && pad_w <= 64, "vc_to_dense_fill_padded: invalid padding");
  VC_REQUIRE(n == 0 || (features && indices), "vc_to_dense_fill: null argument");
  if (ws_bytes < vc_to_dense_fill_workspace_bytes(batch_size, ndim, shape)) { set_error("vc_to_dense_fill: workspace too small"); return VC_ECAPACITY; }
  Dims d = make_dims(ndim, shape);
  hipStream_t st = (hipStream_t)stream;
  const int64_t lines = (int64_t)batch_size * d.D * (d.H + 2 * pad_h);
  const int tiles = (int)cdiv(d.W + 2 * pad_w, 64);
  VC_REQUIRE(lines * tiles < (1LL << 31) && n < (1LL << 31) - 1, "vc_to_dense_fill: tensor too large");
  const size_t lds = (size_t)64 * (c + 1) * sizeof(float);
  VC_REQUIRE(lds <= 60 * 1024, "vc_to_dense_fill: channel count %d too large", c);
  int32_t* rowid = (int32_t*)ws;
  VC_CHECK_HIP(hipMemsetAsync(rowid, 0, vc_to_dense_fill_workspace_bytes(batch_size, ndim, shape), st));
  if (n > 0) {
    hipLaunchKernelGGL(dense_rowid_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, indices, n, ndim, d.D, d.H, d.W, rowid);
    VC_CHECK_LAUNCH("dense_rowid_kernel");
  }
  hipLaunchKernelGGL(dense_fill_kernel, dim3((unsigned)(lines * tiles)), dim3(256), lds, st, features, rowid, c, d.D, d.H, d.W,
                     tiles, dense, pad_h, pad_w);
  VC_CHECK_LAUNCH("dense_fill_kernel");
  return VC_OK;
}

size_t vc_bev_pairs_workspace_bytes(int batch_size, const int32_t* shape) {
  if (!shape || batch_size < 1) return 0;
  return (size_t)batch_size * shape[0] * shape[1] * shape[2] * sizeof(int32_t);
}

int vc_bev_pairs(const int32_t* indices, int64_t n, int batch_size, const int32_t* shape, int ky, int kx, int32_t* pair, void* ws,
                 size_t ws_bytes, void* stream) {
  VC_REQUIRE(shape && batch_size >= 1 && pair && ws && ky >= 1 && kx >= 1 && ky % 2 == 1 && kx % 2 == 1, "vc_bev_pairs: invalid argument");
  VC_REQUIRE(n == 0 || indices, "vc_bev_pairs: null indices");
  const int D = shape[0], H = shape[1], W = shape[2];
  VC_REQUIRE(D >= 1 && H >= 1 && W >= 1 && (int64_t)D * ky * kx <= 128, "vc_bev_pairs: kernel volume D * ky * kx must be <= 128");
  if (ws_bytes < vc_bev_pairs_workspace_bytes(batch_size, shape)) { set_error("vc_bev_pairs: workspace too small"); return VC_ECAPACITY; }
  VC_REQUIRE(n < (1LL << 31) - 1 && (int64_t)batch_size * H * W < (1LL << 31), "vc_bev_pairs: tensor too large");
  hipStream_t st = (hipStream_t)stream;
  int32_t* rowid = (int32_t*)ws;
  VC_CHECK_HIP(hipMemsetAsync(rowid, 0, vc_bev_pairs_workspace_bytes(batch_size, shape), st));
  if (n > 0) {
    hipLaunchKernelGGL(dense_rowid_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, indices, n, 3, D, H, W, rowid);
    VC_CHECK_LAUNCH("dense_rowid_kernel");
  }
  const int64_t cells = (int64_t)batch_size * H * W;
  hipLaunchKernelGGL(bev_pairs_kernel, dim3((unsigned)cdiv(cells, 256)), dim3(256), 0, st, (const int32_t*)rowid, batch_size, D, H, W, ky, kx,
                     pair);
  VC_CHECK_LAUNCH("bev_pairs_kernel");
  return VC_OK;
}

int vc_bev_pairs_backward(const int32_t* indices, int64_t n, int batch_size, const int32_t* shape, int ky, int kx, int32_t* pair_bwd,
                          void* stream) {
  VC_REQUIRE(shape && batch_size >= 1 && ky >= 1 && kx >= 1 && ky % 2 == 1 && kx % 2 == 1, "vc_bev_pairs_backward: invalid argument");
  VC_REQUIRE(n >= 0 && (n == 0 || (indices && pair_bwd)), "vc_bev_pairs_backward: null argument");
  const int D = shape[0], H = shape[1], W = shape[2];
  VC_REQUIRE(D >= 1 && H >= 1 && W >= 1 && (int64_t)D * ky * kx <= 128, "vc_bev_pairs_backward: kernel volume D * ky * kx must be <= 128");
  VC_REQUIRE((int64_t)batch_size * H * W < (1LL << 31), "vc_bev_pairs_backward: map too large");
  if (n == 0) return VC_OK;
  hipLaunchKernelGGL(bev_pairs_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, indices, n, batch_size, D, H, W,
                     ky, kx, pair_bwd);
  VC_CHECK_LAUNCH("bev_pairs_bwd_kernel");
  return VC_OK;
}

int vc_nhwc_to_nchw(const float* x, int batch_size, int64_t hw, int c, const float* scale, const float* shift, int relu, float* out,
                    void* stream) {
  VC_REQUIRE(x && out && batch_size >= 1 && hw >= 1 && c >= 1, "vc_nhwc_to_nchw: invalid argument");
  VC_REQUIRE((scale == nullptr) == (shift == nullptr), "vc_nhwc_to_nchw: scale and shift come together");
  const size_t lds = (size_t)64 * (c + 1) * sizeof(float);
  VC_REQUIRE(lds <= 60 * 1024, "vc_nhwc_to_nchw: channel count %d too large", c);
  const int64_t blocks = (int64_t)batch_size * ((hw + 63) / 64);
  VC_REQUIRE(blocks < (1LL << 31), "vc_nhwc_to_nchw: tensor too large");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, hw, c, scale, shift, relu, out);
  VC_CHECK_LAUNCH("nhwc_to_nchw_kernel");
  return VC_OK;
}

int vc_from_dense(const float* dense, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                  const int32_t* shape, float* features, void* stream) {
  return vc_from_dense_padded(dense, indices, n, c, ndim, batch_size, shape, 0, 0, features, stream);
}

int vc_from_dense_padded(const float* dense, const int32_t* indices, int64_t n, int c, int ndim, int batch_size,
                         const int32_t* shape, int pad_h, int pad_w, float* features, void* stream) {
  VC_REQUIRE((ndim == 2 || ndim == 3) && shape && c > 0 && batch_size >= 1, "vc_from_dense: invalid argument");
  VC_REQUIRE(pad_h >= 0 && pad_w >= 0, "vc_from_dense_padded: invalid padding");
  if (n == 0) return VC_OK;
  VC_REQUIRE(features && indices && dense, "vc_from_dense: null argument");
  Dims d = make_dims(ndim, shape);
  const size_t lds = (size_t)64 * (c + 1) * sizeof(float) + 64 * sizeof(int64_t) + 8;
  VC_REQUIRE(lds <= 64 * 1024, "vc_from_dense: channel count %d too large", c);
  hipLaunchKernelGGL(dense_kernel<false>, dim3((unsigned)cdiv(n, 64)), dim3(256), lds, (hipStream_t)stream, features,
                     indices, n, c, ndim, d.D, d.H, d.W, const_cast<float*>(dense), pad_h, pad_w);
  VC_CHECK_LAUNCH("dense_kernel<from>");
  return VC_OK;
}

// voxeliser workspace: keys u64[cap] | first i32[cap] | cellvid i32[cap] | slots i32[cap*maxp] | pslot i32[p] |
//                      flag i32[p] | blocksum i32[nb+1]
size_t vc_voxelize_workspace_bytes(int64_t p, int max_points) {
  if (p < 0 || max_points < 1) return 0;
  uint64_t cap = hash_capacity(p);
  int64_t nb = cdiv(p > 0 ? p : 1, 256);
  return (size_t)(cap * (8 + 4 + 4 + 4 * (uint64_t)max_points) + (uint64_t)p * 8 + (nb + 1) * 4 + 64);
}

// fused = true : `out` = per-voxel mean features (max_voxels, f)            (vc_voxelize_mean)
// fused = false: `out` = zero-padded voxels (max_voxels, max_points, f)     (vc_voxelize)
static int voxelize_impl(bool fused, const float* points, int64_t p, int f, const float* range, const float* vsize,
                         int max_points, int max_voxels, int vfe_max_last, void* ws, size_t ws_bytes, float* features,
                         int32_t* coords, int32_t* num_points, int32_t* n_voxels_dev, void* stream) {
  VC_REQUIRE(p >= 0 && f >= 3 && range && vsize && max_points >= 1 && max_voxels >= 1 && ws && features && coords &&
                 num_points && n_voxels_dev && (points || p == 0), "vc_voxelize: null/invalid argument");
  if (ws_bytes < vc_voxelize_workspace_bytes(p, max_points)) { set_error("vc_voxelize: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  const uint64_t cap = hash_capacity(p);
  const int64_t nb = cdiv(p > 0 ? p : 1, 256);
  uint64_t* keys = (uint64_t*)ws;
  int32_t* first = (int32_t*)(keys + cap);
  int32_t* cellvid = first + cap;
  int32_t* slots = cellvid + cap;
  int32_t* pslot = slots + cap * max_points;
  int32_t* flag = pslot + p;
  int32_t* blocksum = flag + p;
  VoxGeom g;
  g.minx = range[0]; g.miny = range[1]; g.minz = range[2];
  g.vx = vsize[0]; g.vy = vsize[1]; g.vz = vsize[2];
  g.gx = (int)llround(((double)range[3] - (double)range[0]) / (double)vsize[0]);
  g.gy = (int)llround(((double)range[4] - (double)range[1]) / (double)vsize[1]);
  g.gz = (int)llround(((double)range[5] - (double)range[2]) / (double)vsize[2]);
  hipLaunchKernelGGL(vox_init_kernel, dim3((unsigned)cdiv((int64_t)cap, 256)), dim3(256), 0, st, keys, first, cellvid,
                     slots, (int64_t)cap, max_points);
  VC_CHECK_LAUNCH("vox_init_kernel");
  if (p > 0) {
    hipLaunchKernelGGL(vox_insert_kernel, dim3((unsigned)cdiv(p, 256)), dim3(256), 0, st, points, p, f, g, keys, first,
                       slots, cap - 1, max_points, pslot);
    VC_CHECK_LAUNCH("vox_insert_kernel");
    hipLaunchKernelGGL(vox_creator_kernel, dim3((unsigned)cdiv(p, 256)), dim3(256), 0, st, pslot, first, p, flag);
    VC_CHECK_LAUNCH("vox_creator_kernel");
    hipLaunchKernelGGL(flag_blocksum_kernel, dim3((unsigned)nb), dim3(256), 0, st, flag, p, blocksum);
    VC_CHECK_LAUNCH("flag_blocksum_kernel");
  } else {
    VC_CHECK_HIP(hipMemsetAsync(blocksum, 0, (nb + 1) * 4, st));
  }
  hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(256), 0, st, blocksum, p > 0 ? nb : 0, blocksum + nb,
                     (int32_t*)nullptr);
  VC_CHECK_LAUNCH("scan_blocksums_kernel");
  hipLaunchKernelGGL(vox_count_kernel, dim3(1), dim3(64), 0, st, blocksum + nb, max_voxels, n_voxels_dev);
  VC_CHECK_LAUNCH("vox_count_kernel");
  if (p > 0) {
    hipLaunchKernelGGL(vox_assign_kernel, dim3((unsigned)nb), dim3(256), 0, st, points, f, g, pslot, flag, blocksum, p,
                       max_voxels, cellvid, coords);
    VC_CHECK_LAUNCH("vox_assign_kernel");
    if (fused) {
      hipLaunchKernelGGL(vox_reduce_kernel, dim3((unsigned)cdiv((int64_t)cap * f, 256)), dim3(256), 0, st, points, f,
                         max_points, cellvid, slots, (int64_t)cap, vfe_max_last, features, num_points);
      VC_CHECK_LAUNCH("vox_reduce_kernel");
    } else {
      hipLaunchKernelGGL(vox_fill_kernel, dim3((unsigned)cdiv((int64_t)cap * f, 256)), dim3(256), 0, st, points, f,
                         max_points, cellvid, slots, (int64_t)cap, features, num_points);
      VC_CHECK_LAUNCH("vox_fill_kernel");
    }
  }
  return VC_OK;
}

int vc_voxelize_mean(const float* points, int64_t p, int f, const float* range, const float* vsize, int max_points,
                     int max_voxels, int vfe_max_last, void* ws, size_t ws_bytes, float* features, int32_t* coords,
                     int32_t* num_points, int32_t* n_voxels_dev, void* stream) {
  return voxelize_impl(true, points, p, f, range, vsize, max_points, max_voxels, vfe_max_last, ws, ws_bytes, features,
                       coords, num_points, n_voxels_dev, stream);
}

int vc_voxelize(const float* points, int64_t p, int f, const float* range, const float* vsize, int max_points,
                int max_voxels, void* ws, size_t ws_bytes, float* voxels, int32_t* coords, int32_t* num_points,
                int32_t* n_voxels_dev, void* stream) {
  return voxelize_impl(false, points, p, f, range, vsize, max_points, max_voxels, 0, ws, ws_bytes, voxels, coords,
                       num_points, n_voxels_dev, stream);
}

size_t vc_rep_order_workspace_bytes(int64_t n) { return n < 0 ? 0 : (size_t)(cdiv(n > 0 ? n : 1, 256) + 1) * sizeof(int32_t); }

int vc_rep_order(const int32_t* rep, int64_t n, int32_t* order, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(n >= 0 && n < (1LL << 31), "vc_rep_order: invalid row count");
  if (n == 0) return VC_OK;
  VC_REQUIRE(rep && order && ws, "vc_rep_order: null argument");
  if (ws_bytes < vc_rep_order_workspace_bytes(n)) { set_error("vc_rep_order: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = cdiv(n, 256);
  int32_t* blocksum = (int32_t*)ws;
  hipLaunchKernelGGL(rep_flag_blocksum_kernel, dim3((unsigned)nb), dim3(256), 0, st, rep, n, blocksum);
  VC_CHECK_LAUNCH("rep_flag_blocksum_kernel");
  hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(256), 0, st, blocksum, nb, blocksum + nb, (int32_t*)nullptr);
  VC_CHECK_LAUNCH("scan_blocksums_kernel");
  hipLaunchKernelGGL(rep_partition_kernel, dim3((unsigned)nb), dim3(256), 0, st, rep, n, blocksum, nb, order);
  VC_CHECK_LAUNCH("rep_partition_kernel");
  return VC_OK;
}

int vc_row_order(const int32_t* tbl, int64_t n, int kv, const int32_t* rep, int centre, int window, int32_t* order,
                 void* stream) {
  VC_REQUIRE(n >= 0 && kv >= 1 && kv <= 32, "vc_row_order: invalid argument (kv must be 1..32)");
  VC_REQUIRE(n < (1LL << 31), "vc_row_order: too many rows");
  if (n == 0) return VC_OK;
  VC_REQUIRE(tbl && order, "vc_row_order: null argument");
  VC_REQUIRE(centre >= -1 && centre < kv, "vc_row_order: centre out of range");
  hipStream_t st = (hipStream_t)stream;
  switch (window) {
    case 1024:
      hipLaunchKernelGGL((row_order_kernel<1024, 256>), dim3((unsigned)cdiv(n, 1024)), dim3(256), 0, st, tbl, n, kv, rep,
                         centre, order);
      break;
    case 2048:
      hipLaunchKernelGGL((row_order_kernel<2048, 512>), dim3((unsigned)cdiv(n, 2048)), dim3(512), 0, st, tbl, n, kv, rep,
                         centre, order);
      break;
    case 4096:
      hipLaunchKernelGGL((row_order_kernel<4096, 1024>), dim3((unsigned)cdiv(n, 4096)), dim3(1024), 0, st, tbl, n, kv, rep,
                         centre, order);
      break;
    default:
      set_error("vc_row_order: window must be 1024, 2048 or 4096");
      return VC_EINVAL;
  }
  VC_CHECK_LAUNCH("row_order_kernel");
  return VC_OK;
}

int vc_random_keep(int64_t n, int64_t n_keep, uint64_t seed, int64_t* keep, void* stream) {
  VC_REQUIRE(n >= 0 && n_keep >= 0 && n_keep <= n, "vc_random_keep: need 0 <= n_keep <= n");
  if (n_keep == 0) return VC_OK;
  VC_REQUIRE(keep, "vc_random_keep: null argument");
  int half = 1;
  while (half < 31 && (1ULL << (2 * half)) < (uint64_t)n) ++half;
  VC_REQUIRE((1ULL << (2 * half)) >= (uint64_t)n, "vc_random_keep: n too large");
  hipLaunchKernelGGL(random_keep_kernel, dim3((unsigned)cdiv(n_keep, 256)), dim3(256), 0, (hipStream_t)stream, n, n_keep,
                     seed, half, keep);
  VC_CHECK_LAUNCH("random_keep_kernel");
  return VC_OK;
}

}  // extern "C"

