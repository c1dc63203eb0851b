// Experiment (VERDICT r3 next-round #4 / "what's missing" #2): ONE gather for the input gradient AND the weight gradient of a SubM conv.
//
//   dx[j]            = sum_K dy[tbl[K][j]] * W_{kv-1-K}^T          (the backward-input gather-GEMM: rows j are the tile, dy is gathered)
//   dW_{kv-1-K}[ci][co] = sum_j x[j][ci] * dy[tbl[K][j]][co]       (the same (j, K) pairs: x[j] are the tile's OWN consecutive rows)
//
// so the 16 gathered gradient rows a wave holds for (tile, K) feed two products: G W^T (K dimension = channels, as in
// gather_gemm_v2<BWD>) and x_tile^T G (K dimension = the 16 rows).  The second one needs G with rows on the MFMA K index, i.e.
// TRANSPOSED lanes: through a wave-private LDS tile.  The 16 x 16 results of every (tile, K) are accumulated per block in LDS
// (kv x CI x CO floats: 27 KB at 16 x 16, 110 KB at 32 x 32) -- with ds_add_f32 here, i.e. the OPTIMISTIC variant: the order of the
// additions is not fixed, the product library would need an ordered scheme on top.  If this one does not beat the two separate
// launches, no deterministic version will.
//
// Stand-alone library (tools/fused_bwd_bench.py builds and drives it; not part of libvirconv_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form tools/ubench/fused_bwd.hip -o tools/ubench/libfused_bwd.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int V> struct Ld;
template <> struct Ld<4> {
  static __device__ __forceinline__ void buf(__amdgpu_buffer_rsrc_t rs, unsigned off, float* o) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  static __device__ __forceinline__ void glb(const float* p, float* o) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
};
template <> struct Ld<2> {
  static __device__ __forceinline__ void buf(__amdgpu_buffer_rsrc_t rs, unsigned off, float* o) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
    o[0] = v[0]; o[1] = v[1];
  }
  static __device__ __forceinline__ void glb(const float* p, float* o) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = *reinterpret_cast<const f32x2*>(p);
    o[0] = v[0]; o[1] = v[1];
  }
};

// weight image in MFMA B-fragment order for the dx product: wp[kw][ch][nt][lane][j] = w[co = ch*4V + q*V + j][kw][ci = 16 nt + i]
template <int CI, int CO>
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int kv) {
  constexpr int V = CO >= 16 ? 4 : CO / 4, NCH = CO / (4 * V), NTI = (CI + 15) / 16;
  const int total = kv * NCH * NTI * 64 * V;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int j = e % V, lane = (e / V) % 64, nt = (e / (V * 64)) % NTI, ch = (e / (V * 64 * NTI)) % NCH, kw = e / (V * 64 * NTI * NCH);
    const int i = lane & 15, q = lane >> 4;
    const int co = ch * 4 * V + q * V + j, ci = 16 * nt + i;
    wp[e] = (ci < CI) ? w[((int64_t)co * kv + kw) * CI + ci] : 0.f;
  }
}

template <int CI, int CO>
__global__ void __launch_bounds__(256) fused_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const int32_t* __restrict__ tbl, const float* __restrict__ wp,
                                                        float* __restrict__ dx, float* __restrict__ partial, int64_t n, int kv,
                                                        int ngroups, int do_dw) {
  constexpr int V = CO >= 16 ? 4 : CO / 4;     // floats per lane and chunk of a gathered gradient row
  constexpr int NCH = CO / (4 * V);
  constexpr int NTI = (CI + 15) / 16;          // 16-column tiles of dx = 16-row tiles of dW
  constexpr int NTO = (CO + 15) / 16;          // 16-column tiles of dW
  constexpr int TS = CO + 4;                   // row stride of the transpose tile (floats; 16-byte aligned rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_dw = reinterpret_cast<float*>(smem);                        // [kv][CI][CO]
  int* s_idx = reinterpret_cast<int*>(s_dw + kv * CI * CO);            // [kv][64]
  float* s_t = reinterpret_cast<float*>(s_idx + kv * 64);              // [4 waves][16][TS]
  unsigned* s_mask = reinterpret_cast<unsigned*>(s_t + 4 * 16 * TS);   // [2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  float* T = s_t + wave * 16 * TS;

  for (int e = tid; e < kv * CI * CO; e += 256) s_dw[e] = 0.f;
  if (tid < 2) s_mask[tid] = 0u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (int)(n * CO * 4), 0x00020000);

  // contiguous range of 64-row groups per block, blocks of one XCD (blockIdx % 8) next to each other in the row order
  const unsigned nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7u, qd = nb >> 3, rm = nb & 7u;
  const int lbid = (int)((xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3));
  const int gpb = (ngroups + (int)nb - 1) / (int)nb;
  const int g0 = lbid * gpb, g1 = min(g0 + gpb, ngroups);

  for (int g = g0; g < g1; ++g) {
    const int par = g & 1;
    __syncthreads();    // previous group's s_idx / s_mask reads are done; s_mask[par] was cleared a group ago
    if (tid == 0) s_mask[par ^ 1] = 0u;
    const int64_t brow0 = (int64_t)g * 64;
    {
      const int r = tid & 63;
      const bool inb = brow0 + r < n;
      for (int kb = tid >> 6; kb < kv; kb += 32) {
        int vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = kb + 4 * u;
          vv[u] = (inb && k < kv) ? tbl[(int64_t)k * n + brow0 + r] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = kb + 4 * u;
          if (k < kv) {
            s_idx[k * 64 + r] = vv[u];
            if (__ballot(vv[u] >= 0) != 0ULL && lane == 0) atomicOr(&s_mask[par], 1u << k);
          }
        }
      }
    }
    __syncthreads();
    unsigned bmask = (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[par]);
    const int64_t row0 = brow0 + wave * 16;

    // the tile's own rows of x as the A operand of the dW product: A[m = ci][k = row]
    float xa[4][NTI];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < NTI; ++mt) {
        const int64_t r = row0 + 4 * s + q;
        xa[s][mt] = (do_dw && r < n && 16 * mt + i < CI) ? x[r * CI + 16 * mt + i] : 0.f;
      }
    f32x4 acc[NTI];
#pragma unroll
    for (int nt = 0; nt < NTI; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    float ga[NCH][V], gb[NCH][V], ba[NCH][NTI][V], bb[NCH][NTI][V];
    float sink = 0.f;
    int acta = 0, actb = 0;

#define FB_ISSUE(K, G, B, ACT)                                                                               \
  do {   /* UNCONDITIONAL loads (index -1: out of range -> zeros, no memory access): straight-line code keeps counted vmcnt waits */ \
    const int id_ = s_idx[(K) * 64 + wave * 16 + i];                                                         \
    ACT = __builtin_amdgcn_readfirstlane((int)(__ballot(id_ >= 0) != 0ULL));                                 \
    const unsigned base_ = (unsigned)id_ * (unsigned)(CO * 4) + (unsigned)(q * V * 4);                       \
    _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch) Ld<V>::buf(rs, base_ + (unsigned)(ch * 16 * V), G[ch]); \
    const float* wk_ = wp + (int64_t)(kv - 1 - (K)) * (NCH * NTI * 64 * V);                                  \
    _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                                       \
        _Pragma("unroll") for (int nt = 0; nt < NTI; ++nt) Ld<V>::glb(wk_ + ((ch * NTI + nt) * 64 + lane) * V, B[ch][nt]); \
  } while (0)

#define FB_COMPUTE(K, G, B, ACT)                                                                             \
  do {                                                                                                       \
    if (ACT) {                                                                                               \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                                     \
          _Pragma("unroll") for (int j = 0; j < V; ++j)                                                      \
              _Pragma("unroll") for (int nt = 0; nt < NTI; ++nt)                                             \
                  acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(G[ch][j], B[ch][nt][j], acc[nt], 0, 0, 0);  \
      if (do_dw) {                                                                                           \
        _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch) {                                                 \
          float* d_ = T + i * TS + ch * 4 * V + q * V;                                                       \
          _Pragma("unroll") for (int j = 0; j < V; ++j) d_[j] = G[ch][j];                                    \
        }                                                                                                    \
        __builtin_amdgcn_wave_barrier();                                                                     \
        float bt[4][NTO];                                                                                    \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                        \
            _Pragma("unroll") for (int nt = 0; nt < NTO; ++nt)                                               \
                bt[s][nt] = (16 * nt + i < CO) ? T[(4 * s + q) * TS + 16 * nt + i] : 0.f;                    \
        __builtin_amdgcn_wave_barrier();                                                                     \
        float* dwk_ = s_dw + (kv - 1 - (K)) * (CI * CO);                                                     \
        _Pragma("unroll") for (int mt = 0; mt < NTI; ++mt)                                                   \
            _Pragma("unroll") for (int nt = 0; nt < NTO; ++nt) {                                             \
              f32x4 d_ = f32x4{0.f, 0.f, 0.f, 0.f};                                                          \
              _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                  \
                  d_ = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s][mt], bt[s][nt], d_, 0, 0, 0);              \
              _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
                const int ci_ = 16 * mt + 4 * q + r, co_ = 16 * nt + i;                                      \
                if (do_dw == 2) sink += d_[r];   /* ablation: no LDS accumulation (wrong dW) */                \
                else if (ci_ < CI && co_ < CO) atomicAdd(&dwk_[ci_ * CO + co_], d_[r]);                      \
              }                                                                                              \
            }                                                                                                \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)

    if (bmask != 0u) {
      int k0 = __ffs((int)bmask) - 1;
      bmask &= bmask - 1;
      FB_ISSUE(k0, ga, ba, acta);
      for (;;) {
        const bool more0 = bmask != 0u;
        const int k1 = more0 ? (__ffs((int)bmask) - 1) : k0;
        bmask &= bmask - 1;
        FB_ISSUE(k1, gb, bb, actb);
        FB_COMPUTE(k0, ga, ba, acta);
        if (!more0) break;
        const bool more1 = bmask != 0u;
        k0 = more1 ? (__ffs((int)bmask) - 1) : k1;
        bmask &= bmask - 1;
        FB_ISSUE(k0, ga, ba, acta);
        FB_COMPUTE(k1, gb, bb, actb);
        if (!more1) break;
      }
    }
#undef FB_ISSUE
#undef FB_COMPUTE
#pragma unroll
    for (int nt = 0; nt < NTI; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * q + r;
        if (row < n && 16 * nt + i < CI) dx[row * CI + 16 * nt + i] = acc[nt][r];
      }
    if (do_dw == 2 && sink == 123.456f) dx[0] = sink;
  }
  __syncthreads();
  if (do_dw) {
    float* dst = partial + (int64_t)blockIdx.x * kv * CI * CO;
    for (int e = tid; e < kv * CI * CO; e += 256) dst[e] = s_dw[e];
  }
}

// dW[co][kw][ci] = sum over blocks of partial[b][kw][ci][co].  Block = 16 consecutive elements x 16 segments of the block index
// (segment s adds blocks s, s + 16, ... in that order, 8 loads in flight; the 16 segment sums are combined in segment order).
__global__ void __launch_bounds__(256) reduce_kernel(const float* __restrict__ partial, int nblocks, int kv, int ci_n, int co_n,
                                                     float* __restrict__ dw) {
  __shared__ float red[256];
  const int total = kv * ci_n * co_n;
  const int el = threadIdx.x & 15, seg = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < total) {
    int b = seg;
    for (; b + 7 * 16 < nblocks; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 16 * u) * total + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblocks; b += 16) s += partial[(int64_t)b * total + e];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (seg == 0 && e < total) {
    float t = 0.f;
    for (int u = 0; u < 16; ++u) t += red[u * 16 + el];
    const int co = e % co_n, ci = (e / co_n) % ci_n, kw = e / (co_n * ci_n);
    dw[((int64_t)co * kv + kw) * ci_n + ci] = t;
  }
}

template <int CI, int CO>
int launch(const float* dy, const float* x, const int32_t* tbl, const float* w, float* wp, float* dx, float* partial, float* dw,
           int64_t n, int kv, int nblocks, int do_dw, hipStream_t st) {
  constexpr int TS = CO + 4;
  const size_t lds = (size_t)kv * CI * CO * 4 + (size_t)kv * 64 * 4 + 4 * 16 * TS * 4 + 16;
  hipLaunchKernelGGL((pack_kernel<CI, CO>), dim3(32), dim3(256), 0, st, w, wp, kv);
  const int ngroups = (int)((n + 63) / 64);
  if (lds > 64 * 1024)
    if (hipFuncSetAttribute((const void*)fused_bwd_kernel<CI, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 2;
  hipLaunchKernelGGL((fused_bwd_kernel<CI, CO>), dim3(nblocks), dim3(256), lds, st, dy, x, tbl, wp, dx, partial, n, kv, ngroups, do_dw);
  if (do_dw) {
    const int total = kv * CI * CO;
    hipLaunchKernelGGL(reduce_kernel, dim3((total + 15) / 16), dim3(256), 0, st, partial, nblocks, kv, CI, CO, dw);
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

}  // namespace

extern "C" {

// bytes of `wp` (packed weights) and of `partial` for `nblocks` blocks
size_t fb_wp_bytes(int kv, int ci, int co) { return (size_t)kv * ((ci + 15) / 16) * 64 * (co / 4) * 4 + 256; }
size_t fb_partial_bytes(int kv, int ci, int co, int nblocks) { return (size_t)nblocks * kv * ci * co * 4; }
size_t fb_lds_bytes(int kv, int ci, int co) { return (size_t)kv * ci * co * 4 + (size_t)kv * 64 * 4 + 4 * 16 * (co + 4) * 4 + 16; }

// dy (n, co), x (n, ci), tbl (kv, n) SubM pair table, w (co, kv, ci).  -> dx (n, ci), dw (co, kv, ci).  do_dw = 0: the dx half alone.
int fb_fused_bwd(const float* dy, const float* x, const int32_t* tbl, const float* w, float* wp, float* dx, float* partial, float* dw,
                 int64_t n, int kv, int ci, int co, int nblocks, int do_dw, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define FB_CASE(A, B) if (ci == A && co == B) return launch<A, B>(dy, x, tbl, w, wp, dx, partial, dw, n, kv, nblocks, do_dw, st)
  FB_CASE(8, 8);
  FB_CASE(16, 16);
  FB_CASE(32, 16);
  FB_CASE(32, 32);
#undef FB_CASE
  return 1;
}

}  // extern "C"
