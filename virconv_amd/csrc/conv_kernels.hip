// Feature-side kernels of the VirConv hot path for gfx950:
//   K6/K7  output-stationary gather-GEMM (forward and backward-input) on v_mfma_f32_16x16x4_f32
//   K8     weight gradient: wave-ballot compaction of active pairs + MFMA outer products + 2-stage reduction
//   group-sum for the duplicate-coordinate (2-D image-space) SubM backward
//
// MFMA mapping (wave64, v_mfma_f32_16x16x4_f32; A[i=l&15][k=l>>4], B[k=l>>4][n=l&15], D[row=(l>>4)*4+reg][col=l&15]):
//   M = 16 output rows of the tile, N = 16 output channels, K = 4 source channels per instruction.
//   Lane (i, q) gathers V contiguous source channels [ch*4V + q*V, +V) of row tbl[k][row0+i] with ONE vector load
//   (64 contiguous bytes per gathered row across the 4 q-lanes) and feeds element j to MFMA step j; the weight
//   fragment is permuted identically, so the K order inside a chunk is (j, q) -- a fixed order, results are
//   run-to-run bit-stable.  fp32 MFMA is exact fp32 (bitwise an fmaf chain), so the 1e-4 parity bound holds.
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <atomic>
#include <mutex>

#include "common.h"

namespace vc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ int g_xcd_swizzle_off = 0;  // developer switch (tools/kbench.py --no-xcd)
#ifdef VC_EXPERIMENTS
__device__ int g_pc_ablate = 0;        // developer ablations of the pair-compacted kernel (vc_debug_set conv_pc_ablate; wrong results): 1 no LDS adds | 2 no gathers | 4 no MFMAs | 8 no per-offset barrier
#endif
// v2 MFMA phase order.  0 (default) = the round-1 order: hipcc reads each B fragment right before its four MFMAs, 68 VGPRs,
// 6 waves per SIMD.  1 = every B fragment of an offset read before its first MFMA + accumulator-alternating MFMAs: better
// per-wave code but 97 VGPRs / 4 waves per SIMD, and MEASURED SLOWER (s3.d3_conv1 64->32 forward 242 vs 208 us,
// profiles/r02_kbench_variants.txt): this kernel waits ~2 us for every W_k it requested one iteration earlier, and only
// resident waves hide that.  Kept for A/B builds (hipcc -DVC_V2_SCHED=1).  2 = only the MFMA ORDER of 1 (consecutive MFMAs
// alternate between the output-column accumulators, no back-to-back dependent pair and its 8-cycle pipe bubble), fragment
// reads left where hipcc wants them: same register count as 0 and the same time (bench 6.12-6.17 ms/step both).
#ifndef VC_V2_SCHED
#define VC_V2_SCHED 0
#endif

// Reduced-precision MFMA operands (BASELINE configs[4]: "fp16 MFMA contraction"): features / gradients / weights stay fp32
// in HBM, are rounded (RNE) to fp16 or bf16 in registers right before the MFMA, products are exact and accumulate in fp32.
//   OT = VC_OPERAND_F32 : v_mfma_f32_16x16x4_f32     (exact fp32, the default and the 1e-4 parity path)
//   OT = VC_OPERAND_F16 : v_mfma_f32_16x16x16_f16    (4 K-steps per instruction, 8x the fp32 MFMA rate)
//   OT = VC_OPERAND_BF16: v_mfma_f32_16x16x16_bf16   (same shape; fp32 exponent range, for gradients that underflow fp16)
// A lane's 4 consecutive channels [ch*16 + 4q, +4) are exactly the 4 K-slots k = 4q..4q+3 the 16x16x16 layouts want, so
// the gather pattern, the W fragment order and the accumulator layout are the same as in the fp32 kernels.
template <int OT>
struct Pack4;
template <>
struct Pack4<VC_OPERAND_F16> {
  static __device__ __forceinline__ u32x2 cvt(const float* f) {
    const f32x4 v = {f[0], f[1], f[2], f[3]};
    return __builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4));
  }
};
template <>
struct Pack4<VC_OPERAND_BF16> {
  static __device__ __forceinline__ u32x2 cvt(const float* f) {
    const f32x2 lo = {f[0], f[1]}, hi = {f[2], f[3]};
    u32x2 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
    return r;
  }
};
template <int OT>
__device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
  if constexpr (OT == VC_OPERAND_F16) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
}

// fp32-accurate products on the bf16 matrix cores (round 4, "f32_split"): x = h + m + l EXACTLY, three bf16 values of 8 significant bits
// each obtained by rounding to nearest even (v_cvt_pk_bf16_f32; the differences x - h and (x - h) - m are exact in fp32, |m| <= 2^-8 |x|,
// |l| <= 2^-16 |x|), and x y ~ l_x h_y + h_x l_y + m_x m_y + m_x h_y + h_x m_y + h_x h_y: the three dropped terms together stay below
// 2^-24 |x y|, HALF an fp32 ulp of the product (oracle/split_ref.py, tests/test_split_cpu.py; a truncating split would drop up to 2^-20).
// Six v_mfma_f32_16x16x32_bf16 (~17 cycles each per SIMD) per 32 channels instead of eight v_mfma_f32_16x16x4_f32 (32 cycles each):
// 0.4x the matrix-pipe time, paid for with ~5.5 VALU operations per gathered element.  Tensors and accumulation stay fp32.
static constexpr int VC_OPERAND_X6 = 3;   // internal operand type: selected by vc_debug_set f32_split for operand_type VC_OPERAND_F32
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk_bf16_rne(float a, float b) {   // a in the low half
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3(const float* f, u32x2& h, u32x2& m, u32x2& l) {
  h.x = pk_bf16_rne(f[0], f[1]);
  h.y = pk_bf16_rne(f[2], f[3]);
  const float r0 = f[0] - __uint_as_float(h.x << 16), r1 = f[1] - __uint_as_float(h.x & 0xFFFF0000u);   // exact
  const float r2 = f[2] - __uint_as_float(h.y << 16), r3 = f[3] - __uint_as_float(h.y & 0xFFFF0000u);
  m.x = pk_bf16_rne(r0, r1);
  m.y = pk_bf16_rne(r2, r3);
  const float s0 = r0 - __uint_as_float(m.x << 16), s1 = r1 - __uint_as_float(m.x & 0xFFFF0000u);       // exact, bf16 values
  const float s2 = r2 - __uint_as_float(m.y << 16), s3 = r3 - __uint_as_float(m.y & 0xFFFF0000u);
  l.x = pk_bf16_rne(s0, s1);
  l.y = pk_bf16_rne(s2, s3);
}
__device__ __forceinline__ f32x4 mfma32bf(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int V>
struct VecLoad;
template <>
struct VecLoad<4> {
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
};
template <>
struct VecLoad<2> {
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    float2 v = *reinterpret_cast<const float2*>(p);
    o[0] = v.x; o[1] = v.y;
  }
};
template <>
struct VecLoad<1> {
  static __device__ __forceinline__ void ld(const float* p, float* o) { o[0] = *p; }
};

// out[o, :] = sum_k src_k[tbl[k, o], :] @ Wsel(k)       CK = source channels (GEMM K), CN = output channels (GEMM N)
// BWD = false: Wsel(k)[kk][n] = w[(n*kv + kw)*CK + kk]   (w = (Cout=CN, KV, Cin=CK))
// BWD = true : Wsel(k)[kk][n] = w[(kk*kv + kw)*CN + n]   (w = (Cout=CK, KV, Cin=CN))
template <int CK, int CN, bool BWD, int RT>
__global__ void __launch_bounds__(256) gather_gemm_kernel(const float* __restrict__ src,
                                                          const float* __restrict__ src_centre,
                                                          const int32_t* __restrict__ tbl, const float* __restrict__ w,
                                                          float* __restrict__ out, const int32_t* __restrict__ rep,
                                                          int64_t n_out, int kv, int centre, int mirror) {
  constexpr int V = (CK >= 16) ? 4 : CK / 4;
  constexpr int NCH = CK / (4 * V);
  constexpr int NT = (CN + 15) / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (RT * 16);
  if (row0 >= n_out) return;  // wave-uniform

  f32x4 acc[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  bool centre_only[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int64_t r = row0 + t * 16 + i;
    centre_only[t] = (rep != nullptr) && (r < n_out) && (rep[r] != (int32_t)r);
  }

  for (int k = 0; k < kv; ++k) {
    int id[RT];
    bool anyv = false;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int64_t r = row0 + t * 16 + i;
      int v = (r < n_out) ? tbl[(int64_t)k * n_out + r] : -1;
      if (centre_only[t] && k != centre) v = -1;
      id[t] = v;
      anyv |= (v >= 0);
    }
    if (__ballot(anyv) == 0ULL) continue;  // no active pair in this wave's rows for offset k
    const float* __restrict__ S = (k == centre && src_centre != nullptr) ? src_centre : src;
    const int kw = mirror ? (kv - 1 - k) : k;
    unsigned long long tmask[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) tmask[t] = __ballot(id[t] >= 0);

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      float b[NT][V];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + i;
        if (n < CN) {
          if (!BWD) {
            VecLoad<V>::ld(w + ((int64_t)n * kv + kw) * CK + ch * 4 * V + q * V, b[nt]);
          } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
              const int kk = ch * 4 * V + q * V + j;
              b[nt][j] = w[((int64_t)kk * kv + kw) * CN + n];
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) b[nt][j] = 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        if (tmask[t] == 0ULL) continue;  // wave-uniform
        float a[V];
        if (id[t] >= 0) {
          VecLoad<V>::ld(S + (int64_t)id[t] * CK + ch * 4 * V + q * V, a);
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) a[j] = 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < V; ++j)
            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[nt][j], acc[t][nt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + i;
      if (n >= CN) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t r = row0 + t * 16 + q * 4 + reg;
        if (r < n_out) out[r * CN + n] = acc[t][nt][reg];
      }
    }
}

// --------------------------------------------------------------------------------------------- fragment-ordered weight images
// Measured (tools/ubench/gather_ubench.hip, MI355X): a wave that reads its W_k fragments from the canonical (Cout, KV, Cin) layout
// moves 9.3 TB/s over the chip -- a quad of consecutive lanes touches four weight rows 27*Cin*4 bytes apart, and the vector
// memory pipeline serves one cache line per quad and clock -- against 30 TB/s when every instruction reads 1 KB of contiguous
// memory.  The gather-GEMMs re-read W_k once per 64 output rows and offset, 0.53 GB per launch of the 64 -> 32 SubM layer, i.e.
// 57 us of the memory pipeline per launch against 17 us.  So the weights are repacked ONCE per step and direction into the order
// the MFMA B fragments want (kernel offset k, 16-channel K chunk ch, 16-column N tile nt, lane, 4 floats):
//   packed[(((k * NCH + ch) * NT + nt) * 64 + lane) * 4 + j] = Wsel(k)[ch*16 + (lane >> 4)*4 + j][nt*16 + (lane & 15)]
// (columns >= CN are zero).  spconv's implicit-GEMM path also keeps a reordered copy of the filters; here it lives in caller
// memory (vc_conv_pack_weights) and is looked up by the canonical weight pointer when a conv is launched.
struct PackDesc {
  const float* w;
  float* dst;
  int ck, cn, kv, bwd;
};
static constexpr int kMaxPack = 48;
struct PackArgs {
  PackDesc d[kMaxPack];
};
__global__ void __launch_bounds__(256) pack_weights_kernel(PackArgs a) {
  const PackDesc d = a.d[blockIdx.y];
  const int nch = d.ck / 16, nt_n = (d.cn + 15) / 16;
  const int total = d.kv * nch * nt_n * 256;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int j = e & 3, lane = (e >> 2) & 63;
    int f = e >> 8;
    const int nt = f % nt_n; f /= nt_n;
    const int ch = f % nch;
    const int k = f / nch;
    const int kk = ch * 16 + (lane >> 4) * 4 + j, n = nt * 16 + (lane & 15);
    float v = 0.f;
    if (n < d.cn) v = d.bwd ? d.w[((int64_t)kk * d.kv + k) * d.cn + n] : d.w[((int64_t)n * d.kv + k) * d.ck + kk];
    d.dst[e] = v;
  }
}

// canonical weight pointer -> fragment-ordered image (per thread: the pass executor registers before it launches and clears after)
struct PackedEntry { const float* w; const float* packed; int bwd; };
static thread_local PackedEntry g_packed[2 * kMaxPack];
static thread_local int g_n_packed = 0;
int g_conv_use_packed = 1;  // vc_debug_set conv_packed: 0 = ignore registered images (A/B)
static inline const float* packed_lookup(const float* w, bool bwd) {
  if (!g_conv_use_packed) return nullptr;
  for (int i = 0; i < g_n_packed; ++i)
    if (g_packed[i].w == w && g_packed[i].bwd == (bwd ? 1 : 0)) return g_packed[i].packed;
  return nullptr;
}

// --------------------------------------------------------------------------------------------- K6/K7 v2 (pipelined)
// Same math and the same fixed accumulation order as gather_gemm_kernel, restructured so that the matrix pipe is not
// parked behind dependent loads:
//   * the block's (KV x 128) slice of the pair table is staged into LDS once (one coalesced burst, one latency) and the
//     per-wave / per-block activity of every kernel offset is reduced to two 32-bit masks (ballot + LDS atomicOr)
//   * the block walks ONLY the offsets active somewhere in its 128 rows; for each, the weight slice W_k is staged once
//     per block into LDS in MFMA-fragment order (4x fewer VMEM instructions than per-wave loads; conflict-free
//     ds_read_b128), double buffered
//   * gathers go through a buffer descriptor: row index -1 wraps to an out-of-range offset and the hardware returns
//     zeros, so every load is UNCONDITIONAL (no exec-masked branches => hipcc keeps counted vmcnt waits) and the
//     gathers + W_k loads of the NEXT active offset are in flight under the MFMAs of the current one
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int V>
struct BufLoad;
template <>
struct BufLoad<4> {
  static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float* o) {
    i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    o[0] = __int_as_float(v.x); o[1] = __int_as_float(v.y); o[2] = __int_as_float(v.z); o[3] = __int_as_float(v.w);
  }
};
template <>
struct BufLoad<2> {
  static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float* o) {
    i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    o[0] = __int_as_float(v.x); o[1] = __int_as_float(v.y);
  }
};
template <>
struct BufLoad<1> {
  static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float* o) {
    o[0] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
  }
};

// Epilogue of the forward conv (EPI): what happens to the accumulators besides the plain store
//   VC_EPI_NONE    nothing
//   VC_EPI_STATS   training-mode BatchNorm statistics: every WAVE also writes, per output channel, the sum and the sum of
//                  squares of its 16 rows to epi.partial[block * 4 + wave][2][CN] (fixed order: 4 accumulator rows, q lanes);
//                  vc_bn_stats_from_partial adds them up in fp64.  Saves the read-back pass of bn_reduce.
//   VC_EPI_AFFINE  eval-mode BatchNorm (+ReLU) folded into the store: y = acc * (gamma * istd) + (beta - mean * gamma * istd)
//   VC_EPI_BWD     backward-input conv only: `addend` (optional, a second gradient contribution given as a strided view) is added
//                  to the tile before it is stored, and (optional, y_raw != NULL) the BatchNorm-backward sums of the unit that
//                  PRODUCED this conv's input -- sum(dy_masked), sum(dy_masked * xhat) per channel, per-wave partial rows as in
//                  STATS -- are formed from the stored tile and that unit's pre-BatchNorm output: the unit's separate reduction
//                  pass over (y_raw, dy) and the gradient-add kernel disappear
// In-kernel finish of the BatchNorm sums (round 3): the v2 kernel's STATS / BWD epilogue can carry its partial sums all the way to
// the finished per-channel numbers, instead of leaving per-wave partial rows to bn_partial_reduce_kernel + a finalize kernel -- two
// 5-6 us launches (and two kernel boundaries) per BatchNorm on the main stream's critical path, 55 of them per train step.
// Three levels, arrival tickets, no spinning (MI355X guide, "in-launch split-K reduction", write-through form):
//   level 0  every block adds the rows of its 4 / 8 waves through LDS (fp64, wave order) and stores ONE fp64 row write-through (sc1);
//   level 1  blocks are grouped rpb = ceil(blocks / 64) at a time; the block that draws a group's last ticket adds the group's
//            block rows (fp64; thread = (row sub-index, column), rows ascending, then the sub-indices ascending) and stores the
//            fp64 group row write-through;
//   level 2  the block that draws the launch's last ticket adds the <= 64 group rows per column in a fixed halving tree
//            (v[u] += v[u + h], h = 32, 16, ... 1 over the group index) -- one row per lane, all channels at once -- writes the
//            finished numbers and clears the tickets.
// Which block does a reduction varies from run to run, WHAT it computes does not: results are run-to-run stable and bit-identical
// to bn_finish_reference_kernel, a single block that performs the same three levels one after the other (tests).
struct ConvFinish {
  unsigned* cnt = nullptr;     // [0]: launch tickets, [1 + g]: tickets of group g; nullptr: no in-kernel finish
  double* brows = nullptr;     // [nblocks][2 CN] fp64 block rows (in the launch's partial buffer: 2 x 4 bytes <= waves x 4 bytes)
  double* dpartial = nullptr;  // [ngroups][2 CN]
  float *o0 = nullptr, *o1 = nullptr, *r0 = nullptr, *r1 = nullptr;   // see BnFinishRequest
  long long* nbt = nullptr;
  long long n = 0;
  float momentum = 0.f;
  int bwd = 0;
  int rpb = 0, ngroups = 0, nblocks = 0;   // blocks per group, groups (<= 64), blocks of the launch
};

struct ConvEpilogue {
  float* partial;        // STATS / BWD
  const float* mean;     // AFFINE (running statistics) / BWD (batch statistics of the producing unit)
  const float* var;
  const float* gamma;
  const float* beta;
  float eps;
  int relu;
  const float* y_raw = nullptr;    // BWD: pre-BatchNorm output of the producing unit, (rows of this launch) x CN
  const float* addend = nullptr;   // BWD: out += addend[:, add_col0 : add_col0 + CN] (row stride add_stride)
  int add_stride = 0, add_col0 = 0;
  ConvFinish fin{};                // STATS / BWD statistics: finish the sums inside this launch
};

__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_load sc1: past this CU's L1
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // global_store sc1: write-through
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the three levels as device functions, shared by the ticketed tail and by the single-block reference kernel
// level 0: column `col` of a block row = the block's per-wave rows added in wave order
template <int C2, int NWV>
__device__ __forceinline__ double fin_level0(const float* rows /* [NWV][C2] */, int col) {
  double t = (double)rows[col];   // fp64 from the 16-row wave sums up, as on the two-launch route
#pragma unroll
  for (int w_ = 1; w_ < NWV; ++w_) t += (double)rows[w_ * C2 + col];
  return t;
}
// level 1, thread part: block rows r0 + rs, r0 + rs + RS, ... < r1 of column col, ascending, fp64
template <int C2>
__device__ __forceinline__ double fin_level1_part(const double* __restrict__ brows, int r0, int r1, int rs, int col) {
  constexpr int RS = 256 / C2;
  double acc = 0.0;
  int r = r0 + rs;
  for (; r + 7 * RS < r1; r += 8 * RS) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld_agent(brows + (int64_t)(r + u * RS) * C2 + col);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; r + 3 * RS < r1; r += 4 * RS) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ld_agent(brows + (int64_t)(r + u * RS) * C2 + col);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  for (; r < r1; r += RS) acc += ld_agent(brows + (int64_t)r * C2 + col);
  return acc;
}
template <int C2>
__device__ __forceinline__ double fin_level1_combine(const double* s_red /* [RS][C2] */, int col) {
  constexpr int RS = 256 / C2;
  double t = s_red[col];
#pragma unroll
  for (int u = 1; u < RS; ++u) t += s_red[u * C2 + col];
  return t;
}
// level 2, thread part: the halving tree over the group rows l = p, p + RS, ... (a subtree of the tree over 64 rows), column col
template <int C2>
__device__ __forceinline__ double fin_level2_part(const double* __restrict__ dpartial, int ngroups, int p, int col) {
  constexpr int RS = 256 / C2;
  constexpr int NL = 64 / RS;          // rows of this part: 2 (C2 = 8) ... 32 (C2 = 128)
  double v[NL >= 32 ? NL / 4 : NL];
  if constexpr (NL >= 32) {            // the two upper tree levels folded into the loading, two results per batch: 8 loads in flight
#pragma unroll
    for (int ub = 0; ub < NL / 8; ++ub) {
      double t[2][4];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int l = p + RS * (ub * 2 + k + j * (NL / 4));
          t[k][j] = l < ngroups ? ld_agent(dpartial + (int64_t)l * C2 + col) : 0.0;
        }
#pragma unroll
      for (int k = 0; k < 2; ++k) v[ub * 2 + k] = (t[k][0] + t[k][2]) + (t[k][1] + t[k][3]);   // (u, u + 16) then (+ 8)
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = NL / 8; h >= 1; h >>= 1)
#pragma unroll
      for (int u = 0; u < h; ++u) v[u] += v[u + h];
  } else {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int l = p + RS * u;
      v[u] = l < ngroups ? ld_agent(dpartial + (int64_t)l * C2 + col) : 0.0;
    }
#pragma unroll
    for (int h = NL / 2; h >= 1; h >>= 1)
#pragma unroll
      for (int u = 0; u < h; ++u) v[u] += v[u + h];
  }
  return v[0];
}
template <int C2>
__device__ __forceinline__ double fin_level2_combine(const double* s_part /* [RS][C2] */, int col) {
  constexpr int RS = 256 / C2;
  double w_[RS];
#pragma unroll
  for (int p = 0; p < RS; ++p) w_[p] = s_part[p * C2 + col];
#pragma unroll
  for (int h = RS / 2; h >= 1; h >>= 1)
#pragma unroll
    for (int p = 0; p < h; ++p) w_[p] += w_[p + h];
  return w_[0];
}
template <int CN>
__device__ __forceinline__ void fin_write(const ConvFinish& f, int ch, double s, double ss) {
  if (f.bwd) {
    if (f.o0) f.o0[ch] = (float)s;     // dbeta
    if (f.o1) f.o1[ch] = (float)ss;    // dgamma
    f.r0[ch] = (float)s;               // sums[0][ch]
    f.r0[CN + ch] = (float)ss;         // sums[1][ch]
  } else {
    const double m = s / (double)f.n;
    double v = ss / (double)f.n - m * m;
    if (v < 0.0) v = 0.0;
    f.o0[ch] = (float)m;
    f.o1[ch] = (float)v;
    if (f.r0) f.r0[ch] = (1.f - f.momentum) * f.r0[ch] + f.momentum * (float)m;
    if (f.r1) {
      const double unb = (f.n > 1) ? v * (double)f.n / (double)(f.n - 1) : v;
      f.r1[ch] = (1.f - f.momentum) * f.r1[ch] + f.momentum * (float)unb;
    }
    if (ch == 0 && f.nbt) *f.nbt += 1;
  }
}

// Tail of a v2 block whose launch finishes the BatchNorm sums (see ConvFinish), in two parts so that the block's output stores can
// sit between them: the ticket part must not wait for them (a write-through store is followed by vmcnt(0), which would also drain
// every output store issued before it: 1-2 us per block with all of its waves parked at a barrier).
// conv_finish_ticket: called by ALL threads once the wave's partial sums are final; (pa, pb)[nt] = this wave's partial sums of
// columns nt * 16 + i (valid in the lanes q == 0); `smem` = the block's dynamic LDS (>= 4 KB + 16 B; the caller must have read
// whatever it still needs from it).  Returns (block-uniform) whether this block drew the last ticket of its group.
template <int CN, int NTHR, int NT>
__device__ __forceinline__ int conv_finish_ticket(const ConvFinish& f, int lbid, int wave, const float (&pa)[NT], const float (&pb)[NT],
                                                  unsigned char* smem) {
  constexpr int C2 = 2 * CN;
  constexpr int NWV = NTHR / 64;
  static_assert(C2 <= 128 && C2 >= 16 && (C2 & (C2 - 1)) == 0, "in-kernel BatchNorm finish: power-of-two channel count 8 .. 64");
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int i = lane & 15, q = lane >> 4;
  float* s_rows = reinterpret_cast<float*>(smem);           // [NWV][C2]  (level 0)
  int* s_flag = reinterpret_cast<int*>(smem + 4096);
  __syncthreads();                                          // every wave is done with the main loop's LDS
  if (q == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + i;
      if (n < CN) { s_rows[wave * C2 + n] = pa[nt]; s_rows[wave * C2 + CN + n] = pb[nt]; }
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int grp = lbid / f.rpb;
    const int g_r0 = grp * f.rpb, g_r1 = min(g_r0 + f.rpb, f.nblocks);
    for (int col = lane; col < C2; col += 64) st_agent(f.brows + (int64_t)lbid * C2 + col, fin_level0<C2, NWV>(s_rows, col));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the block row has left (write-through)
    if (lane == 0) {
      const unsigned t = __hip_atomic_fetch_add(&f.cnt[1 + grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t + 1u == (unsigned)(g_r1 - g_r0)) ? 1 : 0;
      if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_flag[0] = last;
    }
  }
  __syncthreads();
  return s_flag[0];
}
// conv_finish_reduce: called by ALL threads of a block whose conv_finish_ticket returned 1: level 1 of its group and, if the group
// is the launch's last one to finish, level 2.
template <int CN, int NTHR>
__device__ __forceinline__ void conv_finish_reduce(const ConvFinish& f, int lbid, int wave, unsigned char* smem) {
  constexpr int C2 = 2 * CN;
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid = wave * 64 + lane;
  double* s_red = reinterpret_cast<double*>(smem);          // [RS][C2]
  double* s_tot = reinterpret_cast<double*>(smem + 2048);   // [C2]
  int* s_flag = reinterpret_cast<int*>(smem + 4096);
  const int grp = lbid / f.rpb;
  const int g_r0 = grp * f.rpb, g_r1 = min(g_r0 + f.rpb, f.nblocks);
  __syncthreads();   // every thread has read the ticket flag
  // ---- level 1
  if (tid < 256) s_red[tid] = fin_level1_part<C2>(f.brows, g_r0, g_r1, tid / C2, tid % C2);
  __syncthreads();
  if (tid < C2) st_agent(f.dpartial + (int64_t)grp * C2 + tid, fin_level1_combine<C2>(s_red, tid));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(&f.cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t + 1u == (unsigned)f.ngroups) ? 1 : 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_flag[0] = last;
  }
  __syncthreads();
  if (s_flag[0] == 0) return;
  // ---- level 2
  if (tid < 256) s_red[tid] = fin_level2_part<C2>(f.dpartial, f.ngroups, tid / C2, tid % C2);
  __syncthreads();
  if (tid < C2) s_tot[tid] = fin_level2_combine<C2>(s_red, tid);
  __syncthreads();
  if (tid < CN) fin_write<CN>(f, tid, s_tot[tid], s_tot[CN + tid]);
  for (int j = tid; j <= f.ngroups; j += NTHR) f.cnt[j] = 0u;   // every ticket of this launch has been drawn
}

// The same three levels by ONE block, one after the other, from the per-wave partial rows a launch WITHOUT the finish leaves
// behind (tests: vc_debug_set conv_bn_finish = 2).  brows / dpartial: scratch as in ConvFinish.
template <int CN>
__global__ void __launch_bounds__(256) bn_finish_reference_kernel(ConvFinish f, const float* __restrict__ wave_rows, int nwv) {
  constexpr int C2 = 2 * CN;
  __shared__ double s_red[256];
  __shared__ double s_tot[C2];
  const int tid = threadIdx.x;
  for (int b = 0; b < f.nblocks; ++b) {
    // in place: block b's fp64 row covers floats [2 b C2, 2 (b + 1) C2) of the buffer, i.e. wave rows of blocks <= b / 2 (b = 0:
    // its own first two wave rows, read just before)
    double t = 0.0;
    if (tid < C2) {
      const float* rows = wave_rows + (int64_t)b * nwv * C2;
      t = (nwv == 8) ? fin_level0<C2, 8>(rows, tid) : fin_level0<C2, 4>(rows, tid);
    }
    __syncthreads();
    if (tid < C2) f.brows[(int64_t)b * C2 + tid] = t;
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  for (int g = 0; g < f.ngroups; ++g) {
    const int r0 = g * f.rpb, r1 = min(r0 + f.rpb, f.nblocks);
    s_red[tid] = fin_level1_part<C2>(f.brows, r0, r1, tid / C2, tid % C2);
    __syncthreads();
    if (tid < C2) f.dpartial[(int64_t)g * C2 + tid] = fin_level1_combine<C2>(s_red, tid);
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  s_red[tid] = fin_level2_part<C2>(f.dpartial, f.ngroups, tid / C2, tid % C2);
  __syncthreads();
  if (tid < C2) s_tot[tid] = fin_level2_combine<C2>(s_red, tid);
  __syncthreads();
  if (tid < CN) fin_write<CN>(f, tid, s_tot[tid], s_tot[CN + tid]);
}
static constexpr int VC_EPI_BWD = 3;  // internal (not part of vc_epilogue: selected by vc_conv_backward_input_epilogue)

// DXS ("dx shift", 27-offset tables, one tile per wave): the offsets of one (dz, dy) group are the dx = -1 / 0 / +1 taps.  Where
// the output rows of a tile are x-adjacent voxels -- the normal case on a tensor whose rows are in ascending (b, z, y, x) order:
// 81-90 % of the dx = +-1 rows at stages 2-4, tests/analysis_dx_shift.py -- the row a lane needs for dx = -1 is the row its left
// neighbour lane needs for dx = 0 (tbl[3g][i] == tbl[3g+1][i-1]), and for dx = +1 its right neighbour's.  So the centre
// fragment of a group is gathered ONCE and kept; a side offset takes it shifted by one lane inside the 16-lane MFMA row (DPP
// row_shr / row_shl) wherever the table entries agree, and gathers only the lanes that disagree (index -1 for the others: no
// memory access; no instruction at all when every lane agrees).  The check is per lane on the table itself, so any table is
// handled correctly; the arithmetic -- operands, MFMA order -- is unchanged, results are bit-identical.  What it buys is the
// CU's vector-memory pipeline: 1.3-1.5 gathered rows per (tile, side offset) instead of 7-13.
// IL ("interleaved source", round-3 experiment, VC_CONV_SRC_INTERLEAVED): `src` is stored in 16-row groups, chunk-major inside a
// group -- [row / 16][channel / 4][row % 16][4 floats] -- so that the 16 lanes of one MFMA row-slot quarter (same K chunk q, rows
// i = 0..15) read 256 CONTIGUOUS bytes when their rows are consecutive (x-adjacent voxels of a sorted tensor), i.e. 4 cache lines
// per quarter instead of 16: the L1 tag look-ups per gather instruction are what paces this kernel's vector-memory pipeline
// (8.0-8.7 TB/s for the row-major MFMA mapping against 14.5 TB/s when a quad of lanes shares a line).
// Occupancy target of an instantiation (waves per SIMD the register allocation must allow; 1 = no requirement).  The epilogue
// kernels with 64 output channels sat on or just below the 64-VGPR line (8 waves) before the in-kernel BatchNorm finish was
// added; with the tail inlined the allocator takes 74.  Pinned back: the main loop is unaffected, the tail (run by every
// block once, its reduction levels by a few blocks per launch) spills 44-52 bytes (tools/vgpr_table.py).
template <int CK, int CN, int EPI, int NW, bool PK, bool DXS, int OT = VC_OPERAND_F32>
constexpr int v2_min_waves() {
  if (OT != VC_OPERAND_F32) return 1;
  if (DXS || CN != 64 || !(EPI == 1 /* STATS */ || EPI == 3 /* VC_EPI_BWD */)) return 1;
  if (CK <= 32) return (EPI == 3 && CK == 32 && NW == 4 && !PK) ? 1 : 8;   // were 42-64 VGPRs without the tail
  return (NW == 8 && PK) ? 7 : 1;                                           // 64 -> 64, 8 waves: was 72
}

template <int CK, int CN, bool BWD, int RT, int OT, int EPI, int NW = 4, bool PK = false, bool DXS = false, bool IL = false>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(v2_min_waves<CK, CN, EPI, NW, PK, DXS, OT>())))
gather_gemm_v2_kernel(const float* __restrict__ src,
                                                             const float* __restrict__ src_centre, int64_t n_src,
                                                             const int32_t* __restrict__ tbl,
                                                             const float* __restrict__ w, float* __restrict__ out,
                                                             const int32_t* __restrict__ rep,
                                                             const int32_t* __restrict__ order, int64_t n_out, int kv,
                                                             int centre, int mirror, ConvEpilogue epi) {
  static_assert(EPI == VC_EPI_NONE || RT == 1, "epilogues exist for the one-tile-per-wave kernel only");
  static_assert(!PK || (CK % 16 == 0 && (OT == VC_OPERAND_F32 || OT == VC_OPERAND_X6)), "packed weight images: fp32 operands, 16-channel K chunks");
  static_assert(OT != VC_OPERAND_X6 || (RT == 1 && CK % 16 == 0 && !IL), "split-bf16 products: one tile per wave, 16-channel K chunks");
  static_assert(!DXS || (RT == 1 && CK % 16 == 0 && (OT == VC_OPERAND_F32 || OT == VC_OPERAND_X6)), "dx shift: one tile per wave, 16-byte row chunks, fp32 rows (round 6: also with split products)");
  static_assert(!IL || (CK % 16 == 0 && !DXS), "interleaved source: 16-byte row chunks");
  static_assert(EPI == VC_EPI_NONE || (EPI == VC_EPI_BWD) == BWD, "STATS / AFFINE: forward kernel; BWD: backward-input kernel");
  constexpr int V = (CK >= 16) ? 4 : CK / 4;
  constexpr int NCH = CK / (4 * V);
  constexpr int NT = (CN + 15) / 16;
  // NW waves per block (4, or 8: the W_k image and its barrier are shared by twice the rows -- half the LDS per wave, which
  // is what limits the occupancy of the 64-channel instantiations: 23.6 KB per 4-wave block = 6 waves per SIMD)
  constexpr int NTHR = 64 * NW;
  constexpr int TM = 16 * NW * RT;                       // output rows per block: NW waves x RT tiles of 16
  constexpr int NFRAG = NCH * NT * 64;                   // fragment vectors (V floats each) of one W_k image
  constexpr int BF = NFRAG * V;
  constexpr int BBYTES = BF * (OT == VC_OPERAND_F32 ? 4 : (OT == VC_OPERAND_X6 ? 6 : 2));  // one W_k image in LDS (fp32, 16-bit operands, or three bf16 planes)
  constexpr int XPL = NFRAG * 8;                               // X6: bytes of one bf16 plane
  constexpr int BLD = (NFRAG + NTHR - 1) / NTHR;               // fragment vectors staged per thread
  static_assert(OT == VC_OPERAND_F32 || V == 4, "16-bit MFMA operands need >= 16 source channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_b = smem;                             // [2][BBYTES]
  int* s_idx = reinterpret_cast<int*>(smem + 2 * BBYTES);              // [kv][TM]
  int* s_row = s_idx + kv * TM;                                        // [TM] output row of each tile slot (-1: none)
  unsigned* s_mask = reinterpret_cast<unsigned*>(s_row + TM);          // [1] offsets active in this block

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  // XCD-aware block -> row-range mapping: the dispatcher places block b on XCD b % 8 (each XCD has its own 4 MiB L2).
  // Give every XCD one CONTIGUOUS eighth of the rows so that the rows gathered by neighbouring blocks (same (y,z)
  // neighbourhood) are served by the same L2 instead of being fetched eight times.  Bijective for any grid size.
  int64_t lbid;
  {
    const unsigned nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7u, qd = nb >> 3, rm = nb & 7u;
    lbid = (int64_t)(xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    if (g_xcd_swizzle_off) lbid = bid;
  }
  const int64_t brow0 = lbid * TM;

  const int64_t n_src_buf = IL ? ((n_src + 15) & ~(int64_t)15) : n_src;   // interleaved sources are padded to whole 16-row groups
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(n_src_buf * CK * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ctr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src_centre ? src_centre : src), 0, (int)(n_src_buf * CK * 4), 0x00020000);

  if (tid == 0) s_mask[0] = 0u;
  __syncthreads();
  {  // ---- phase 0: stage the pair-table slice; thread handles row r of offsets k0, k0 + 256/TM, ...
    // `order` (optional) is a permutation of the rows that puts rows with equal active-offset sets next to each other:
    // tile slot s computes output row order[s].  Results do not depend on it (every row is computed independently with
    // the same offset order); it only makes the per-tile union of active offsets -- the work actually issued -- smaller.
    const int r = tid % TM;
    const bool inb = brow0 + r < n_out;
    const int64_t row = inb ? (order ? (int64_t)order[brow0 + r] : brow0 + r) : -1;
    if (tid < TM) s_row[r] = (int)row;
    const bool centre_only = (rep != nullptr) && inb && (rep[row] != (int32_t)row);
    // all of a thread's table entries in flight together (up to 8 per trip): the plain loop waited for every load before its
    // LDS store and ballot -- 7 dependent global round trips at the head of EVERY block (KV = 27), most of the run time of the
    // single-round launches of stage 1
    constexpr int KSTEP = NTHR / TM;
    for (int kb = tid / TM; kb < kv; kb += 8 * KSTEP) {
      int vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + u * KSTEP;
        vv[u] = (inb && k < kv) ? tbl[(int64_t)k * n_out + row] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + u * KSTEP;   // wave-uniform (TM is a multiple of 64)
        if (k < kv) {
          int v = vv[u];
          if (centre_only && k != centre) v = -1;
          s_idx[k * TM + r] = v;
          if (__ballot(v >= 0) != 0ULL && lane == 0) atomicOr(&s_mask[0], 1u << k);
        }
      }
    }
  }
  __syncthreads();
  unsigned bmask = (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[0]);

  f32x4 acc[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  float breg[BLD][V];
  float a_cur[RT][NCH][V], a_nxt[RT][NCH][V];
  int act_cur[RT], act_nxt[RT];

#define VC_LOAD_B(K)                                                                               \
  do {                                                                                             \
    const int kw_ = mirror ? (kv - 1 - (K)) : (K);                                                 \
    _Pragma("unroll") for (int u = 0; u < BLD; ++u) {                                              \
      const int f = tid + u * NTHR;                                                                \
      if (NFRAG % NTHR == 0 || f < NFRAG) {                                                         \
        const int fl = f & 63, nt_ = (f >> 6) % NT, ch_ = (f >> 6) / NT;                           \
        const int n_ = nt_ * 16 + (fl & 15), kk0 = ch_ * 4 * V + (fl >> 4) * V;                    \
        if constexpr (PK) {  /* fragment-ordered image (vc_conv_pack_weights): 16 contiguous bytes per lane */ \
          VecLoad<V>::ld(w + ((int64_t)kw_ * NFRAG + f) * V, breg[u]);                             \
        } else                                                                                     \
        if (CN % 16 == 0 || n_ < CN) {                                                             \
          if (!BWD) {                                                                              \
            VecLoad<V>::ld(w + ((int64_t)n_ * kv + kw_) * CK + kk0, breg[u]);                      \
          } else {                                                                                 \
            _Pragma("unroll") for (int j = 0; j < V; ++j)                                          \
                breg[u][j] = w[((int64_t)(kk0 + j) * kv + kw_) * CN + n_];                         \
          }                                                                                        \
        } else {                                                                                   \
          _Pragma("unroll") for (int j = 0; j < V; ++j) breg[u][j] = 0.f;                          \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  } while (0)

#define VC_STORE_B(BUF)                                                                            \
  do {                                                                                             \
    _Pragma("unroll") for (int u = 0; u < BLD; ++u) {                                              \
      const int f = tid + u * NTHR;                                                                \
      if (NFRAG % NTHR == 0 || f < NFRAG) {                                                         \
        if constexpr (OT == VC_OPERAND_F32) {                                                      \
          float* d_ = reinterpret_cast<float*>(s_b + (BUF) * BBYTES) + f * V;                      \
          _Pragma("unroll") for (int j = 0; j < V; ++j) d_[j] = breg[u][j];                        \
        } else if constexpr (OT == VC_OPERAND_X6) {                                                \
          /* three bf16 planes; with an even chunk count the fragments of chunks 2c and 2c+1 sit side by side (one 16-byte read) */ \
          u32x2 h_, m_, l_;                                                                        \
          split3(breg[u], h_, m_, l_);                                                             \
          const int fl_ = f & 63, nt_ = (f >> 6) % NT, ch_ = (f >> 6) / NT;                        \
          const int off_ = (NCH % 2 == 0) ? ((((ch_ >> 1) * NT + nt_) * 64 + fl_) * 16 + (ch_ & 1) * 8) : f * 8; \
          unsigned char* d_ = s_b + (BUF) * BBYTES + off_;                                         \
          *reinterpret_cast<u32x2*>(d_) = h_;                                                      \
          *reinterpret_cast<u32x2*>(d_ + XPL) = m_;                                                \
          *reinterpret_cast<u32x2*>(d_ + 2 * XPL) = l_;                                            \
        } else {                                                                                   \
          *reinterpret_cast<u32x2*>(s_b + (BUF) * BBYTES + f * 8) = Pack4<OT == VC_OPERAND_F32 ? VC_OPERAND_F16 : OT>::cvt(breg[u]); \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  } while (0)

#define VC_GATHER_A(K, A, ACT)                                                                     \
  do {                                                                                             \
    const __amdgpu_buffer_rsrc_t rs_ = ((K) == centre) ? rs_ctr : rs_src;                          \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                               \
      const int id = s_idx[(K) * TM + wave * (RT * 16) + t * 16 + i];                              \
      ACT[t] = __builtin_amdgcn_readfirstlane((int)(__ballot(id >= 0) != 0ULL));                   \
      /* index -1: (unsigned)(-1 >> 4) * CK * 64 + ... stays just below 2^32 -> out of range -> zeros, as in the row-major form */ \
      const unsigned base = IL ? ((unsigned)(id >> 4) * (unsigned)(CK * 64) + (unsigned)((id & 15) * 16) + (unsigned)(q * 256)) \
                               : ((unsigned)id * (unsigned)(CK * 4) + (unsigned)(q * V * 4));      \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
          BufLoad<V>::ld(rs_, base + (unsigned)(IL ? ch * 1024 : ch * 4 * V * 4), A[t][ch]);       \
    }                                                                                              \
  } while (0)

  // Software pipeline, two offsets per trip with ping-pong register sets (A0/A1): the loads of offset n+1 (W slice ->
  // breg, gathers -> the other A set) are issued before the MFMAs of offset n and are all UNCONDITIONAL (past the last
  // active offset they re-load the current one), so the loop body is straight-line code and hipcc keeps counted vmcnt
  // waits: the MFMAs never wait for the loads issued in their own half-trip.
#define VC_MFMA(A, ACT, BUF)                                                                       \
  do {                                                                                             \
    if constexpr (OT == VC_OPERAND_F32) {                                                          \
      const float* __restrict__ B_ = reinterpret_cast<const float*>(s_b + (BUF) * BBYTES);         \
      float b[NCH][NT][V];                                                                         \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
          _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                        \
              VecLoad<V>::ld(B_ + ((ch * NT + nt) * 64 + lane) * V, b[ch][nt]);                    \
      if (VC_V2_SCHED == 1 && NCH * NT <= 8) __builtin_amdgcn_sched_barrier(0);                    \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                             \
        if (ACT[t]) {                                                                              \
          if constexpr (VC_V2_SCHED != 0) {                                                        \
            _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                     \
                _Pragma("unroll") for (int j = 0; j < V; ++j)                                      \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                              \
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[t][ch][j], b[ch][nt][j], acc[t][nt], 0, 0, 0); \
          } else {                                                                                 \
            _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                     \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
                    _Pragma("unroll") for (int j = 0; j < V; ++j)                                  \
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[t][ch][j], b[ch][nt][j], acc[t][nt], 0, 0, 0); \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
    } else if constexpr (OT == VC_OPERAND_X6) {                                                    \
      const unsigned char* __restrict__ B_ = s_b + (BUF) * BBYTES;                                 \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                             \
        if (ACT[t]) {                                                                              \
          if constexpr (NCH % 2 == 0) {                                                            \
            _Pragma("unroll") for (int cp = 0; cp < NCH / 2; ++cp) {                               \
              u32x2 h0_, m0_, l0_, h1_, m1_, l1_;                                                  \
              split3(A[t][2 * cp], h0_, m0_, l0_);                                                 \
              split3(A[t][2 * cp + 1], h1_, m1_, l1_);                                             \
              const u32x4 ah = {h0_.x, h0_.y, h1_.x, h1_.y}, am = {m0_.x, m0_.y, m1_.x, m1_.y},    \
                          al = {l0_.x, l0_.y, l1_.x, l1_.y};                                       \
              _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                  \
                const unsigned char* p_ = B_ + ((cp * NT + nt) * 64 + lane) * 16;                  \
                const u32x4 bh = *reinterpret_cast<const u32x4*>(p_), bm = *reinterpret_cast<const u32x4*>(p_ + XPL), \
                            bl = *reinterpret_cast<const u32x4*>(p_ + 2 * XPL);                    \
                f32x4 c_ = acc[t][nt];                                                             \
                c_ = mfma32bf(al, bh, c_); c_ = mfma32bf(ah, bl, c_); c_ = mfma32bf(am, bm, c_);   \
                c_ = mfma32bf(am, bh, c_); c_ = mfma32bf(ah, bm, c_); c_ = mfma32bf(ah, bh, c_);   \
                acc[t][nt] = c_;                                                                   \
              }                                                                                    \
            }                                                                                      \
          } else {                                                                                 \
            _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch) {                                   \
              u32x2 ah, am, al;                                                                    \
              split3(A[t][ch], ah, am, al);                                                        \
              _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                  \
                const unsigned char* p_ = B_ + ((ch * NT + nt) * 64 + lane) * 8;                   \
                const u32x2 bh = *reinterpret_cast<const u32x2*>(p_), bm = *reinterpret_cast<const u32x2*>(p_ + XPL), \
                            bl = *reinterpret_cast<const u32x2*>(p_ + 2 * XPL);                    \
                f32x4 c_ = acc[t][nt];                                                             \
                c_ = mfma16<VC_OPERAND_BF16>(al, bh, c_); c_ = mfma16<VC_OPERAND_BF16>(ah, bl, c_); \
                c_ = mfma16<VC_OPERAND_BF16>(am, bm, c_); c_ = mfma16<VC_OPERAND_BF16>(am, bh, c_); \
                c_ = mfma16<VC_OPERAND_BF16>(ah, bm, c_); c_ = mfma16<VC_OPERAND_BF16>(ah, bh, c_); \
                acc[t][nt] = c_;                                                                   \
              }                                                                                    \
            }                                                                                      \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
    } else {                                                                                       \
      constexpr int OT_ = (OT == VC_OPERAND_F32) ? VC_OPERAND_F16 : OT;                            \
      const unsigned char* __restrict__ B_ = s_b + (BUF) * BBYTES;                                 \
      u32x2 b[NCH][NT];                                                                            \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
          _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                        \
              b[ch][nt] = *reinterpret_cast<const u32x2*>(B_ + ((ch * NT + nt) * 64 + lane) * 8);  \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                             \
        if (ACT[t]) {                                                                              \
          _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch) {                                     \
            const u32x2 ah = Pack4<OT_>::cvt(A[t][ch]);                                            \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                      \
                acc[t][nt] = mfma16<OT_>(ah, b[ch][nt], acc[t][nt]);                               \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  if constexpr (DXS) {
    // C = centre fragment of the loaded group; idc = its table entries; F0 / F1 = operand sets (a_cur / a_nxt): first the target of
    // the fix-up gather, then (VC_DXS_FORM, right before the MFMAs) the finished operand
    float c_reg[NCH][V];
    int idc = -1, gl = -1;
    int mt_cur = 0, mt_nxt = 0, side_cur = 1, side_nxt = 1;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int j = 0; j < V; ++j) { c_reg[ch][j] = 0.f; a_cur[0][ch][j] = 0.f; a_nxt[0][ch][j] = 0.f; }

#define VC_DXS_ISSUE(K, F, MT, SIDE, ACT)                                                          \
  do {                                                                                             \
    const int g_ = (K) / 3;                                                                        \
    SIDE = (K) - 3 * g_;                                                                           \
    if (g_ != gl) {   /* new group: its centre rows, all of them */                                \
      gl = g_;                                                                                     \
      idc = s_idx[(3 * g_ + 1) * TM + wave * 16 + i];                                              \
      const unsigned cb_ = (unsigned)idc * (unsigned)(CK * 4) + (unsigned)(q * V * 4);             \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
          BufLoad<V>::ld(rs_src, cb_ + (unsigned)(ch * 4 * V * 4), c_reg[ch]);                     \
    }                                                                                              \
    const int idk_ = s_idx[(K) * TM + wave * 16 + i];                                              \
    ACT[0] = __builtin_amdgcn_readfirstlane((int)(__ballot(idk_ >= 0) != 0ULL));                   \
    MT = 1;                                                                                        \
    if (SIDE != 1) {                                                                               \
      const int sh_ = (SIDE == 0) ? __builtin_amdgcn_update_dpp(-2, idc, 0x111, 0xf, 0xf, false)   \
                                  : __builtin_amdgcn_update_dpp(-2, idc, 0x101, 0xf, 0xf, false);  \
      MT = (idk_ == sh_) ? 1 : 0;                                                                  \
      if (__ballot(MT == 0) != 0ULL) {   /* some lane's row is not its neighbour's centre row: gather those lanes only */ \
        const int idf_ = MT ? -1 : idk_;                                                           \
        const unsigned fb_ = (unsigned)idf_ * (unsigned)(CK * 4) + (unsigned)(q * V * 4);          \
        _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                         \
            BufLoad<V>::ld(rs_src, fb_ + (unsigned)(ch * 4 * V * 4), F[0][ch]);                    \
      }                                                                                            \
    }                                                                                              \
  } while (0)

#define VC_DXS_FORM(F, MT, SIDE)                                                                   \
  do {                                                                                             \
    if (SIDE == 1) {                                                                               \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
          _Pragma("unroll") for (int j = 0; j < V; ++j) F[0][ch][j] = c_reg[ch][j];                \
    } else if (SIDE == 0) {                                                                        \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
          _Pragma("unroll") for (int j = 0; j < V; ++j) {                                          \
            const int s_ = __builtin_amdgcn_update_dpp(0, __float_as_int(c_reg[ch][j]), 0x111, 0xf, 0xf, false); \
            F[0][ch][j] = MT ? __int_as_float(s_) : F[0][ch][j];                                   \
          }                                                                                        \
    } else {                                                                                       \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch)                                           \
          _Pragma("unroll") for (int j = 0; j < V; ++j) {                                          \
            const int s_ = __builtin_amdgcn_update_dpp(0, __float_as_int(c_reg[ch][j]), 0x101, 0xf, 0xf, false); \
            F[0][ch][j] = MT ? __int_as_float(s_) : F[0][ch][j];                                   \
          }                                                                                        \
    }                                                                                              \
  } while (0)

    if (bmask != 0u) {
      int kcur = __ffs((int)bmask) - 1;
      bmask &= bmask - 1;
      VC_LOAD_B(kcur);
      VC_DXS_ISSUE(kcur, a_cur, mt_cur, side_cur, act_cur);
      for (;;) {
        VC_STORE_B(0);
        __syncthreads();
        VC_DXS_FORM(a_cur, mt_cur, side_cur);
        const bool more0 = bmask != 0u;
        const int k1 = more0 ? (__ffs((int)bmask) - 1) : kcur;
        bmask &= bmask - 1;
        if (more0) {
          VC_LOAD_B(k1);
          VC_DXS_ISSUE(k1, a_nxt, mt_nxt, side_nxt, act_nxt);
        }
        VC_MFMA(a_cur, act_cur, 0);
        if (!more0) break;
        VC_STORE_B(1);
        __syncthreads();
        VC_DXS_FORM(a_nxt, mt_nxt, side_nxt);
        const bool more1 = bmask != 0u;
        kcur = more1 ? (__ffs((int)bmask) - 1) : k1;
        bmask &= bmask - 1;
        if (more1) {
          VC_LOAD_B(kcur);
          VC_DXS_ISSUE(kcur, a_cur, mt_cur, side_cur, act_cur);
        }
        VC_MFMA(a_nxt, act_nxt, 1);
        if (!more1) break;
      }
    }
#undef VC_DXS_FORM
#undef VC_DXS_ISSUE
  } else
  if (bmask != 0u) {
    int kcur = __ffs((int)bmask) - 1;
    bmask &= bmask - 1;
    VC_LOAD_B(kcur);
    VC_GATHER_A(kcur, a_cur, act_cur);
    for (;;) {
      // ---- half-trip 0: consume (breg, A0 = a_cur) as offset kcur, prefetch the next into (breg, A1 = a_nxt)
      VC_STORE_B(0);
      __syncthreads();
      const bool more0 = bmask != 0u;
      const int k1 = more0 ? (__ffs((int)bmask) - 1) : kcur;
      bmask &= bmask - 1;
      VC_LOAD_B(k1);
      VC_GATHER_A(k1, a_nxt, act_nxt);
      VC_MFMA(a_cur, act_cur, 0);
      if (!more0) break;
      // ---- half-trip 1: consume (breg, A1) as offset k1, prefetch the next into (breg, A0)
      VC_STORE_B(1);
      __syncthreads();
      const bool more1 = bmask != 0u;
      kcur = more1 ? (__ffs((int)bmask) - 1) : k1;
      bmask &= bmask - 1;
      VC_LOAD_B(kcur);
      VC_GATHER_A(kcur, a_cur, act_cur);
      VC_MFMA(a_nxt, act_nxt, 1);
      if (!more1) break;
    }
  }
#undef VC_MFMA
#undef VC_LOAD_B
#undef VC_STORE_B
#undef VC_GATHER_A

  float fin_a[NT], fin_b[NT];   // this wave's partial sums (STATS / BWD) for the in-kernel finish
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { fin_a[nt] = 0.f; fin_b[nt] = 0.f; }
  if constexpr (EPI == VC_EPI_STATS) {
    // per-WAVE partial sums (the wave's 16 rows): rows beyond n_out gathered nothing, their accumulators are exact zeros; fixed
    // order (4 accumulator rows, then the q lanes), no LDS and no barrier -- the block-level reduce this replaces cost two
    // barriers per block and made the fused statistics slower than the pass over y they save (round 1)
    float* prow = epi.partial + ((lbid * NW + wave) * 2) * CN;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float sm = ((acc[0][nt][0] + acc[0][nt][1]) + acc[0][nt][2]) + acc[0][nt][3];
      float sq = ((acc[0][nt][0] * acc[0][nt][0] + acc[0][nt][1] * acc[0][nt][1]) + acc[0][nt][2] * acc[0][nt][2]) +
                 acc[0][nt][3] * acc[0][nt][3];
      sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
      sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
      const int n = nt * 16 + i;
      fin_a[nt] = sm; fin_b[nt] = sq;   // in-kernel finish: kept until the block's tail
      if (q == 0 && n < CN && epi.fin.cnt == nullptr) { prow[n] = sm; prow[CN + n] = sq; }
    }
  }
  float sc[NT], sh[NT];
  if constexpr (EPI == VC_EPI_AFFINE) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + i;
      sc[nt] = 1.f; sh[nt] = 0.f;
      if (n < CN) {
        const float istd = 1.0f / sqrtf(epi.var[n] + epi.eps);
        sc[nt] = (epi.gamma ? epi.gamma[n] : 1.f) * istd;
        sh[nt] = (epi.beta ? epi.beta[n] : 0.f) - epi.mean[n] * sc[nt];
      }
    }
  }
  // VC_EPI_BWD: per-channel constants of the producing unit's BatchNorm (the same expressions as bn_bwd_dx_*: identical xhat,
  // identical ReLU mask) and the running sums of this lane's 4 rows
  float b_mu[NT], b_istd[NT], b_g[NT], b_bt[NT], b_sa[NT], b_sb[NT];
  const bool bwd_stats = (EPI == VC_EPI_BWD) && epi.y_raw != nullptr;
  if constexpr (EPI == VC_EPI_BWD) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + i;
      b_mu[nt] = 0.f; b_istd[nt] = 0.f; b_g[nt] = 1.f; b_bt[nt] = 0.f; b_sa[nt] = 0.f; b_sb[nt] = 0.f;
      if (bwd_stats && n < CN) {
        b_mu[nt] = epi.mean[n];
        b_istd[nt] = 1.0f / sqrtf(epi.var[n] + epi.eps);
        b_g[nt] = epi.gamma ? epi.gamma[n] : 1.f;
        b_bt[nt] = epi.beta ? epi.beta[n] : 0.f;
      }
    }
  }
  // ---- phase A: the final values, in place in the accumulators (AFFINE: scale / shift / ReLU; BWD: + addend, and the sums)
  int orow[RT][4];   // output rows of this wave's tile slots, read before the finish part reuses the LDS
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) orow[t][reg] = s_row[wave * (RT * 16) + t * 16 + q * 4 + reg];
  if constexpr (EPI == VC_EPI_AFFINE || EPI == VC_EPI_BWD) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + i;
        if (n >= CN) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          float v = acc[t][nt][reg];
          if constexpr (EPI == VC_EPI_AFFINE) {
            v = v * sc[nt] + sh[nt];
            if (epi.relu) v = fmaxf(v, 0.f);
          }
          if constexpr (EPI == VC_EPI_BWD) {
            if (orow[t][reg] >= 0) {
              if (epi.addend != nullptr) v += epi.addend[(int64_t)orow[t][reg] * epi.add_stride + epi.add_col0 + n];
              if (bwd_stats) {
                const float xh = (epi.y_raw[(int64_t)orow[t][reg] * CN + n] - b_mu[nt]) * b_istd[nt];
                float d = v;
                if (epi.relu && !(xh * b_g[nt] + b_bt[nt] > 0.f)) d = 0.f;
                b_sa[nt] += d;
                b_sb[nt] += d * xh;
              }
            }
          }
          acc[t][nt][reg] = v;
        }
      }
  }
  if constexpr (EPI == VC_EPI_BWD) {
    if (bwd_stats) {  // per-WAVE partial row [2][CN]: (sum dy_masked, sum dy_masked * xhat), fixed order, no LDS, no barrier
      float* prow = epi.partial + ((lbid * NW + wave) * 2) * CN;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float sa = b_sa[nt], sb = b_sb[nt];
        sa += __shfl_xor(sa, 16, 64); sb += __shfl_xor(sb, 16, 64);
        sa += __shfl_xor(sa, 32, 64); sb += __shfl_xor(sb, 32, 64);
        const int n = nt * 16 + i;
        fin_a[nt] = sa; fin_b[nt] = sb;
        if (q == 0 && n < CN && epi.fin.cnt == nullptr) { prow[n] = sa; prow[CN + n] = sb; }
      }
    }
  }
  // ---- in-kernel BatchNorm finish, ticket part: BEFORE the output stores (it waits for its own write-through store only)
  int fin_last = 0;
  if constexpr ((EPI == VC_EPI_STATS || EPI == VC_EPI_BWD) && RT == 1 && (CN & (CN - 1)) == 0 && CN >= 8) {
    if (epi.fin.cnt != nullptr) fin_last = conv_finish_ticket<CN, NTHR, NT>(epi.fin, (int)lbid, wave, fin_a, fin_b, smem);   // kernel-uniform
  }
  // ---- phase B: the stores
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + i;
      if (n >= CN) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        if (orow[t][reg] >= 0) out[(int64_t)orow[t][reg] * CN + n] = acc[t][nt][reg];
    }
  if constexpr ((EPI == VC_EPI_STATS || EPI == VC_EPI_BWD) && RT == 1 && (CN & (CN - 1)) == 0 && CN >= 8) {
    if (fin_last) conv_finish_reduce<CN, NTHR>(epi.fin, (int)lbid, wave, smem);   // block-uniform
  }
}

// The measured-and-rejected gather-GEMM variants (pair-compacted pc, wave-autonomous v4, loader/MFMA roles v5, LDS row windows v3)
// live in csrc/experiments/ and are compiled only with -DVC_EXPERIMENTS.
#ifdef VC_EXPERIMENTS
#include "experiments/conv_pc.inc"
#include "experiments/conv_v4.inc"
#include "experiments/conv_v5.inc"
#include "experiments/conv_v3.inc"
#endif

// --------------------------------------------------------------------------------------------- K8 weight gradient
// grid (nsplit, kv); each block owns offset k = blockIdx.y and a contiguous range of output rows.  Each wave scans its
// rows 64 at a time, ballot/prefix-compacts the active (in, out) pairs into an LDS queue, and consumes the queue four
// pairs per MFMA step:  dW_k[ci][co] += x[in_p][ci] * dy[out_p][co]   (M = ci, N = co, K = pair).
// Operand vectorisation: lane (i, q) loads VA = CI/16 contiguous input channels [VA*i, VA*i+VA) of pair q's input row and
// VB = CO/16 contiguous output channels of its dy row with ONE vector load each and issues VA x VB MFMAs per 4 pairs;
// tile (ja, jb) therefore holds dW[ci = VA*m + ja][co = VB*n + jb] (a fixed permutation of the M / N dimensions).
template <int CI, int CO, int OT>
__device__ __forceinline__ void bw_group16(const float* __restrict__ x, const float* __restrict__ dy, const int* qi,
                                           const int* qo, int npairs, int i, int q, bool a_ok, bool b_ok,
                                           f32x4 (&acc)[(CI >= 16) ? CI / 16 : 1][(CO >= 16) ? CO / 16 : 1]) {
  constexpr int VA = (CI >= 16) ? CI / 16 : 1, VB = (CO >= 16) ? CO / 16 : 1;
  constexpr int OT_ = (OT == VC_OPERAND_F32) ? VC_OPERAND_F16 : OT;
  float a[4][VA], b[4][VB];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int slot = q * 4 + r;
    const bool ok = slot < npairs;
    const int pin = ok ? qi[slot] : 0, pout = ok ? qo[slot] : 0;
    if (ok && a_ok) VecLoad<VA>::ld(x + (int64_t)pin * CI + VA * i, a[r]);
    else {
#pragma unroll
      for (int j = 0; j < VA; ++j) a[r][j] = 0.f;
    }
    if (ok && b_ok) VecLoad<VB>::ld(dy + (int64_t)pout * CO + VB * i, b[r]);
    else {
#pragma unroll
      for (int j = 0; j < VB; ++j) b[r][j] = 0.f;
    }
  }
  u32x2 ah[VA], bh[VB];
#pragma unroll
  for (int j = 0; j < VA; ++j) {
    const float t[4] = {a[0][j], a[1][j], a[2][j], a[3][j]};
    ah[j] = Pack4<OT_>::cvt(t);
  }
#pragma unroll
  for (int j = 0; j < VB; ++j) {
    const float t[4] = {b[0][j], b[1][j], b[2][j], b[3][j]};
    bh[j] = Pack4<OT_>::cvt(t);
  }
#pragma unroll
  for (int ja = 0; ja < VA; ++ja)
#pragma unroll
    for (int jb = 0; jb < VB; ++jb) acc[ja][jb] = mfma16<OT_>(ah[ja], bh[jb], acc[ja][jb]);
}

// ---- operand gathers of the 32-pair group (round 6: 16-byte loads for 32- and 16-channel rows) ----------------------------------------
// The MFMA wants, in lane (i, q), ONE channel per operand register set and the 8 pairs 8q..8q+7 of the group (K-slots e = 0..7).  The plain
// form lets the lane load its VC = C / 16 channels from each of the 8 rows: 8-byte loads at C = 32, 4-byte loads at C = 16 -- and the unit
// that is busy in this kernel is the texture addresser, which spends as long on a 512-byte instruction as on a 1-KB one
// (profiles/r05_mfma_busy_dw.md: TA 58 % busy, matrix pipes 18 %).  The wide form gives the S = 64 / C lanes of a 16-lane row that share a
// 16-byte channel block DIFFERENT rows of the octet (8 / S each), so every load is 16 bytes, and hands each lane the channel it owns from
// the others' registers with DPP row rotations (v_mov_b32_dpp row_ror:8 / 4 / 12 with bank masks: register-to-register, no LDS):
//   C = 32: lanes i, i ^ 8 share block i & 7; the lower loads slots 0..3, the upper 4..7; the lower ends up with channels 4 (i & 7) + {0, 1},
//           the upper with + {2, 3}: 4 loads + 16 DPP moves instead of 8 loads.
//   C = 16: lanes with equal i & 3 share block i & 3; lane loads slots 2 (i >> 2) + {0, 1}; ends up with channel 4 (i & 3) + (i >> 2):
//           2 loads + 32 DPP moves instead of 8 loads.
// The K-slot -> pair assignment (slot 8q + e = pair 8q + e of the group) is the plain form's, so a wide operand pairs with a plain one
// (C = 64, already 16-byte loads); only the lane -> channel map differs (BwChan), and that is the epilogue's business.
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_banks(float old, float src) {   // lanes of the banks in BANK take `src` through CTRL, the others keep `old`
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, BANK, false));
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int C, bool WIDE>
struct BwChan {   // channel of M / N index m (0..15), register set j (0..C/16-1)
  static __device__ __forceinline__ int of(int m, int j) {
    if constexpr (WIDE && C == 32) return 4 * (m & 7) + 2 * (m >> 3) + j;
    else if constexpr (WIDE && C == 16) return 4 * (m & 3) + (m >> 2);
    else return ((C >= 16) ? C / 16 : 1) * m + j;
  }
};

// v[j][e]: channel BwChan<C, WIDE>::of(i, j) of the row of K-slot 8q + e (`rows` = its index in `base`'s rows), zero from slot `npairs` on
template <int C, bool WIDE>
__device__ __forceinline__ void bw_gather8(const float* __restrict__ base, const int* rows, int npairs, int i, int q, bool lane_ok,
                                           float (&v)[(C >= 16) ? C / 16 : 1][8]) {
  constexpr int VC = (C >= 16) ? C / 16 : 1;
  if constexpr (WIDE && C == 32) {
    float r[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int slot = q * 8 + (i >> 3) * 4 + t;
      if (slot < npairs) VecLoad<4>::ld(base + (int64_t)rows[slot] * 32 + 4 * (i & 7), r[t]);
      else r[t][0] = r[t][1] = r[t][2] = r[t][3] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v[j][t] = dpp_banks<0x128, 0xC>(r[t][j], r[t][2 + j]);       // upper lanes: the lower partner's channel 2 + j of slot t
        v[j][4 + t] = dpp_banks<0x128, 0x3>(r[t][2 + j], r[t][j]);   // lower lanes: the upper partner's channel j of slot 4 + t
      }
  } else if constexpr (WIDE && C == 16) {
    float r[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int slot = q * 8 + (i >> 2) * 2 + t;
      if (slot < npairs) VecLoad<4>::ld(base + (int64_t)rows[slot] * 16 + 4 * (i & 3), r[t]);
      else r[t][0] = r[t][1] = r[t][2] = r[t][3] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[0][e] = 0.f;
    // receiver bank h (= i >> 2) takes channel h of slot 2 h' + t from the lane 4 rho below it in its row (h' = h - rho mod 4)
    static_for<4>([&](auto rho_) {
      constexpr int RHO = decltype(rho_)::value;
      static_for<4>([&](auto h_) {
        constexpr int H = decltype(h_)::value;
        constexpr int E0 = 2 * ((H - RHO) & 3);
        constexpr int CTRL = (RHO == 0) ? 0xE4 : 0x120 + 4 * RHO;    // quad_perm:[0,1,2,3] (own registers) | row_ror:4 rho
        v[0][E0] = dpp_banks<CTRL, 1 << H>(v[0][E0], r[0][H]);
        v[0][E0 + 1] = dpp_banks<CTRL, 1 << H>(v[0][E0 + 1], r[1][H]);
      });
    });
  } else {
    float a[8][VC];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int slot = q * 8 + e;
      if (slot < npairs && lane_ok) VecLoad<VC>::ld(base + (int64_t)rows[slot] * C + VC * i, a[e]);
      else {
#pragma unroll
        for (int j = 0; j < VC; ++j) a[e][j] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < VC; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = a[e][j];
  }
}

// OT = VC_OPERAND_X6 (fp32 products as six bf16 terms, see split3): 32 pairs per step on v_mfma_f32_16x16x32_bf16; lane (i, q) loads the
// rows of the 8 pairs 8q..8q+7 of the group; per channel those 8 values are cut into three bf16 octets.  The gradient rows' pieces are
// formed once per group, the input rows' pieces per channel column (register pressure: 8 (VA + VB) operand floats + 12 VB piece
// registers are live beside the VA VB accumulator quads).
template <int CI, int CO, bool WA, bool WB>
__device__ __forceinline__ void bw_group32_x6(const float* __restrict__ x, const float* __restrict__ dy, const int* qi, const int* qo,
                                              int npairs, int i, int q, bool a_ok, bool b_ok,
                                              f32x4 (&acc)[(CI >= 16) ? CI / 16 : 1][(CO >= 16) ? CO / 16 : 1]) {
  constexpr int VA = (CI >= 16) ? CI / 16 : 1, VB = (CO >= 16) ? CO / 16 : 1;
  float a[VA][8], b[VB][8];
  bw_gather8<CI, WA>(x, qi, npairs, i, q, a_ok, a);
  bw_gather8<CO, WB>(dy, qo, npairs, i, q, b_ok, b);
  u32x4 bh[VB], bm[VB], bl[VB];
#pragma unroll
  for (int j = 0; j < VB; ++j) {
    u32x2 h0, m0, l0, h1, m1, l1;
    split3(&b[j][0], h0, m0, l0);
    split3(&b[j][4], h1, m1, l1);
    bh[j] = u32x4{h0.x, h0.y, h1.x, h1.y}; bm[j] = u32x4{m0.x, m0.y, m1.x, m1.y}; bl[j] = u32x4{l0.x, l0.y, l1.x, l1.y};
  }
#pragma unroll
  for (int ja = 0; ja < VA; ++ja) {
    u32x2 h0, m0, l0, h1, m1, l1;
    split3(&a[ja][0], h0, m0, l0);
    split3(&a[ja][4], h1, m1, l1);
    const u32x4 ah = {h0.x, h0.y, h1.x, h1.y}, am = {m0.x, m0.y, m1.x, m1.y}, al = {l0.x, l0.y, l1.x, l1.y};
#pragma unroll
    for (int jb = 0; jb < VB; ++jb) {
      f32x4 c = acc[ja][jb];
      c = mfma32bf(al, bh[jb], c); c = mfma32bf(ah, bl[jb], c); c = mfma32bf(am, bm[jb], c);
      c = mfma32bf(am, bh[jb], c); c = mfma32bf(ah, bm[jb], c); c = mfma32bf(ah, bh[jb], c);
      acc[ja][jb] = c;
    }
  }
}

// OT != VC_OPERAND_F32: 16 pairs per MFMA step (v_mfma_f32_16x16x16_{f16,bf16}); lane (i, q) loads the rows of the 4 pairs
// 4q..4q+3 of the group and packs, per channel, those 4 values (rounded to 16 bit) into one operand.
// A/B builds only (VIRCONV_HIPCC_EXTRA=-DVC_BW_WAVES=N): force N waves per SIMD on the weight-gradient kernel (round 6: the split form needs
// 96-200 VGPRs = 2-5 waves; does a latency-bound gather kernel gain from more resident waves at the price of spills?  profiles/r06_dw_occupancy.md)
#ifndef VC_BW_WAVES
#define VC_BW_WAVES 0
#endif
#if VC_BW_WAVES > 0
#define VC_BW_ATTR __attribute__((amdgpu_waves_per_eu(VC_BW_WAVES)))
#else
#define VC_BW_ATTR
#endif
template <int CI, int CO, int OT, bool WIDE = false>
__global__ void __launch_bounds__(256) VC_BW_ATTR bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const int32_t* __restrict__ tbl, int64_t n_out, int kv,
                                                         int64_t rows_per_block, int nsplit, int legacy_order,
                                                         float* __restrict__ partial, const int32_t* __restrict__ rep,
                                                         int centre, const float* __restrict__ dy_grp) {
  // Duplicate-pixel tables (rep != NULL; the 2-D image-space SubM convs): every row of a pixel group reads the SAME neighbour
  // row through a non-centre offset k, so sum_{o in group} x[tbl[k][o]]^T dy[o] = x[tbl[k][r]]^T dy_grp[r] with r the group's
  // representative and dy_grp the group-summed gradient the backward-input conv needs anyway (vc_group_sum_sorted): the
  // non-centre offsets walk the representatives only (19-53 % of the rows) against dy_grp, the centre offset every row against dy.
  constexpr int VA = (CI >= 16) ? CI / 16 : 1, VB = (CO >= 16) ? CO / 16 : 1;
  constexpr int MA = (CI >= 16) ? 16 : CI, NB = (CO >= 16) ? 16 : CO;  // lanes of the tile that carry data
#ifndef VC_BW_UMUL
#define VC_BW_UMUL 1   // A/B builds: groups per trip x 2 (LOG.md A.13: every layer within 1 %)
#endif
  constexpr int U = ((VA * VB >= 8) ? 2 : 4) * VC_BW_UMUL;             // groups of 4 pairs gathered per iteration
  constexpr int GP = (OT == VC_OPERAND_F32) ? 4 : (OT == VC_OPERAND_X6 ? 32 : 16);   // pairs per MFMA K-step
  // measured (profiles/r06_dw_wide.md): 32-channel operands gain 5-11 % where the other operand is not the 64-channel one; the 16-channel
  // form (2 loads + 32 DPP moves) loses 2-9 %: kept in the source (VC_BW_WIDE16) but not instantiated by default
#ifndef VC_BW_WIDE16
#define VC_BW_WIDE16 0
#endif
  constexpr bool WA = WIDE && OT == VC_OPERAND_X6 && (CI == 32 || (VC_BW_WIDE16 && CI == 16));
  constexpr bool WB = WIDE && OT == VC_OPERAND_X6 && ((CO == 32 && CI != 64) || (VC_BW_WIDE16 && CO == 16));
  __shared__ int q_in[4][136];
  __shared__ int q_out[4][136];
  __shared__ float red[CI * CO];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  // Block order: the kv offset-blocks of one row range are adjacent in launch order AND on the same XCD (hardware
  // places block b on XCD b % 8), so the range's x / dy rows are pulled from HBM once and the other kv-1 passes hit that
  // XCD's L2.  The legacy order (all ranges of offset 0, then offset 1, ...) streamed x and dy from HBM kv times.
  int k, split;
  if (legacy_order) {
    k = blockIdx.x / nsplit;
    split = blockIdx.x - k * nsplit;
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rl = j / kv;
    k = j - rl * kv;
    split = rl * 8 + xcd;
  }
  if (split >= nsplit) return;
  const bool reps_only = rep != nullptr && k != centre;
  if (reps_only) dy = dy_grp;
  const int64_t brow0 = (int64_t)split * rows_per_block;
  const int64_t bend = min(brow0 + rows_per_block, n_out);
  const int64_t rpw = rows_per_block / 4;  // rows_per_block is a multiple of 256
  const int64_t wstart = brow0 + wave * rpw;
  const int64_t wend = min(wstart + rpw, bend);
  const bool a_ok = i < MA, b_ok = i < NB;

  f32x4 acc[VA][VB];
#pragma unroll
  for (int ja = 0; ja < VA; ++ja)
#pragma unroll
    for (int jb = 0; jb < VB; ++jb) acc[ja][jb] = f32x4{0.f, 0.f, 0.f, 0.f};

  int* qi = q_in[wave];
  int* qo = q_out[wave];
  int qlen = 0;
  auto entry = [&](int64_t r) -> int {
    if (r >= wend) return -1;
    const int v = tbl[(int64_t)k * n_out + r];
    return (reps_only && rep[r] != (int32_t)r) ? -1 : v;
  };
  int v_next = entry(wstart + lane);
  for (int64_t base = wstart; base < wend; base += 64) {
    const int64_t r = base + lane;
    const int v = v_next;
    // prefetch the next 64 table entries: their latency hides under the gathers / MFMAs of this batch
    v_next = entry(r + 64);
    const bool valid = v >= 0;
    const unsigned long long m = __ballot(valid);
    if (m == 0ULL) continue;
    const int pos = __popcll(m & ((1ULL << lane) - 1ULL));
    if (valid) { qi[qlen + pos] = v; qo[qlen + pos] = (int)r; }
    qlen += __popcll(m);
    __builtin_amdgcn_wave_barrier();
    if constexpr (OT == VC_OPERAND_F32) {
      const int ng = qlen >> 2;
      int g = 0;
      // U groups of 4 pairs per trip: request, wait, 16 MFMAs.  A software-pipelined version (ping-pong register sets, the rows of
      // trip t+1 requested before the MFMAs of trip t, counted vmcnt waits) was MEASURED SLOWER: 84 instead of 70 VGPRs for
      // <64,32> = 5 instead of 7 waves per SIMD, dW total 1251 vs 1110 us per step (gpurun r2i) -- like the gather-GEMM, this
      // kernel hides its gather latency with resident waves, not with per-wave prefetch.
      for (; g + U <= ng; g += U) {
        float a[U][VA], b[U][VB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int pin = qi[(g + u) * 4 + q], pout = qo[(g + u) * 4 + q];
          if (a_ok) VecLoad<VA>::ld(x + (int64_t)pin * CI + VA * i, a[u]);
          else {
#pragma unroll
            for (int j = 0; j < VA; ++j) a[u][j] = 0.f;
          }
          if (b_ok) VecLoad<VB>::ld(dy + (int64_t)pout * CO + VB * i, b[u]);
          else {
#pragma unroll
            for (int j = 0; j < VB; ++j) b[u][j] = 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int ja = 0; ja < VA; ++ja)
#pragma unroll
            for (int jb = 0; jb < VB; ++jb)
              acc[ja][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][ja], b[u][jb], acc[ja][jb], 0, 0, 0);
      }
      for (; g < ng; ++g) {
        const int pin = qi[g * 4 + q], pout = qo[g * 4 + q];
        float a[VA], b[VB];
        if (a_ok) VecLoad<VA>::ld(x + (int64_t)pin * CI + VA * i, a);
        else {
#pragma unroll
          for (int j = 0; j < VA; ++j) a[j] = 0.f;
        }
        if (b_ok) VecLoad<VB>::ld(dy + (int64_t)pout * CO + VB * i, b);
        else {
#pragma unroll
          for (int j = 0; j < VB; ++j) b[j] = 0.f;
        }
#pragma unroll
        for (int ja = 0; ja < VA; ++ja)
#pragma unroll
          for (int jb = 0; jb < VB; ++jb)
            acc[ja][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ja], b[jb], acc[ja][jb], 0, 0, 0);
      }
    } else if constexpr (OT == VC_OPERAND_X6) {
      const int ng = qlen >> 5;
      for (int g = 0; g < ng; ++g) bw_group32_x6<CI, CO, WA, WB>(x, dy, qi + g * 32, qo + g * 32, 32, i, q, a_ok, b_ok, acc);
    } else {
      const int ng = qlen >> 4;
      for (int g = 0; g < ng; ++g) bw_group16<CI, CO, OT>(x, dy, qi + g * 16, qo + g * 16, 16, i, q, a_ok, b_ok, acc);
    }
    const int rem = qlen & (GP - 1);
    const int done = qlen - rem;
    int t1 = 0, t2 = 0;
    if (lane < rem) { t1 = qi[done + lane]; t2 = qo[done + lane]; }
    __builtin_amdgcn_wave_barrier();
    if (lane < rem) { qi[lane] = t1; qo[lane] = t2; }
    qlen = rem;
    __builtin_amdgcn_wave_barrier();
  }
  if (qlen > 0) {  // tail group, padded with zero operands
    if constexpr (OT == VC_OPERAND_F32) {
      const bool ok = q < qlen;
      const int pin = ok ? qi[q] : 0, pout = ok ? qo[q] : 0;
      float a[VA], b[VB];
      if (ok && a_ok) VecLoad<VA>::ld(x + (int64_t)pin * CI + VA * i, a);
      else {
#pragma unroll
        for (int j = 0; j < VA; ++j) a[j] = 0.f;
      }
      if (ok && b_ok) VecLoad<VB>::ld(dy + (int64_t)pout * CO + VB * i, b);
      else {
#pragma unroll
        for (int j = 0; j < VB; ++j) b[j] = 0.f;
      }
#pragma unroll
      for (int ja = 0; ja < VA; ++ja)
#pragma unroll
        for (int jb = 0; jb < VB; ++jb)
          acc[ja][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ja], b[jb], acc[ja][jb], 0, 0, 0);
    } else if constexpr (OT == VC_OPERAND_X6) {
      bw_group32_x6<CI, CO, WA, WB>(x, dy, qi, qo, qlen, i, q, a_ok, b_ok, acc);
    } else {
      bw_group16<CI, CO, OT>(x, dy, qi, qo, qlen, i, q, a_ok, b_ok, acc);
    }
  }

  // fixed-order cross-wave reduction through LDS (wave 0 stores, waves 1..3 add in order)
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int ja = 0; ja < VA; ++ja)
#pragma unroll
        for (int jb = 0; jb < VB; ++jb)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int m = q * 4 + reg, ci = BwChan<CI, WA>::of(m, ja), co = BwChan<CO, WB>::of(i, jb);
            if (m < MA && i < NB) {
              float* p = &red[ci * CO + co];
              *p = (wv == 0) ? acc[ja][jb][reg] : (*p + acc[ja][jb][reg]);
            }
          }
    }
    __syncthreads();
  }
  float* dst = partial + ((int64_t)split * kv + k) * (CI * CO);
  for (int e = threadIdx.x; e < CI * CO; e += 256) dst[e] = red[e];
}

// --------------------------------------------------------------------------------------------- K8 small channels (round 4)
// bwd_weight_kernel feeds v_mfma_f32_16x16x4_f32 with ONE channel per lane per operand: at C = 16 that is a 4-byte gather per lane
// and two vector-memory instructions for every MFMA (16 matrix-pipe cycles per load instruction; 9-16 % of the fp32 peak on the
// 16-channel layers, 1.2 % at C = 8 -- profiles/r04_kbench.txt).  v_mfma_f32_4x4x1_16B_f32 turns the shape around: 16 independent
// 4 x 4 x 1 outer products per instruction, one per PAIR.  Lane (p, i), p = lane / 4 the pair slot, i = lane % 4, loads VA = CI / 4
// CONTIGUOUS input channels [VA i, VA i + VA) of pair p's input row and VB = CO / 4 contiguous channels of its gradient row -- at
// C = 16 one 16-byte load each, a whole 64-byte row per 4 lanes -- and instruction (t, u) multiplies channel VA i + t of x by channel
// VB j + u of dy for all 16 pairs at once: VA VB instructions of 8 cycles per 16 pairs = the same matrix-pipe time as before, with 2
// (3 at 32 x 16) load instructions per 16 pairs instead of 8.  Each pair slot accumulates its own 4 x 4 tiles; the 16 slots are added
// once per wave at the end (xor-butterfly over lane / 4: a fixed tree).  Served shapes: CI, CO in {4, 8, 16, 32} with CI CO <= 512
// (VA VB <= 32 accumulator quads per lane).  Same queue compaction, block order, partial layout and split-N reduction as above;
// results differ from bwd_weight_kernel only in the order of the fp32 additions (bit-stable run to run).
template <int V>
__device__ __forceinline__ void ld_contig(const float* p, float* o) {
  if constexpr (V == 8) { VecLoad<4>::ld(p, o); VecLoad<4>::ld(p + 4, o + 4); }
  else VecLoad<V>::ld(p, o);
}

template <int CI, int CO>
__global__ void __launch_bounds__(256) bwd_weight_small_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const int32_t* __restrict__ tbl, int64_t n_out, int kv,
                                                               int64_t rows_per_block, int nsplit, int legacy_order,
                                                               float* __restrict__ partial, const int32_t* __restrict__ rep,
                                                               int centre, const float* __restrict__ dy_grp) {
  constexpr int VA = CI / 4, VB = CO / 4;
  static_assert(CI % 4 == 0 && CO % 4 == 0 && VA <= 8 && VB <= 8 && VA * VB <= 32, "small-channel weight gradient: CI CO <= 512");
  constexpr int U = (VA * VB >= 16) ? 1 : 2;          // groups of 16 pairs requested per trip
  __shared__ int q_in[4][160];
  __shared__ int q_out[4][160];
  __shared__ __attribute__((aligned(16))) float s_part[16 * (CI * CO + 16)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane >> 2, i = lane & 3;
  int k, split;
  if (legacy_order) {
    k = blockIdx.x / nsplit;
    split = blockIdx.x - k * nsplit;
  } else {   // the kv offset-blocks of one row range adjacent in launch order and on one XCD (see bwd_weight_kernel)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rl = j / kv;
    k = j - rl * kv;
    split = rl * 8 + xcd;
  }
  if (split >= nsplit) return;
  const bool reps_only = rep != nullptr && k != centre;
  if (reps_only) dy = dy_grp;
  const int64_t brow0 = (int64_t)split * rows_per_block;
  const int64_t bend = min(brow0 + rows_per_block, n_out);
  const int64_t rpw = rows_per_block / 4;
  const int64_t wstart = brow0 + wave * rpw;
  const int64_t wend = min(wstart + rpw, bend);

  f32x4 acc[VA][VB];
#pragma unroll
  for (int t = 0; t < VA; ++t)
#pragma unroll
    for (int u = 0; u < VB; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  int* qi = q_in[wave];
  int* qo = q_out[wave];
  int qlen = 0;
  auto entry = [&](int64_t r) -> int {
    if (r >= wend) return -1;
    const int v = tbl[(int64_t)k * n_out + r];
    return (reps_only && rep[r] != (int32_t)r) ? -1 : v;
  };
  auto consume = [&](const float (&a)[VA], const float (&b)[VB]) {
#pragma unroll
    for (int t = 0; t < VA; ++t)
#pragma unroll
      for (int u = 0; u < VB; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t], b[u], acc[t][u], 0, 0, 0);
  };
  int v_next = entry(wstart + lane);
  for (int64_t base = wstart; base < wend; base += 64) {
    const int64_t r = base + lane;
    const int v = v_next;
    v_next = entry(r + 64);
    const bool valid = v >= 0;
    const unsigned long long m = __ballot(valid);
    if (m == 0ULL) continue;
    const int pos = __popcll(m & ((1ULL << lane) - 1ULL));
    if (valid) { qi[qlen + pos] = v; qo[qlen + pos] = (int)r; }
    qlen += __popcll(m);
    __builtin_amdgcn_wave_barrier();
    const int ng = qlen >> 4;
    int g = 0;
    for (; g + U <= ng; g += U) {
      float a[U][VA], b[U][VB];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        const int pin = qi[(g + uu) * 16 + p], pout = qo[(g + uu) * 16 + p];
        ld_contig<VA>(x + (int64_t)pin * CI + VA * i, a[uu]);
        ld_contig<VB>(dy + (int64_t)pout * CO + VB * i, b[uu]);
      }
#pragma unroll
      for (int uu = 0; uu < U; ++uu) consume(a[uu], b[uu]);
    }
    for (; g < ng; ++g) {
      const int pin = qi[g * 16 + p], pout = qo[g * 16 + p];
      float a[VA], b[VB];
      ld_contig<VA>(x + (int64_t)pin * CI + VA * i, a);
      ld_contig<VB>(dy + (int64_t)pout * CO + VB * i, b);
      consume(a, b);
    }
    const int rem = qlen & 15;
    const int done = qlen - rem;
    int t1 = 0, t2 = 0;
    if (lane < rem) { t1 = qi[done + lane]; t2 = qo[done + lane]; }
    __builtin_amdgcn_wave_barrier();
    if (lane < rem) { qi[lane] = t1; qo[lane] = t2; }
    qlen = rem;
    __builtin_amdgcn_wave_barrier();
  }
  if (qlen > 0) {   // tail group: the pair slots beyond the queue contribute zeros
    const bool ok = p < qlen;
    const int pin = ok ? qi[p] : 0, pout = ok ? qo[p] : 0;
    float a[VA], b[VB];
    if (ok) {
      ld_contig<VA>(x + (int64_t)pin * CI + VA * i, a);
      ld_contig<VB>(dy + (int64_t)pout * CO + VB * i, b);
    } else {
#pragma unroll
      for (int t = 0; t < VA; ++t) a[t] = 0.f;
#pragma unroll
      for (int u = 0; u < VB; ++u) b[u] = 0.f;
    }
    consume(a, b);
  }
  // Every pair slot holds its own partial tiles: 16 slots x 4 waves to add per element.  Through LDS, wave after wave: the wave's
  // lanes store their VA VB accumulator quads slot-major, then thread e adds the 16 slots of element e in slot order (a fixed
  // order: wave 0 slots 0..15, wave 1 slots 0..15, ...).  (A first cut reduced the slots with a shuffle butterfly and let four
  // lanes per wave walk all elements serially: 64-128 dependent LDS read-modify-writes per block -- 2-8x SLOWER than
  // bwd_weight_kernel on every shape, profiles/r04_dw_small_channels.md.)
  // Tile (t, u), accumulator register r, lane % 4 = l holds D[row r][col l] = dW[ci = VA r + t][co = VB l + u].
  constexpr int E = CI * CO, SLOT = E + 16, EPT = (E + 255) / 256;
  float tot[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) tot[j] = 0.f;
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < VA; ++t)
#pragma unroll
        for (int u = 0; u < VB; ++u)
          *reinterpret_cast<f32x4*>(&s_part[p * SLOT + ((t * VB + u) * 4 + i) * 4]) = acc[t][u];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = threadIdx.x + 256 * j;
      if (e < E) {
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) tot[j] += s_part[sl * SLOT + e];
      }
    }
    __syncthreads();
  }
  float* dst = partial + ((int64_t)split * kv + k) * (CI * CO);
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = threadIdx.x + 256 * j;
    if (e < E) {
      const int tile = e >> 4, l = (e >> 2) & 3, r = e & 3;
      const int t = tile / VB, u = tile - t * VB;
      // D[row][col] of a 4x4 block: row = accumulator register, col = lane % 4 (confirmed on gfx950, r4m)
      dst[(VA * r + t) * CO + VB * l + u] = tot[j];
    }
  }
}

#ifdef VC_EXPERIMENTS
#include "experiments/bwd_weight_v2.inc"
#endif

// dweight[(co*kv + k)*CI + ci] = sum_s partial[s][k][ci][co]   (fixed order: 4 interleaved partial sums, then 0+1+2+3)
// 64 consecutive (k, ci, co) elements per block in the partial's native order (coalesced reads), 4 split-groups.
__global__ void __launch_bounds__(256) bwd_weight_reduce_kernel(const float* __restrict__ partial, int nsplit, int kv,
                                                                int ci_n, int co_n, float* __restrict__ dweight) {
  __shared__ float red[4][64];
  const int total = kv * ci_n * co_n;
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sg = threadIdx.x >> 6;
  float s = 0.f;
  if (e < total) {
    // same addition order as a plain loop; the loads of four trips are issued together (one dependent load per trip made
    // this 13 us kernel latency-bound)
    int sp = sg;
    for (; sp + 12 < nsplit; sp += 16) {
      const float v0 = partial[(int64_t)sp * total + e], v1 = partial[(int64_t)(sp + 4) * total + e];
      const float v2 = partial[(int64_t)(sp + 8) * total + e], v3 = partial[(int64_t)(sp + 12) * total + e];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; sp < nsplit; sp += 4) s += partial[(int64_t)sp * total + e];
  }
  red[sg][threadIdx.x & 63] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    const int co = e % co_n;
    const int ci = (e / co_n) % ci_n;
    const int k = e / (co_n * ci_n);
    dweight[((int64_t)co * kv + k) * ci_n + ci] = v;
  }
}

// All split-N reductions of a reverse sweep in ONE launch (round 3).  The per-layer reduce above is a 14 us latency-bound launch for
// <= 14 MB of partial sums, 20 of them per train step (0.28 ms of side-stream kernel time, 3.4 % of the step's kernel time); the
// feature pass gives every weight gradient its own partial buffer and reduces them all here, after its last weight-gradient
// kernel: the same 64-element blocks, the same fixed addition order (bit-identical results), one launch whose blocks overlap
// each other's latency.
static constexpr int kMaxBwDefer = 48;
struct BwReduceDesc {
  const float* partial;
  float* dweight;
  int nsplit, kv, ci, co, block0;   // block0: first block of this tensor in the launch
};
struct BwReduceArgs {
  BwReduceDesc d[kMaxBwDefer];
  int n;
};

__global__ void __launch_bounds__(256) bwd_weight_reduce_multi_kernel(BwReduceArgs a) {
  __shared__ float red[4][64];
  int u = 0;
  for (int j = 1; j < a.n; ++j)
    if ((int)blockIdx.x >= a.d[j].block0) u = j;
  const BwReduceDesc D = a.d[u];
  const float* __restrict__ partial = D.partial;
  const int total = D.kv * D.ci * D.co;
  const int e = ((int)blockIdx.x - D.block0) * 64 + (threadIdx.x & 63);
  const int sg = threadIdx.x >> 6;
  float s = 0.f;
  if (e < total) {
    int sp = sg;
    for (; sp + 12 < D.nsplit; sp += 16) {
      const float v0 = partial[(int64_t)sp * total + e], v1 = partial[(int64_t)(sp + 4) * total + e];
      const float v2 = partial[(int64_t)(sp + 8) * total + e], v3 = partial[(int64_t)(sp + 12) * total + e];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; sp < D.nsplit; sp += 4) s += partial[(int64_t)sp * total + e];
  }
  red[sg][threadIdx.x & 63] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    const int co = e % D.co;
    const int ci = (e / D.co) % D.ci;
    const int k = e / (D.co * D.ci);
    D.dweight[((int64_t)co * D.kv + k) * D.ci + ci] = v;
  }
}

// deferral list of the calling thread (set by the feature pass around its side-stream weight gradients; NULL = reduce at once)
static thread_local BwReduceArgs* g_bw_defer = nullptr;

#ifdef VC_EXPERIMENTS
#include "experiments/group_sum_fixed.inc"
#endif

// --------------------------------------------------------------------------------------------- kernel timing (vc_trace_*)
__global__ void __launch_bounds__(256) count_pairs_kernel(const int32_t* __restrict__ tbl, int64_t total, int64_t* __restrict__ out) {
  // 16 entries per thread per trip as four independent 16-byte loads (the first version walked the table one int per trip:
  // 52 us for 21 MB, inside the timed step)
  const int4* t4 = reinterpret_cast<const int4*>(tbl);
  const int64_t n4 = total >> 2;
  int c = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; e + 3 * stride < n4; e += 4 * stride) {
    int4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = t4[e + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) c += (v[u].x >= 0) + (v[u].y >= 0) + (v[u].z >= 0) + (v[u].w >= 0);
  }
  for (; e < n4; e += stride) {
    const int4 v = t4[e];
    c += (v.x >= 0) + (v.y >= 0) + (v.z >= 0) + (v.w >= 0);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(total & 3)) c += tbl[(n4 << 2) + threadIdx.x] >= 0 ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd((unsigned long long*)out, (unsigned long long)c);
}

struct TraceState {
  bool on = false;
  int dir = 0, ck = 0, cn = 0, cap = 0, n = 0;
  int64_t* dev_pairs = nullptr;
  hipEvent_t* ev = nullptr;  // 2 per record, created on first use, kept for the life of the process
  int n_ev = 0;
  vc_trace_record* rec = nullptr;
};
static TraceState g_trace;
static int g_trace_counted = 16;   // records of a trace whose pairs are counted exactly (set per trace in vc_trace_begin)
static bool g_last_windowed = false;  // set by launch_gg: the launch just issued was the LDS row-window kernel

// -> record slot (its start event is on the stream) or -1
static inline int trace_open(int dir, int ck, int cn, hipStream_t st) {
  TraceState& T = g_trace;
  // T.dir == -1: every gather-GEMM (0 forward, 1 backward-input) and weight-gradient (2) launch is recorded
  if (!T.on || T.n >= T.cap || (T.dir != -1 && (T.dir != dir || T.ck != ck || T.cn != cn))) return -1;
  const int i = T.n;
  if (hipEventRecord(T.ev[2 * i], st) != hipSuccess) return -1;
  return i;
}
static inline void trace_close(int i, int dir, int ck, int cn, const int32_t* tbl, int kv, int64_t n_src, int64_t n_out,
                               hipStream_t st) {
  TraceState& T = g_trace;
  if (hipEventRecord(T.ev[2 * i + 1], st) != hipSuccess) return;
  T.rec[i] = vc_trace_record{0.f, kv, ck, cn, (dir != 2 && g_last_windowed) ? 1 : 0, n_src, n_out, 0, dir, 0.f};
  // The pair count walks the whole table (40 us for 21 MB): it is measurement work INSIDE the timed step, so only the first
  // g_trace_counted (16 for a one-kernel trace, 128 for direction -1) records of a trace count exactly; later ones take the pairs-per-row ratio of the last counted record of the same
  // kernel shape (layer discard draws a new permutation per step: the ratio moves by < 1 %).  pairs = -1 marks "to be estimated".
  if (i < g_trace_counted) {
    int64_t nb = cdiv((int64_t)kv * n_out, 256 * 16);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(count_pairs_kernel, dim3((unsigned)nb), dim3(256), 0, st, tbl, (int64_t)kv * n_out, T.dev_pairs + i);
    (void)hipGetLastError();
  } else {
    T.rec[i].pairs = -1;
  }
  T.n = i + 1;
}

int trace_open_aux(int dir, int c, hipStream_t st) {
  TraceState& T = g_trace;
  if (!T.on || T.n >= T.cap || T.dir != dir || (T.ck != 0 && T.ck != c)) return -1;
  const int i = T.n;
  if (hipEventRecord(T.ev[2 * i], st) != hipSuccess) return -1;
  return i;
}
void trace_close_aux(int i, int dir, int c, int64_t n, hipStream_t st) {
  TraceState& T = g_trace;
  if (hipEventRecord(T.ev[2 * i + 1], st) != hipSuccess) return;
  T.rec[i] = vc_trace_record{0.f, 0, c, c, 0, n, n, 0, dir, 0.f};
  T.n = i + 1;
}

// --------------------------------------------------------------------------------------------- dispatch
static constexpr int kRT = 2;  // 32 rows per wave, 128 rows per 256-thread block
int g_conv_variant = 2;        // 1 = gather_gemm_kernel (per-wave loads), 2 = gather_gemm_v2_kernel (LDS-staged, pipelined)
int g_conv_rt = 0;             // v2 row tiles per wave: 1 (64 rows/block) | 2 (128 rows/block) | 0 = heuristic

// Waves per block of the direct kernel where both channel counts are >= 16 (vc_debug_set conv_nw): 4 | 8 | 0 = per shape.
// Measured on the VirConv-L layers (gpurun r2m, tools/kbench.py --nw): 8-wave blocks lift the LDS-bound occupancy of the
// 64-channel instantiations from 6 to 8 waves per SIMD, yet only <CK=32, CN=64, backward-input> gets faster (235 -> 202 us,
// 103 -> 96 us); the forward kernels tie (1207 vs 1210 us per pass) and the strided backward tables lose 5-7 %.
int g_conv_nw = 0;
// ... and for SMALL launches (vc_debug_set conv_nw8_below: output rows below which every >= 16-channel shape takes 8-wave blocks; 0 = never).
// VirConv8x at its benchmark size (16 000 voxels per frame: launches of 20-60 k rows = 300-900 four-wave blocks for 256 CUs) gains 1.5-2 % with
// 8-wave blocks below 62 000 rows (3.437 / 3.436 -> 3.385 / 3.374 ms per step; 40 000: 3.39, 100 000: 3.36-3.38, everywhere: 3.38-3.39; r6an);
// VirConv-L's launches (64-310 k rows) are all above the threshold and lose with 8 waves everywhere (4.03 -> 4.14 ms, same call).  Results per
// row do not depend on the block shape; the BatchNorm partial sums are grouped per wave, so their last bits do (fixed per shape, run to run).
int g_conv_nw8_below = 62000;
// developer switch conv_autopack (tools/kbench.py --autopack): the stand-alone conv entry points repack the weights into a
// library-owned scratch right before the launch, so that a kernel can be timed with a fragment-ordered image without a pass
// executor around it.  The product path never allocates: the pass executor packs into its caller-provided arena.
int g_conv_autopack = 0;
static constexpr size_t kAutopackBytes = (size_t)27 * 64 * 64 * sizeof(float);
static float* autopack_scratch(bool bwd) {
  static float* buf[2] = {nullptr, nullptr};
  float*& b = buf[bwd ? 1 : 0];
  if (b == nullptr && hipMalloc((void**)&b, kAutopackBytes) != hipSuccess) b = nullptr;
  return b;
}
// Wave-autonomous kernel (v4): vc_debug_set conv_v4 = 0 never (default) | 1 every eligible shape (both channel counts multiples
// of 16, fp32 operands) | 2 = per launch (the rule below, from tools/kbench.py --v4 A/B runs)
int g_conv_v3_split = 1;   // vc_debug_set conv_v3_split: the window kernel (experiments) takes the bf16-split products when f32_split is on
int g_conv_v4 = 0;   // off: inside the train step (weight-gradient stream contending for the same CUs) the table ties or loses, 5.64 vs 5.60 ms
// dx shift in the LDS-staged kernel (vc_debug_set conv_dxs; needs a weight image).  OFF: measured slower although it halves the
// gathered rows -- SubM 64->32 194 -> 231 us, 32->32 113 -> 129 us, 16->16 65 -> 88 us, train step 5.60 -> 5.74 ms
// (tools/kbench.py --autopack --dxs 0|1).  The third fragment set costs the 64-channel shapes three waves per SIMD (62 -> 94
// VGPRs; forcing 6 waves spills), but the 32-channel shapes keep all eight and lose as well: the 32 DPP / select operations per
// side offset sit between the loads' arrival and the MFMAs.  Fewer gathered rows alone do not buy time.
int g_conv_dxs = 0;
// vc_debug_set f32_split: 1 (default) = fp32 products of the gather-GEMM on the bf16 matrix cores as six split terms (split3) for every
// shape whose channel counts are multiples of 16 (measured per layer: never slower, profiles/r04_split_products.md);
// 0 = v_mfma_f32_16x16x4_f32 (exact products, the round 1-3 kernels)
int g_f32_split = 1;
int g_conv_v5 = 0;             // developer: loader / MFMA wave-role kernel (plain launches with a weight image)
int g_conv_v4_pf = 1;          // developer: gather prefetch distance of the v4 kernel (1 | 2 | 4), plain launches with an image only
int g_conv_v4_ablate = 0;      // developer ablations of the v4 kernel (see its ABL parameter); results are wrong when set
// Measured (tools/kbench.py --v4 0|1 --autopack, VirConv-L bs 4, profiles/r02_kbench_v4.txt): a wave per 64 rows needs about
// 2.3 of them per SIMD to hide its own latencies -- the 195 k / 310 k-row layers of stages 2 and 3 gain 5-20 % (SubM 16->16
// 65 -> 53 us, 32->32 116 -> 101 us, 64->32 196 -> 188 us), the 76 k-row layers of stage 4 lose 20-30 % -- and the strided
// backward tables (row-ordered) are better off with the LDS-staged kernel's 8-wave blocks.
static constexpr int64_t kV4MinRows = 150000;
static inline bool conv_use_v4(int ck, int cn, bool bwd, int64_t rows, bool ordered) {
  if (ck % 16 != 0 || cn % 16 != 0 || g_conv_v4 == 0) return false;
  if (g_conv_v4 == 1) return true;
  (void)bwd;
  return rows >= kV4MinRows && !ordered;
}
// rows = output rows of the launch; ordered = the launch takes a row order (vc_row_order)
static inline int conv_block_waves(int ck, int cn, bool bwd, int64_t rows, bool ordered) {
  if (ck < 16 || cn < 16) return 4;
  if (conv_use_v4(ck, cn, bwd, rows, ordered)) return 4;  // 64 rows per workgroup: the partial-row count of a 4-wave v2 block
  if (g_conv_nw == 8) return 8;
  if (g_conv_nw == 4) return 4;
  if (rows < g_conv_nw8_below) return 8;
  return (bwd && ck == 32 && cn == 64) ? 8 : 4;
}
// Pair-compacted forward kernel (gather_gemm_pc_kernel) for tables with few active offsets per row.  The library recognises them
// by n_in != n_out: a strided conv or its inverse (a SubM table has n_in == n_out; a strided table that happens to keep the row
// count just stays on v2).  vc_debug_set "conv_pc": 1 = take it.  OFF by default -- measured (profiles/r03_pair_compacted_kernel.md):
// it issues 2-5x fewer MFMA tiles and gathers than v2 and still only ties it on the stage-3 conv (171 vs 170 us) and loses on the
// others (stage 2: 113 vs 73, stage 4: 232 vs 167, conv_out: 29 vs 24 us): the offsets of a block run in lock step, so a block's
// critical path is 27 x (one tile on one or two of its eight waves) whatever the fill, and only 1-2 such blocks fit a CU next to
// their 35 KB of LDS accumulators, where v2 keeps six blocks with every wave busy.
int g_conv_pc = 0;
static inline bool conv_use_pc(int ck, int cn, int kv, int64_t n_in, int64_t n_out, int ot) {
  return g_conv_pc && g_conv_variant == 2 && ck % 16 == 0 && cn % 16 == 0 && ck <= 64 && cn <= 64 && kv > 1 && kv <= 32 &&
         n_in != n_out && ot == VC_OPERAND_F32 && n_in * (int64_t)ck * 4 < (1LL << 31);
}
int g_conv_window = 1;         // 0 = never take the LDS-window kernel (A/B measurements)
int g_conv_wdma = 0;           // 1 = W images through the LDS-DMA engine in the window kernel
int g_conv_winrows = 32;       // 24 = smaller per-wave windows (one more block per CU at 64 channels)

// the LDS row-window kernel (v3) serves: fp32 operands, >= 16 source channels, natural row order, no duplicate-pixel rule,
// tables the caller flags as coordinate-sorted (VC_CONV_SORTED_ROWS)
template <int CK>
static inline bool use_window_kernel(int flags, int ot, const int32_t* rep, const int32_t* order, int64_t n_src, int kv,
                                     const float* src_centre) {
  return CK >= 16 && g_conv_window && g_conv_variant == 2 && (flags & VC_CONV_SORTED_ROWS) && ot == VC_OPERAND_F32 &&
         rep == nullptr && order == nullptr && src_centre == nullptr && kv <= 32 && n_src * CK * 4 < (1LL << 31);
}

// ---- in-kernel BatchNorm finish: request hand-over (see BnFinishRequest in common.h) and the ticket pool
// vc_debug_set "conv_bn_finish": 1 = the epilogue launches finish the BatchNorm sums themselves (conv_finish_tail); 0 = partial rows
// are left to the BatchNorm kernels; 2 = tests: partial rows, then bn_finish_reference_kernel (the same three levels, one block).
int g_conv_bn_finish = 1;
static std::atomic<long long> g_fin_launches{0};   // vc_debug_get "conv_bn_finish_launches"
// vc_debug_get "conv_launch_seq": conv kernel launches (gather-GEMM and weight gradient, any operand type) enqueued by this process so
// far.  A caller that recorded an event behind a feature pass can tell later whether any conv kernel was enqueued since (backbone.py:
// the event the pixel projection waits for, LOG.md A.17).
static std::atomic<long long> g_conv_launch_seq{0};
struct FinState {
  bool armed = false, taken = false;
  BnFinishRequest req{};
  int nw = 0, nblocks = 0, cn = 0;   // the qualifying launch since the last arm (0: none)
  float* partial = nullptr;
};
static thread_local FinState t_fin;
void conv_finish_arm(const BnFinishRequest& r) {
  t_fin = FinState{};
  t_fin.armed = true;
  t_fin.req = r;
}
// Tickets: kFinSlots x kFinSlotWords zero words per device, allocated once; launches take the slots round-robin.  A slot is
// all-zero again when its launch ends (the last block clears it); launches of one stream are ordered, so a slot could only be
// handed to a second launch in flight if kFinSlots finishing launches were outstanding on concurrent streams at once.
static constexpr int kFinSlots = 256, kFinSlotWords = 80;   // 1 + <= 64 group tickets
static unsigned* fin_ticket_slot() {
  static std::mutex mu;
  static unsigned* base[32] = {};
  static unsigned next = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (base[dev] == nullptr) {
    void* p = nullptr;
    const size_t bytes = (size_t)kFinSlots * kFinSlotWords * sizeof(unsigned);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
    base[dev] = (unsigned*)p;
  }
  return base[dev] + (size_t)(next++ % kFinSlots) * kFinSlotWords;
}
static void fin_fill(ConvFinish& f, const BnFinishRequest& r, float* partial, int nblocks) {
  f.brows = reinterpret_cast<double*>(partial);
  f.dpartial = r.dpartial;
  f.o0 = r.o0; f.o1 = r.o1; f.r0 = r.r0; f.r1 = r.r1;
  f.nbt = r.nbt;
  f.n = (long long)r.n;
  f.momentum = r.momentum;
  f.bwd = r.bwd ? 1 : 0;
  f.nblocks = nblocks;
  f.rpb = (int)cdiv((int64_t)nblocks, 64);
  f.ngroups = (int)cdiv((int64_t)nblocks, (int64_t)f.rpb);
}
// attach the armed request to this launch's epilogue if the launch qualifies; `lds` = the launch's dynamic LDS bytes
static void fin_attach(ConvEpilogue& epi, int epi_kind, unsigned nblocks, int nw, int cn, size_t& lds) {
  if (!t_fin.armed || t_fin.taken || t_fin.nw != 0 || !g_conv_bn_finish) return;
  const bool bwd = epi_kind == VC_EPI_BWD;
  if (!(epi_kind == VC_EPI_STATS || (bwd && epi.y_raw != nullptr)) || (t_fin.req.bwd != 0) != bwd) return;
  if ((int64_t)nblocks * nw <= 512 || (cn & (cn - 1)) != 0 || cn < 8 || cn > 64 || t_fin.req.dpartial == nullptr) return;
  if (bwd ? (t_fin.req.r0 == nullptr) : (t_fin.req.o0 == nullptr || t_fin.req.o1 == nullptr)) return;
  t_fin.nw = nw;
  t_fin.nblocks = (int)nblocks;
  t_fin.cn = cn;
  t_fin.partial = epi.partial;
  if (g_conv_bn_finish != 1) return;   // 2: the caller's conv_finish_take launches the reference kernel behind this launch
  unsigned* cnt = fin_ticket_slot();
  if (cnt == nullptr) return;
  epi.fin.cnt = cnt;
  fin_fill(epi.fin, t_fin.req, epi.partial, (int)nblocks);
  if (lds < 4096 + 16) lds = 4096 + 16;
  t_fin.taken = true;
  g_fin_launches.fetch_add(1, std::memory_order_relaxed);
}
bool conv_finish_take(hipStream_t st) {
  bool done = t_fin.taken;
  if (!done && t_fin.armed && g_conv_bn_finish == 2 && t_fin.nw != 0) {   // tests: the same arithmetic by one block
    ConvFinish f;
    fin_fill(f, t_fin.req, t_fin.partial, t_fin.nblocks);
    switch (t_fin.cn) {
#define VC_FR(C_) case C_: hipLaunchKernelGGL((bn_finish_reference_kernel<C_>), dim3(1), dim3(256), 0, st, f, (const float*)t_fin.partial, t_fin.nw); break
      VC_FR(8); VC_FR(16); VC_FR(32); VC_FR(64);
#undef VC_FR
    }
    done = hipGetLastError() == hipSuccess;
  }
  t_fin = FinState{};
  return done;
}

template <int CK, int CN, bool BWD>
static int launch_gg(const float* src, const float* src_centre, int64_t n_src, const int32_t* tbl, const float* w,
                     float* out, const int32_t* rep, const int32_t* order, int64_t n_out, int kv, int centre, int mirror, int ot,
                       int epi_kind, const ConvEpilogue& epi_in, int flags, hipStream_t st) {
  const int64_t rows_per_block = 4 * kRT * 16;
  ConvEpilogue epi = epi_in;
  g_last_windowed = false;
  g_conv_launch_seq.fetch_add(1, std::memory_order_relaxed);
  // fragment-ordered weight image registered for this weight tensor and direction (vc_conv_pack_weights), or -- developer
  // switch conv_autopack, for stand-alone kernel timing -- packed right here into a scratch the library owns
  const float* wpk = nullptr;
  if constexpr (CK % 16 == 0 && CN % 16 == 0) {
    if (ot == VC_OPERAND_F32 && kv <= 32) {
      wpk = packed_lookup(w, BWD);
      if (wpk == nullptr && (g_conv_autopack || g_f32_split)) {   // (split products read the fragment-ordered image on every route)
        float* scratch = autopack_scratch(BWD);
        if (scratch != nullptr && (size_t)kv * CK * CN * sizeof(float) <= kAutopackBytes) {
          PackArgs pa;
          pa.d[0] = PackDesc{w, scratch, CK, CN, kv, BWD ? 1 : 0};
          hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)cdiv((int64_t)kv * CK * CN, 1024), 1), dim3(256), 0, st, pa);
          VC_CHECK_LAUNCH("pack_weights_kernel");
          wpk = scratch;
        }
      }
    }
  }
#ifdef VC_EXPERIMENTS
  if constexpr (!BWD && CK % 16 == 0 && CN % 16 == 0) {
    if (conv_use_pc(CK, CN, kv, n_src, n_out, ot) && rep == nullptr && src_centre == nullptr && !(flags & VC_CONV_SRC_INTERLEAVED) &&
        (epi_kind == VC_EPI_NONE || epi_kind == VC_EPI_STATS || epi_kind == VC_EPI_AFFINE)) {
      constexpr int NFRAG_ = (CK / 16) * (CN / 16) * 64;
      const size_t ldsp = (size_t)2 * NFRAG_ * 16 + (size_t)128 * (CN + 4) * 4 + (size_t)kv * 128 * 4 + 128 + 144 + (size_t)kv * 128 + 16;
      const dim3 gridp((unsigned)cdiv(n_out, 128));
#define VC_LP(E_)                                                                                                              \
  do {                                                                                                                         \
    if (wpk) {                                                                                                                 \
      static const hipError_t attr_ = hipFuncSetAttribute((const void*)gather_gemm_pc_kernel<CK, CN, E_, true>,                \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);            \
      (void)attr_;                                                                                                             \
      hipLaunchKernelGGL((gather_gemm_pc_kernel<CK, CN, E_, true>), gridp, dim3(512), ldsp, st, src, n_src, tbl, wpk, out,     \
                         n_out, kv, epi);                                                                                      \
    } else {                                                                                                                   \
      static const hipError_t attr_ = hipFuncSetAttribute((const void*)gather_gemm_pc_kernel<CK, CN, E_, false>,               \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);            \
      (void)attr_;                                                                                                             \
      hipLaunchKernelGGL((gather_gemm_pc_kernel<CK, CN, E_, false>), gridp, dim3(512), ldsp, st, src, n_src, tbl, w, out,      \
                         n_out, kv, epi);                                                                                      \
    }                                                                                                                          \
  } while (0)
      if (epi_kind == VC_EPI_STATS) VC_LP(VC_EPI_STATS);
      else if (epi_kind == VC_EPI_AFFINE) VC_LP(VC_EPI_AFFINE);
      else VC_LP(VC_EPI_NONE);
#undef VC_LP
      VC_CHECK_LAUNCH("gather_gemm_pc_kernel");
      return VC_OK;
    }
  }
  if constexpr (CK % 16 == 0 && CN % 16 == 0 && ((CK / 16) * (CN / 16)) % 4 == 0) {
    if (g_conv_v5 && wpk && epi_kind == VC_EPI_NONE && kv <= 32 && n_src * CK * 4 < (1LL << 31)) {
      constexpr size_t lds_ab = (size_t)2 * 4 * (CK / 16) * 1024 + (size_t)2 * (CK / 16) * (CN / 16) * 1024;
      const size_t lds5 = lds_ab + (size_t)(kv + 1) * 64 * sizeof(int) + 64;
      if (lds5 <= 64 * 1024) {
        hipLaunchKernelGGL((gather_gemm_v5_kernel<CK, CN, VC_EPI_NONE>), dim3((unsigned)cdiv(n_out, 64)), dim3(512), lds5, st, src,
                           src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);
        VC_CHECK_LAUNCH("gather_gemm_v5_kernel");
        return VC_OK;
      }
    }
  }
  if constexpr (CK % 16 == 0 && CN % 16 == 0) {
    if (conv_use_v4(CK, CN, BWD, n_out, order != nullptr) && ot == VC_OPERAND_F32 && kv <= 32 && n_src * CK * 4 < (1LL << 31) &&
        (int64_t)kv * CK * CN * 4 < (1LL << 31)) {
      const size_t lds4 = (size_t)(kv + 1) * 64 * sizeof(int);
      const dim3 grid4((unsigned)cdiv(n_out, 64));
#define VC_L4(E_)                                                                                                              \
  do {                                                                                                                         \
    if (wpk) hipLaunchKernelGGL((gather_gemm_v4_kernel<CK, CN, BWD, E_, true>), grid4, dim3(64), lds4, st, src, src_centre, n_src, \
                                tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);                                    \
    else hipLaunchKernelGGL((gather_gemm_v4_kernel<CK, CN, BWD, E_, false>), grid4, dim3(64), lds4, st, src, src_centre, n_src, \
                            tbl, w, out, rep, order, n_out, kv, centre, mirror, epi);                                          \
  } while (0)
      bool done = true;
#ifdef VC_EXPERIMENTS   // ablations and prefetch-depth variants of v4: measured, kept out of the default build
      if constexpr (!BWD && CN == 32 && (CK == 64 || CK == 32)) {
        if (g_conv_v4_ablate && wpk && epi_kind == VC_EPI_NONE) {
#define VC_L4A(A_) hipLaunchKernelGGL((gather_gemm_v4_kernel<CK, CN, false, VC_EPI_NONE, true, A_, 1>), grid4, dim3(64), lds4, st, src, \
                                      src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi)
          switch (g_conv_v4_ablate) {
            case 1: VC_L4A(1); break;
            case 2: VC_L4A(2); break;
            case 3: VC_L4A(3); break;
            default: VC_L4A(4); break;
          }
#undef VC_L4A
          VC_CHECK_LAUNCH("gather_gemm_v4_kernel<ablation>");
          return VC_OK;
        }
      }
      if (epi_kind == VC_EPI_NONE && wpk && (g_conv_v4_pf == 2 || g_conv_v4_pf == 4 || g_conv_v4_pf == 11)) {
        if (g_conv_v4_pf == 11)
          hipLaunchKernelGGL((gather_gemm_v4_kernel<CK, CN, BWD, VC_EPI_NONE, true, 0, 11>), grid4, dim3(64), lds4, st, src, src_centre,
                             n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);
        else if (g_conv_v4_pf == 2)
          hipLaunchKernelGGL((gather_gemm_v4_kernel<CK, CN, BWD, VC_EPI_NONE, true, 0, 2>), grid4, dim3(64), lds4, st, src, src_centre,
                             n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);
        else
          hipLaunchKernelGGL((gather_gemm_v4_kernel<CK, CN, BWD, VC_EPI_NONE, true, 0, 4>), grid4, dim3(64), lds4, st, src, src_centre,
                             n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);
      }
      else
#endif
      if (epi_kind == VC_EPI_NONE) VC_L4(VC_EPI_NONE);
      else if (BWD && epi_kind == VC_EPI_BWD) { if constexpr (BWD) VC_L4(VC_EPI_BWD); }
      else if (!BWD && epi_kind == VC_EPI_STATS) { if constexpr (!BWD) VC_L4(VC_EPI_STATS); }
      else if (!BWD && epi_kind == VC_EPI_AFFINE) { if constexpr (!BWD) VC_L4(VC_EPI_AFFINE); }
      else done = false;
#undef VC_L4
      if (done) {
        VC_CHECK_LAUNCH("gather_gemm_v4_kernel");
        return VC_OK;
      }
    }
  }
  if constexpr (CK >= 16) {
    if (conv_block_waves(CK, CN, BWD, n_out, order != nullptr) == 4 && epi_kind != VC_EPI_BWD && use_window_kernel<CK>(flags, ot, rep, order, n_src, kv, src_centre)) {
      constexpr int NCH = CK / 16, NT = (CN + 15) / 16;
      // experiment switches (vc_debug_set): conv_wdma = W images through the LDS-DMA engine, conv_winrows = 24-row windows
      const bool wdma = g_conv_wdma && CN % 16 == 0 && epi_kind == VC_EPI_NONE;
      const int winrows = (g_conv_winrows == 24 && epi_kind == VC_EPI_NONE) ? 24 : 32;
      // round 6: the six-term bf16 split on the window kernel (vc_debug_set conv_v3_split, default 1 = follow f32_split)
      bool x6 = false;
      if constexpr (CK % 32 == 0 && CN % 16 == 0) x6 = g_f32_split && g_conv_v3_split && ot == VC_OPERAND_F32 && !wdma;
      const size_t lds = (size_t)(wdma ? 3 : 2) * NCH * NT * 64 * (x6 ? 24 : 16) + (size_t)kv * 64 * sizeof(int) +
                         (size_t)4 * (winrows + 1) * (CK + 4) * sizeof(float) + 16;
      const dim3 grid((unsigned)cdiv(n_out, 64));
#define VC_ARGS3 src, n_src, tbl, w, out, n_out, kv, mirror, epi
#define VC_L3(B_, E_, D_, R_) hipLaunchKernelGGL((gather_gemm_v3_kernel<CK, CN, B_, E_, D_, R_>), grid, dim3(256), lds, st, VC_ARGS3)
      if constexpr (CK % 32 == 0 && CN % 16 == 0) {
        if (x6) {
#define VC_L3X(B_, E_, R_) hipLaunchKernelGGL((gather_gemm_v3_kernel<CK, CN, B_, E_, false, R_, VC_OPERAND_X6>), grid, dim3(256), lds, st, VC_ARGS3)
          if (epi_kind == VC_EPI_NONE) { if (winrows == 24) VC_L3X(BWD, VC_EPI_NONE, 24); else VC_L3X(BWD, VC_EPI_NONE, 32); }
          else if constexpr (!BWD) { if (epi_kind == VC_EPI_STATS) VC_L3X(false, VC_EPI_STATS, 32); else VC_L3X(false, VC_EPI_AFFINE, 32); }
#undef VC_L3X
          VC_CHECK_LAUNCH("gather_gemm_v3_kernel<split bf16>");
          g_last_windowed = true;
          return VC_OK;
        }
      }
      if (epi_kind == VC_EPI_NONE) {
        if constexpr (CN % 16 == 0) {
          if (wdma && winrows == 24) VC_L3(BWD, VC_EPI_NONE, true, 24);
          else if (wdma) VC_L3(BWD, VC_EPI_NONE, true, 32);
          else if (winrows == 24) VC_L3(BWD, VC_EPI_NONE, false, 24);
          else VC_L3(BWD, VC_EPI_NONE, false, 32);
        } else {
          VC_L3(BWD, VC_EPI_NONE, false, 32);
        }
      } else if constexpr (!BWD) {
        if (epi_kind == VC_EPI_STATS) VC_L3(false, VC_EPI_STATS, false, 32);
        else VC_L3(false, VC_EPI_AFFINE, false, 32);
      }
#undef VC_L3
#undef VC_ARGS3
      VC_CHECK_LAUNCH("gather_gemm_v3_kernel");
      g_last_windowed = true;
      return VC_OK;
    }
  }
#endif
#ifdef VC_EXPERIMENTS
  if constexpr (CK % 16 == 0 && CN % 16 == 0 && !BWD) {
    if (flags & VC_CONV_SRC_INTERLEAVED) {   // round-3 experiment: see the kernel's IL parameter
      if (!(wpk && epi_kind == VC_EPI_NONE && ot == VC_OPERAND_F32 && src_centre == nullptr && kv <= 32 &&
            ((n_src + 15) & ~(int64_t)15) * CK * 4 < (1LL << 31))) {
        set_error("gather-GEMM: VC_CONV_SRC_INTERLEAVED needs a packed weight image, fp32 operands, no epilogue");
        return VC_EINVAL;
      }
      constexpr int NCH_ = CK / 16, NT_ = CN / 16;
      const size_t lds_il = (size_t)2 * NCH_ * NT_ * 64 * 4 * sizeof(float) + (size_t)(kv + 1) * 64 * sizeof(int) + 16;
      hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, false, 1, VC_OPERAND_F32, VC_EPI_NONE, 4, true, false, true>),
                         dim3((unsigned)cdiv(n_out, 64)), dim3(256), lds_il, st, src, src_centre, n_src, tbl, wpk, out, rep, order,
                         n_out, kv, centre, mirror, epi);
      VC_CHECK_LAUNCH("gather_gemm_v2_kernel<interleaved source>");
      return VC_OK;
    }
  }
#endif
  if (flags & VC_CONV_SRC_INTERLEAVED) {
    set_error("gather-GEMM: VC_CONV_SRC_INTERLEAVED is an experiment: forward convs with channel counts that are multiples of 16, library built with -DVC_EXPERIMENTS");
    return VC_EINVAL;
  }
  // dx shift (see the kernel's DXS parameter): SubM-shaped 27-offset tables in natural row order, where it has something to find
#ifdef VC_EXPERIMENTS
  const bool dxs = g_conv_dxs && wpk != nullptr && kv == 27 && n_src == n_out && order == nullptr && rep == nullptr &&
                   src_centre == nullptr;
#endif
  if (g_conv_variant == 2 && kv <= 32 && n_src * CK * 4 < (1LL << 31)) {
    constexpr int V = (CK >= 16) ? 4 : CK / 4;
    constexpr int NCH = CK / (4 * V);
    constexpr int NT = (CN + 15) / 16;
    // fp32 products as six bf16 MFMA terms (vc_debug_set f32_split; see split3): the same block shapes, epilogues and weight image as
    // the fp32 kernels below, operand type X6
    if constexpr (CK % 16 == 0 && CN % 16 == 0) {
      if (g_f32_split && ot == VC_OPERAND_F32 && wpk != nullptr) {
        const bool w8 = conv_block_waves(CK, CN, BWD, n_out, order != nullptr) == 8 && epi_kind != VC_EPI_AFFINE;
        size_t ldsx = (size_t)2 * NCH * NT * 64 * V * 6 + (size_t)(kv + 1) * (w8 ? 128 : 64) * sizeof(int) + 16;
        const dim3 gridx((unsigned)cdiv(n_out, w8 ? 128 : 64));
        fin_attach(epi, epi_kind, gridx.x, w8 ? 8 : 4, CN, ldsx);
#ifdef VC_EXPERIMENTS
#define VC_LX_DXS(B_, E_)                                                                                                         \
    if (dxs && !w8) {  /* round 6: the dx shift (half the gathered rows) with split products: measured in profiles/r06_dxs_split.md */ \
      hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_X6, E_, 4, true, true>), gridx, dim3(256), ldsx, st, src, \
                         src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);                           \
      break;                                                                                                                      \
    }
#else
#define VC_LX_DXS(B_, E_)
#endif
#define VC_LX(B_, E_)                                                                                                             \
  do {                                                                                                                            \
    VC_LX_DXS(B_, E_)                                                                                                             \
    if (w8) hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_X6, E_, 8, true>), gridx, dim3(512), ldsx, st, src, \
                               src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);                    \
    else hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_X6, E_, 4, true>), gridx, dim3(256), ldsx, st, src,  \
                            src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);                       \
  } while (0)
        bool done = true;
        if constexpr (BWD) {
          if (epi_kind == VC_EPI_BWD) VC_LX(true, VC_EPI_BWD);
          else if (epi_kind == VC_EPI_NONE) VC_LX(true, VC_EPI_NONE);
          else done = false;
        } else {
          if (epi_kind == VC_EPI_STATS) VC_LX(false, VC_EPI_STATS);
          else if (epi_kind == VC_EPI_AFFINE) hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, false, 1, VC_OPERAND_X6, VC_EPI_AFFINE, 4, true>),
                                                                 gridx, dim3(256), ldsx, st, src, src_centre, n_src, tbl, wpk, out, rep, order,
                                                                 n_out, kv, centre, mirror, epi);
          else if (epi_kind == VC_EPI_NONE) VC_LX(false, VC_EPI_NONE);
          else done = false;
        }
#undef VC_LX
#undef VC_LX_DXS
        if (done) {
          VC_CHECK_LAUNCH("gather_gemm_v2_kernel<split bf16>");
          return VC_OK;
        }
      }
    }
    // finer blocks fill the last round of the grid better and raise occupancy; coarser blocks reuse W_k more
    // rows per block = 64 * rt.  Measured (tools/kbench.py): rt = 1 wins or ties everywhere -- the kernel is bound by
    // L2 latency / occupancy, not by the W_k re-staging traffic: a variant looping 2/4/8 row tiles per staged W_k (W
    // traffic and barriers / RT) was 5-60 % SLOWER because of its lower occupancy, and was removed.
    // 16-bit operands: only where both channel counts are >= 16 (the 4/8-channel layers are bandwidth-bound and stay fp32)
    const bool half_ops = (ot != VC_OPERAND_F32) && CK >= 16 && CN >= 16;
    const int rt = (g_conv_rt == 2 && !half_ops && epi_kind == VC_EPI_NONE) ? 2 : 1;
    if constexpr (CK >= 16 && CN >= 16) {
      if (conv_block_waves(CK, CN, BWD, n_out, order != nullptr) == 8 && !half_ops && rt == 1 && epi_kind != VC_EPI_AFFINE) {
        // 8-wave blocks (128 rows): see the kernel's NW parameter
        size_t lds8 = (size_t)2 * NCH * NT * 64 * V * sizeof(float) + (size_t)(kv + 1) * 128 * sizeof(int) + 16;
        const dim3 grid8((unsigned)cdiv(n_out, 128));
        fin_attach(epi, epi_kind, grid8.x, 8, CN, lds8);
#ifdef VC_EXPERIMENTS
#define VC_DXS8(B_, E_)                                                                                                        \
  if (wpk && dxs) {                                                                                                            \
    hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_F32, E_, 8, true, true>), grid8, dim3(512), lds8, st, src, \
                       src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);                         \
    break;                                                                                                                     \
  }
#define VC_DXS4(B_, E_)                                                                                                        \
  if (wpk && dxs) {                                                                                                            \
    hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_F32, E_, 4, true, true>), grid, dim3(256), lds, st, VC_ARGS_PK); \
    break;                                                                                                                     \
  }
#else
#define VC_DXS8(B_, E_) do { } while (0)
#define VC_DXS4(B_, E_) do { } while (0)
#endif
#define VC_L8(B_, E_)                                                                                                          \
  do {                                                                                                                         \
    if constexpr (CN % 16 == 0) {                                                                                              \
      VC_DXS8(B_, E_);                                                                                                         \
      if (wpk) {                                                                                                               \
        hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_F32, E_, 8, true>), grid8, dim3(512), lds8, st, src, \
                           src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi);                     \
        break;                                                                                                                 \
      }                                                                                                                        \
    }                                                                                                                          \
    hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_F32, E_, 8, false>), grid8, dim3(512), lds8, st, src,  \
                       src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, epi);                           \
  } while (0)
        if constexpr (!BWD) {
          if (epi_kind == VC_EPI_STATS) {
            VC_L8(false, VC_EPI_STATS);
            VC_CHECK_LAUNCH("gather_gemm_v2_kernel<stats, 8 waves>");
            return VC_OK;
          }
        }
        if constexpr (BWD) {
          if (epi_kind == VC_EPI_BWD) {
            VC_L8(true, VC_EPI_BWD);
            VC_CHECK_LAUNCH("gather_gemm_v2_kernel<bwd epilogue, 8 waves>");
            return VC_OK;
          }
        }
        VC_L8(BWD, VC_EPI_NONE);
#undef VC_L8
        VC_CHECK_LAUNCH("gather_gemm_v2_kernel<8 waves>");
        return VC_OK;
      }
    }
    size_t lds = (size_t)2 * NCH * NT * 64 * V * (half_ops ? 2 : sizeof(float)) +
                 (size_t)(kv + 1) * 64 * rt * sizeof(int) + 16;
    const dim3 grid((unsigned)cdiv(n_out, (int64_t)64 * rt));
    if (!half_ops && rt == 1) fin_attach(epi, epi_kind, grid.x, 4, CN, lds);
#define VC_ARGS src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, epi
#define VC_ARGS_PK src, src_centre, n_src, tbl, wpk, out, rep, order, n_out, kv, centre, mirror, epi
    // fp32, one tile per wave, 4-wave blocks: the fragment-ordered weight image when there is one
#define VC_L2(B_, E_)                                                                                                          \
  do {                                                                                                                         \
    if constexpr (CK % 16 == 0 && CN % 16 == 0) {                                                                              \
      VC_DXS4(B_, E_);                                                                                                         \
      if (wpk) {                                                                                                               \
        hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_F32, E_, 4, true>), grid, dim3(256), lds, st, VC_ARGS_PK); \
        break;                                                                                                                 \
      }                                                                                                                        \
    }                                                                                                                          \
    hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, B_, 1, VC_OPERAND_F32, E_, 4, false>), grid, dim3(256), lds, st, VC_ARGS); \
  } while (0)
    if constexpr (BWD) {
      if (epi_kind == VC_EPI_BWD) {
        if (half_ops) { set_error("gather-GEMM: epilogues are implemented for fp32 operands only"); return VC_EINVAL; }
        VC_L2(true, VC_EPI_BWD);
        VC_CHECK_LAUNCH("gather_gemm_v2_kernel<bwd epilogue>");
        return VC_OK;
      }
    }
    if constexpr (!BWD) {
      if (epi_kind != VC_EPI_NONE) {
        if (half_ops) { set_error("gather-GEMM: epilogues are implemented for fp32 operands only"); return VC_EINVAL; }
        if (epi_kind == VC_EPI_STATS) VC_L2(false, VC_EPI_STATS);
        else VC_L2(false, VC_EPI_AFFINE);
        VC_CHECK_LAUNCH("gather_gemm_v2_kernel<epilogue>");
        return VC_OK;
      }
    }
    if constexpr (CK >= 16 && CN >= 16) {
      if (half_ops && ot == VC_OPERAND_F16) {
        hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, BWD, 1, VC_OPERAND_F16, VC_EPI_NONE>), grid, dim3(256), lds, st, VC_ARGS);
        VC_CHECK_LAUNCH("gather_gemm_v2_kernel<f16>");
        return VC_OK;
      }
      if (half_ops) {
        hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, BWD, 1, VC_OPERAND_BF16, VC_EPI_NONE>), grid, dim3(256), lds, st, VC_ARGS);
        VC_CHECK_LAUNCH("gather_gemm_v2_kernel<bf16>");
        return VC_OK;
      }
    }
    if (rt == 2) hipLaunchKernelGGL((gather_gemm_v2_kernel<CK, CN, BWD, 2, VC_OPERAND_F32, VC_EPI_NONE>), grid, dim3(256), lds, st, VC_ARGS);
    else VC_L2(BWD, VC_EPI_NONE);
#undef VC_L2
#undef VC_ARGS_PK
#undef VC_ARGS
    VC_CHECK_LAUNCH("gather_gemm_v2_kernel");
    return VC_OK;
  }
  if (epi_kind != VC_EPI_NONE) {
    set_error("gather-GEMM: epilogue requested on the fallback kernel (query vc_conv_epilogue_supported first)");
    return VC_EINVAL;
  }
  hipLaunchKernelGGL((gather_gemm_kernel<CK, CN, BWD, kRT>), dim3((unsigned)cdiv(n_out, rows_per_block)), dim3(256), 0,
                     st, src, src_centre, tbl, w, out, rep, n_out, kv, centre, mirror);
  VC_CHECK_LAUNCH("gather_gemm_kernel");
  return VC_OK;
}

template <int CK, bool BWD>
static int dispatch_cn(int cn, const float* src, const float* src_centre, int64_t n_src, const int32_t* tbl, const float* w, float* out,
                       const int32_t* rep, const int32_t* order, int64_t n_out, int kv, int centre, int mirror, int ot,
                       int epi_kind, const ConvEpilogue& epi, int flags, hipStream_t st) {
  switch (cn) {
    case 4: return launch_gg<CK, 4, BWD>(src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 8: return launch_gg<CK, 8, BWD>(src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 16: return launch_gg<CK, 16, BWD>(src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 32: return launch_gg<CK, 32, BWD>(src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 64: return launch_gg<CK, 64, BWD>(src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
  }
  set_error("gather-GEMM: unsupported output channel count %d (supported: 4,8,16,32,64)", cn);
  return VC_EINVAL;
}

template <bool BWD>
static int dispatch_ck(int ck, int cn, const float* src, const float* src_centre, int64_t n_src, const int32_t* tbl, const float* w,
                       float* out, const int32_t* rep, const int32_t* order, int64_t n_out, int kv, int centre, int mirror, int ot,
                       int epi_kind, const ConvEpilogue& epi, int flags, hipStream_t st) {
  switch (ck) {
    case 4: return dispatch_cn<4, BWD>(cn, src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 8: return dispatch_cn<8, BWD>(cn, src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 16: return dispatch_cn<16, BWD>(cn, src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 32: return dispatch_cn<32, BWD>(cn, src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
    case 64: return dispatch_cn<64, BWD>(cn, src, src_centre, n_src, tbl, w, out, rep, order, n_out, kv, centre, mirror, ot, epi_kind, epi, flags, st);
  }
  set_error("gather-GEMM: unsupported source channel count %d (supported: 4,8,16,32,64)", ck);
  return VC_EINVAL;
}

int g_bw_rows_per_split = 1024;  // weight gradient: target rows per block (vc_debug_set bw_rows_per_split); more rows = fewer, longer blocks and fewer partial sums
int g_bw_legacy_order = 0;      // debug: 1 = offset-major block order of the weight-gradient kernel
int g_bw_split = 1;             // vc_debug_set bw_split: 1 (default) = weight-gradient products on the bf16 matrix cores as six split terms (bw_group32_x6); 0 = v_mfma_f32_16x16x4_f32
int g_bw_wide = 1;              // vc_debug_set bw_wide: 16-byte operand gathers + DPP exchange in the split weight gradient at 32 / 16 channels (0: A/B)
int g_bw_small = 1;             // vc_debug_set bw_small: 0 = bwd_weight_kernel for every shape; 1 = bwd_weight_small_kernel where it measured faster; 3 = wherever it applies
int g_bw_variant = 1;           // vc_debug_set bw_variant: 1 = bwd_weight_kernel, 2 = bwd_weight_v2_kernel (dy window in LDS) where it applies
extern int g_pass_dw_main_tail; // pass.hip
extern int g_pass_bwd_epilogue;
extern int g_pass_pack_all;
extern int g_pass_fork_ext_event; // pass.hip
extern int g_bn_fused_partial;  // bn_kernels.hip
extern int g_pass_defer_dw_reduce, g_pass_dw_flush_mb;  // pass.hip
extern int g_plan_subm_bitmap, g_plan_image_2d, g_plan_parity_order, g_plan_params_pad, g_plan_reprepare, g_plan_uv_mode, g_plan_uv_poison, g_plan_uv_lds, g_plan_uv_pad;   // plan.hip
extern long long g_plan_uv_dbg;   // plan.hip
int64_t a17_mismatch_read();   // index_kernels.hip
extern int g_group_plan_radix, g_group_plan_onesweep, g_group_plan_multi_onesweep, g_plan_group_multi;   // group_kernels.hip
extern int g_sp_mark_variant;    // index_kernels.hip
static constexpr int kMaxSplit = 256;
static constexpr size_t kMaxPartialBytes = 24u << 20;

static inline int bw_max_split(int kv, int cin, int cout) {
  size_t per = (size_t)kv * cin * cout * sizeof(float);
  int m = (int)(kMaxPartialBytes / per);
  if (m < 16) m = 16;
  if (m > kMaxSplit) m = kMaxSplit;
  return m;
}

static inline void bw_split(int64_t n_out, int kv, int cin, int cout, int& nsplit, int64_t& rows_per_block) {
  int64_t want = cdiv(n_out, g_bw_rows_per_split);
  const int cap = bw_max_split(kv, cin, cout);
  if (want < 1) want = 1;
  if (want > cap) want = cap;
  rows_per_block = cdiv(cdiv(n_out, want), 256) * 256;
  if (rows_per_block < 256) rows_per_block = 256;
  nsplit = (int)cdiv(n_out, rows_per_block);
  if (nsplit < 1) nsplit = 1;
}

template <int CI, int CO>
static int launch_bw(const float* x, const float* dy, const int32_t* tbl, int64_t n_out, int kv, float* dweight,
                     float* partial, int ot, hipStream_t st, const int32_t* rep = nullptr, int centre = -1,
                     const float* dy_grp = nullptr) {
  int nsplit;
  int64_t rpb;
  bw_split(n_out, kv, CI, CO, nsplit, rpb);
  g_conv_launch_seq.fetch_add(1, std::memory_order_relaxed);
#ifdef VC_EXPERIMENTS
  if constexpr (CI % 16 == 0 && CO % 16 == 0) {
    if (g_bw_variant == 2 && ot == VC_OPERAND_F32 && rep == nullptr) {   // v2: dy window in LDS, offsets split over the waves
      constexpr int NOFF = ((CI / 16) * (CO / 16) >= 16) ? 1 : 2;
      const int ngroups = (kv + 4 * NOFF - 1) / (4 * NOFF);
      hipLaunchKernelGGL((bwd_weight_v2_kernel<CI, CO, NOFF>), dim3((unsigned)(cdiv(nsplit, 8) * 8 * ngroups)), dim3(256), 0, st, x, dy,
                         tbl, n_out, kv, rpb, nsplit, partial);
      VC_CHECK_LAUNCH("bwd_weight_v2_kernel");
      const int total2 = kv * CI * CO;
      if (g_bw_defer != nullptr && g_bw_defer->n < kMaxBwDefer) {
        BwReduceArgs& A = *g_bw_defer;
        const int b0 = A.n ? A.d[A.n - 1].block0 + (int)cdiv((int64_t)A.d[A.n - 1].kv * A.d[A.n - 1].ci * A.d[A.n - 1].co, 64) : 0;
        A.d[A.n++] = BwReduceDesc{partial, dweight, nsplit, kv, CI, CO, b0};
        return VC_OK;
      }
      hipLaunchKernelGGL(bwd_weight_reduce_kernel, dim3((unsigned)cdiv(total2, 64)), dim3(256), 0, st, partial, nsplit, kv, CI, CO,
                         dweight);
      VC_CHECK_LAUNCH("bwd_weight_reduce_kernel");
      return VC_OK;
    }
  }
#endif
  const unsigned nblocks = g_bw_legacy_order ? (unsigned)(nsplit * kv) : (unsigned)(cdiv(nsplit, 8) * 8 * kv);
#define VC_ARGS x, dy, tbl, n_out, kv, rpb, nsplit, g_bw_legacy_order, partial, rep, centre, dy_grp
  bool launched = false;
  if constexpr (CI <= 32 && CO <= 32 && CI * CO <= 512) {   // small channels: 16 pairs per v_mfma_f32_4x4x1_16B, 16-byte operand loads
    // measured (profiles/r04_dw_small_channels.md): wins at 8 x 8 (25.6 -> 19.2-20.4 us) and on the 16 x 16 duplicate-pixel tables
    // (44.9 -> 39.2), loses from 16 x 16 SubM upward (71 -> 80, 32 x 16: 90 -> 138, the sparse 16 x 32 strided table 38 -> 86): the
    // 4x4x1 form runs below the fp32 matrix-pipe rate of the 16x16x4 form here.  Default: C_in C_out <= 64 only; 3 = every served shape.
    const bool f32_operands = ot == VC_OPERAND_F32 || CI < 16 || CO < 16;   // reduced operands apply from 16 channels up only
    if (f32_operands && (g_bw_small == 3 || (g_bw_small == 1 && CI * CO <= 64))) {
      hipLaunchKernelGGL((bwd_weight_small_kernel<CI, CO>), dim3(nblocks), dim3(256), 0, st, VC_ARGS);
      launched = true;
    }
  }
  if constexpr (CI >= 16 && CO >= 16) {  // 16-bit operands only where both channel counts are >= 16 (as in the gather-GEMM)
    if (launched) {
    } else if (ot == VC_OPERAND_F32 && g_bw_split) {   // fp32 products as six bf16 terms (vc_debug_set bw_split)
      constexpr bool HAS_WIDE = CI == 32 || (CO == 32 && CI != 64);   // 16-byte operand gathers (vc_debug_set bw_wide, default 1)
      if (HAS_WIDE && g_bw_wide) hipLaunchKernelGGL((bwd_weight_kernel<CI, CO, VC_OPERAND_X6, HAS_WIDE>), dim3(nblocks), dim3(256), 0, st, VC_ARGS);
      else hipLaunchKernelGGL((bwd_weight_kernel<CI, CO, VC_OPERAND_X6>), dim3(nblocks), dim3(256), 0, st, VC_ARGS);
      launched = true;
    } else if (ot == VC_OPERAND_F16) {
      hipLaunchKernelGGL((bwd_weight_kernel<CI, CO, VC_OPERAND_F16>), dim3(nblocks), dim3(256), 0, st, VC_ARGS);
      launched = true;
    } else if (ot == VC_OPERAND_BF16) {
      hipLaunchKernelGGL((bwd_weight_kernel<CI, CO, VC_OPERAND_BF16>), dim3(nblocks), dim3(256), 0, st, VC_ARGS);
      launched = true;
    }
  }
  if (!launched) hipLaunchKernelGGL((bwd_weight_kernel<CI, CO, VC_OPERAND_F32>), dim3(nblocks), dim3(256), 0, st, VC_ARGS);
#undef VC_ARGS
  VC_CHECK_LAUNCH("bwd_weight_kernel");
  const int total = kv * CI * CO;
  if (g_bw_defer != nullptr && g_bw_defer->n < kMaxBwDefer) {   // reduced later, together with the sweep's other weight gradients
    BwReduceArgs& A = *g_bw_defer;
    const int b0 = A.n ? A.d[A.n - 1].block0 + (int)cdiv((int64_t)A.d[A.n - 1].kv * A.d[A.n - 1].ci * A.d[A.n - 1].co, 64) : 0;
    A.d[A.n++] = BwReduceDesc{partial, dweight, nsplit, kv, CI, CO, b0};
    return VC_OK;
  }
  hipLaunchKernelGGL(bwd_weight_reduce_kernel, dim3((unsigned)cdiv(total, 64)), dim3(256), 0, st, partial, nsplit, kv,
                     CI, CO, dweight);
  VC_CHECK_LAUNCH("bwd_weight_reduce_kernel");
  return VC_OK;
}

// feature pass <-> weight gradient: defer the split-N reductions of the calls made between begin and flush (same host thread)
static thread_local BwReduceArgs g_bw_defer_store;
void bw_defer_begin() { g_bw_defer_store.n = 0; }
void bw_defer_enable(bool on) { g_bw_defer = on ? &g_bw_defer_store : nullptr; }
int bw_defer_flush(hipStream_t st) {
  g_bw_defer = nullptr;
  BwReduceArgs& A = g_bw_defer_store;
  if (A.n == 0) return VC_OK;
  const BwReduceDesc& L = A.d[A.n - 1];
  const unsigned blocks = (unsigned)(L.block0 + cdiv((int64_t)L.kv * L.ci * L.co, 64));
  hipLaunchKernelGGL(bwd_weight_reduce_multi_kernel, dim3(blocks), dim3(256), 0, st, A);
  A.n = 0;
  VC_CHECK_LAUNCH("bwd_weight_reduce_multi_kernel");
  return VC_OK;
}

template <int CI>
static int dispatch_bw_co(int co, const float* x, const float* dy, const int32_t* tbl, int64_t n_out, int kv,
                          float* dweight, float* partial, int ot, hipStream_t st, const int32_t* rep = nullptr, int centre = -1,
                          const float* dy_grp = nullptr) {
  switch (co) {
    case 4: return launch_bw<CI, 4>(x, dy, tbl, n_out, kv, dweight, partial, ot, st, rep, centre, dy_grp);
    case 8: return launch_bw<CI, 8>(x, dy, tbl, n_out, kv, dweight, partial, ot, st, rep, centre, dy_grp);
    case 16: return launch_bw<CI, 16>(x, dy, tbl, n_out, kv, dweight, partial, ot, st, rep, centre, dy_grp);
    case 32: return launch_bw<CI, 32>(x, dy, tbl, n_out, kv, dweight, partial, ot, st, rep, centre, dy_grp);
    case 64: return launch_bw<CI, 64>(x, dy, tbl, n_out, kv, dweight, partial, ot, st, rep, centre, dy_grp);
  }
  set_error("bwd-weight: unsupported output channel count %d", co);
  return VC_EINVAL;
}

}  // namespace vc

using namespace vc;

extern "C" {

int vc_debug_get(const char* key, int64_t* value) {
  VC_REQUIRE(key && value, "vc_debug_get: null argument");
  if (!strcmp(key, "conv_launch_seq")) { *value = (int64_t)g_conv_launch_seq.load(std::memory_order_relaxed); return VC_OK; }
  if (!strcmp(key, "conv_bn_finish_launches")) { *value = (int64_t)g_fin_launches.load(std::memory_order_relaxed); return VC_OK; }
  if (!strcmp(key, "f32_split")) { *value = g_f32_split; return VC_OK; }
  if (!strcmp(key, "a17_mismatch")) { *value = a17_mismatch_read(); return VC_OK; }
  if (!strcmp(key, "bw_split")) { *value = g_bw_split; return VC_OK; }
  if (!strcmp(key, "experiments")) {   // 1: the library carries the measured-and-rejected variants of csrc/experiments/
#ifdef VC_EXPERIMENTS
    *value = 1;
#else
    *value = 0;
#endif
    return VC_OK;
  }
  set_error("vc_debug_get: unknown key %s", key);
  return VC_EINVAL;
}

// keys that select a measured-and-rejected variant (csrc/experiments/): only a library built with -DVC_EXPERIMENTS has them; the
// product build accepts their "off" value and rejects anything else, so that tools and tests can tell the two builds apart
static int experiment_key(const char* key, int value, int off_value, int* slot) {
#ifdef VC_EXPERIMENTS
  (void)off_value;
  *slot = value;
  return VC_OK;
#else
  (void)slot;
  if (value == off_value) return VC_OK;
  set_error("vc_debug_set %s = %d: an experiment variant; build the library with -DVC_EXPERIMENTS (VIRCONV_HIPCC_EXTRA)", key, value);
  return VC_EINVAL;
#endif
}

int vc_debug_set(const char* key, int value) {
  VC_REQUIRE(key, "vc_debug_set: null key");
  if (!strcmp(key, "conv_variant")) { g_conv_variant = value; return VC_OK; }
  if (!strcmp(key, "conv_rt")) { g_conv_rt = value; return VC_OK; }
  if (!strcmp(key, "f32_split")) { g_f32_split = value; return VC_OK; }
  if (!strcmp(key, "conv_autopack")) { g_conv_autopack = value; return VC_OK; }
  if (!strcmp(key, "conv_v4")) return experiment_key(key, value, 0, &g_conv_v4);
  if (!strcmp(key, "conv_v4_ablate")) return experiment_key(key, value, 0, &g_conv_v4_ablate);
  if (!strcmp(key, "conv_v4_pf")) return experiment_key(key, value, 1, &g_conv_v4_pf);
  if (!strcmp(key, "conv_v5")) return experiment_key(key, value, 0, &g_conv_v5);
  if (!strcmp(key, "conv_dxs")) return experiment_key(key, value, 0, &g_conv_dxs);
  if (!strcmp(key, "conv_pc")) return experiment_key(key, value, 0, &g_conv_pc);
  if (!strcmp(key, "conv_wdma")) return experiment_key(key, value, 0, &g_conv_wdma);
  if (!strcmp(key, "conv_winrows")) return experiment_key(key, value, 32, &g_conv_winrows);
  if (!strcmp(key, "bw_variant")) return experiment_key(key, value, 1, &g_bw_variant);
  if (!strcmp(key, "conv_packed")) { g_conv_use_packed = value; return VC_OK; }
  if (!strcmp(key, "conv_window")) { g_conv_window = value; return VC_OK; }
  if (!strcmp(key, "conv_v3_split")) { g_conv_v3_split = value; return VC_OK; }
  if (!strcmp(key, "conv_nw")) { g_conv_nw = (value == 8 || value == 4) ? value : 0; return VC_OK; }
  if (!strcmp(key, "conv_nw8_below")) { g_conv_nw8_below = value; return VC_OK; }
  if (!strcmp(key, "bw_legacy_order")) { g_bw_legacy_order = value; return VC_OK; }
  if (!strcmp(key, "bw_small")) { g_bw_small = value; return VC_OK; }
  if (!strcmp(key, "bw_split")) { g_bw_split = value; return VC_OK; }
  if (!strcmp(key, "bw_wide")) { g_bw_wide = value; return VC_OK; }
  if (!strcmp(key, "bw_rows_per_split")) { if (value >= 256) g_bw_rows_per_split = value; return VC_OK; }
  if (!strcmp(key, "pass_dw_main_tail")) { g_pass_dw_main_tail = value; return VC_OK; }
  if (!strcmp(key, "pass_bwd_epilogue")) { g_pass_bwd_epilogue = value; return VC_OK; }
  if (!strcmp(key, "pass_pack_all")) { g_pass_pack_all = value; return VC_OK; }
  if (!strcmp(key, "pass_fork_ext_event")) { g_pass_fork_ext_event = value; return VC_OK; }
  if (!strcmp(key, "bn_fused_partial")) { g_bn_fused_partial = value; return VC_OK; }
  if (!strcmp(key, "conv_bn_finish")) { g_conv_bn_finish = value; return VC_OK; }
  if (!strcmp(key, "pass_defer_dw_reduce")) { g_pass_defer_dw_reduce = value; return VC_OK; }
  if (!strcmp(key, "pass_dw_flush_mb")) { g_pass_dw_flush_mb = value; return VC_OK; }
  if (!strcmp(key, "group_plan_onesweep")) { g_group_plan_onesweep = value; return VC_OK; }
  if (!strcmp(key, "sp_mark_variant")) { g_sp_mark_variant = value; return VC_OK; }
  if (!strcmp(key, "plan_subm_bitmap")) { g_plan_subm_bitmap = value; return VC_OK; }
  if (!strcmp(key, "plan_image_2d")) { g_plan_image_2d = value; return VC_OK; }
  if (!strcmp(key, "plan_group_multi")) { g_plan_group_multi = value; return VC_OK; }
  if (!strcmp(key, "group_plan_multi_onesweep")) { g_group_plan_multi_onesweep = value; return VC_OK; }
  if (!strcmp(key, "plan_parity_order")) { g_plan_parity_order = value; return VC_OK; }
  if (!strcmp(key, "plan_params_pad")) { g_plan_params_pad = value < 0 ? 0 : (value + 255) & ~255; return VC_OK; }
  if (!strcmp(key, "plan_reprepare")) { g_plan_reprepare = value; return VC_OK; }
  if (!strcmp(key, "plan_uv_mode")) { g_plan_uv_mode = value; return VC_OK; }
  if (!strcmp(key, "plan_uv_pad")) { g_plan_uv_pad = value != 0; return VC_OK; }
  if (!strcmp(key, "plan_uv_dbg_lo")) { g_plan_uv_dbg = (g_plan_uv_dbg & ~0xFFFFFFFFLL) | (long long)(unsigned)value; return VC_OK; }   // a device address in two halves
  if (!strcmp(key, "plan_uv_dbg_hi")) { g_plan_uv_dbg = (g_plan_uv_dbg & 0xFFFFFFFFLL) | ((long long)(unsigned)value << 32); return VC_OK; }
  if (!strcmp(key, "plan_uv_lds")) { g_plan_uv_lds = value < 0 ? 0 : (value > 160 * 1024 ? 160 * 1024 : value); return VC_OK; }
  if (!strcmp(key, "plan_uv_poison")) { g_plan_uv_poison = value; return VC_OK; }
  if (!strcmp(key, "plan_radix_sort")) return experiment_key(key, value, 0, &g_group_plan_radix);
#ifdef VC_EXPERIMENTS
  if (!strcmp(key, "conv_pc_ablate"))
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pc_ablate), &value, sizeof(int)) == hipSuccess ? VC_OK : VC_EHIP;
#endif
  if (!strcmp(key, "xcd_swizzle_off")) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_swizzle_off), &value, sizeof(int)) == hipSuccess ? VC_OK : VC_EHIP;
  }
  set_error("vc_debug_set: unknown key %s", key);
  return VC_EINVAL;
}

int vc_trace_begin(int direction, int ck, int cn, int max_records, int64_t* dev_pairs) {
  VC_REQUIRE(((direction == -1) || ((direction >= 0 && direction <= 2) && ck >= 1 && cn >= 1) || (direction == 3 && ck >= 0)) &&
                 max_records >= 1 && dev_pairs,
             "vc_trace_begin: invalid argument");
  TraceState& T = g_trace;
  T.on = false;
  if (T.n_ev < 2 * max_records) {
    hipEvent_t* ev = (hipEvent_t*)realloc(T.ev, sizeof(hipEvent_t) * 2 * (size_t)max_records);
    vc_trace_record* rec = (vc_trace_record*)realloc(T.rec, sizeof(vc_trace_record) * (size_t)max_records);
    if (!ev || !rec) { set_error("vc_trace_begin: out of host memory"); return VC_EINVAL; }
    T.ev = ev;
    T.rec = rec;
    for (; T.n_ev < 2 * max_records; ++T.n_ev) VC_CHECK_HIP(hipEventCreate(&T.ev[T.n_ev]));
  }
  VC_CHECK_HIP(hipDeviceSynchronize());
  VC_CHECK_HIP(hipMemset(dev_pairs, 0, sizeof(int64_t) * (size_t)max_records));
  T.dir = direction; T.ck = ck; T.cn = cn; T.cap = max_records; T.n = 0; T.dev_pairs = dev_pairs;
  g_trace_counted = (direction == -1) ? 128 : 16;
  T.on = true;
  return VC_OK;
}

int vc_trace_end(vc_trace_record* out, int capacity, int* n_records) {
  TraceState& T = g_trace;
  VC_REQUIRE(n_records && (out || capacity == 0), "vc_trace_end: null argument");
  *n_records = 0;
  if (!T.on) return VC_OK;
  T.on = false;
  VC_CHECK_HIP(hipDeviceSynchronize());
  const int n = T.n < capacity ? T.n : capacity;
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    VC_CHECK_HIP(hipEventElapsedTime(&ms, T.ev[2 * i], T.ev[2 * i + 1]));
    out[i] = T.rec[i];
    out[i].ms = ms;
    float t0 = 0.f;
    if (i > 0) VC_CHECK_HIP(hipEventElapsedTime(&t0, T.ev[0], T.ev[2 * i]));   // start of this launch since the first traced one
    out[i].t0_ms = t0;
    if (T.rec[i].pairs >= 0) {
      VC_CHECK_HIP(hipMemcpy(&out[i].pairs, T.dev_pairs + i, sizeof(int64_t), hipMemcpyDeviceToHost));
    } else {   // estimated: pairs-per-row ratio of the counted record of the same kernel shape whose table is closest in size
      // (the same layer of an earlier step: two layers can share a shape, e.g. s3.d3_conv1 and s4.d3_conv1 are both 64 -> 32)
      double ratio = 0.0;
      int64_t best = -1;
      for (int j = 0; j < i; ++j)
        if (T.rec[j].pairs >= 0 && out[j].kv == out[i].kv && out[j].ck == out[i].ck && out[j].cn == out[i].cn &&
            out[j].direction == out[i].direction && out[j].n_out > 0) {
          const int64_t d = out[j].n_out > out[i].n_out ? out[j].n_out - out[i].n_out : out[i].n_out - out[j].n_out;
          if (best < 0 || d < best) {
            best = d;
            ratio = (double)out[j].pairs / (double)out[j].n_out;
          }
        }
      out[i].pairs = (int64_t)(ratio * (double)out[i].n_out + 0.5);
    }
  }
  *n_records = n;
  return VC_OK;
}

size_t vc_conv_packed_weight_floats(int cin, int cout, int kv, int backward) {
  const int ck = backward ? cout : cin, cn = backward ? cin : cout;
  if (cin < 1 || cout < 1 || kv < 1 || kv > 32 || ck % 16 != 0 || cn % 16 != 0) return 0;  // shapes the gather-GEMMs take an image for
  return (size_t)kv * ck * cn;
}

int vc_conv_pack_weights(int n, const float* const* weights, const int* cin, const int* cout, const int* kv, int backward,
                         float* const* packed, void* stream) {
  VC_REQUIRE(n >= 0 && n <= kMaxPack && (n == 0 || (weights && cin && cout && kv && packed)),
             "vc_conv_pack_weights: null argument or more than %d tensors", kMaxPack);
  PackArgs pa;
  int m = 0, biggest = 0;
  for (int i = 0; i < n; ++i) {
    if (packed[i] == nullptr) continue;
    const size_t fl = vc_conv_packed_weight_floats(cin[i], cout[i], kv[i], backward);
    VC_REQUIRE(fl != 0 && weights[i], "vc_conv_pack_weights: tensor %d (%d -> %d, kv %d) takes no packed image", i, cin[i], cout[i], kv[i]);
    if (g_n_packed >= 2 * kMaxPack) continue;  // registry full: this conv simply keeps reading the canonical layout
    pa.d[m++] = PackDesc{weights[i], packed[i], backward ? cout[i] : cin[i], backward ? cin[i] : cout[i], kv[i], backward ? 1 : 0};
    g_packed[g_n_packed++] = PackedEntry{weights[i], packed[i], backward ? 1 : 0};
    biggest = std::max(biggest, (int)fl);
  }
  if (m == 0) return VC_OK;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<int64_t>(cdiv(biggest, 1024), 64), (unsigned)m), dim3(256), 0,
                     (hipStream_t)stream, pa);
  VC_CHECK_LAUNCH("pack_weights_kernel");
  return VC_OK;
}

int vc_conv_clear_packed_weights(void) {
  g_n_packed = 0;
  return VC_OK;
}

int vc_conv_forward(const float* x, int64_t n_in, const int32_t* pair_fwd, int64_t n_out, int kv, const float* weight,
                    int cin, int cout, const int32_t* row_order, int operand_type, int flags, float* y, void* stream) {
  VC_REQUIRE(n_in >= 0 && n_out >= 0 && kv >= 1 && weight, "vc_conv_forward: null/invalid argument");
  if (n_out == 0) return VC_OK;
  VC_REQUIRE(pair_fwd && y && (x || n_in == 0), "vc_conv_forward: null argument");
  VC_REQUIRE(operand_type >= VC_OPERAND_F32 && operand_type <= VC_OPERAND_BF16, "vc_conv_forward: unknown operand_type");
  const int tr = trace_open(0, cin, cout, (hipStream_t)stream);
  const int rc = dispatch_ck<false>(cin, cout, x, nullptr, n_in, pair_fwd, weight, y, nullptr, row_order, n_out, kv, -1, 0,
                                    operand_type, VC_EPI_NONE, ConvEpilogue{}, flags, (hipStream_t)stream);
  if (tr >= 0) trace_close(tr, 0, cin, cout, pair_fwd, kv, n_in, n_out, (hipStream_t)stream);
  return rc;
}

int vc_conv_epilogue_supported(int64_t n_in, int cin, int cout, int kv, int operand_type) {
  (void)cout;
  return (g_conv_variant == 2 && kv <= 32 && n_in * (int64_t)cin * 4 < (1LL << 31) && operand_type == VC_OPERAND_F32) ? 1 : 0;
}

size_t vc_conv_stats_partial_floats(int64_t n_in, int64_t n_out, int cin, int cout, int kv, int flags) {
  (void)flags;
  if (n_out < 0 || cout < 1) return 0;
  // one partial row (sum, sum of squares per channel) per 16-row wave tile: 4 per 64-row block (8 per 128-row block), direct and
  // window kernel alike; the pair-compacted kernel: 8 per 128-row block
  if (conv_use_pc(cin, cout, kv, n_in, n_out, VC_OPERAND_F32)) return (size_t)cdiv(n_out, 128) * 8 * 2 * cout;
  if (conv_block_waves(cin, cout, false, n_out, false) == 8) return (size_t)cdiv(n_out, 128) * 8 * 2 * cout;
  return (size_t)cdiv(n_out, 64) * 4 * 2 * cout;
}

int vc_conv_forward_epilogue(const float* x, int64_t n_in, const int32_t* pair_fwd, int64_t n_out, int kv, const float* weight,
                             int cin, int cout, const int32_t* row_order, int epilogue, int flags, float* stats_partial,
                             const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                             int relu, float* y, void* stream) {
  VC_REQUIRE(n_in >= 0 && n_out >= 0 && kv >= 1 && weight, "vc_conv_forward_epilogue: null/invalid argument");
  VC_REQUIRE(epilogue == VC_EPI_STATS || epilogue == VC_EPI_AFFINE, "vc_conv_forward_epilogue: unknown epilogue %d", epilogue);
  if (n_out == 0) return VC_OK;
  VC_REQUIRE(pair_fwd && y && (x || n_in == 0), "vc_conv_forward_epilogue: null argument");
  VC_REQUIRE(epilogue != VC_EPI_STATS || stats_partial, "vc_conv_forward_epilogue: stats_partial is null");
  VC_REQUIRE(epilogue != VC_EPI_AFFINE || (mean && var), "vc_conv_forward_epilogue: mean/var are null");
  VC_REQUIRE(vc_conv_epilogue_supported(n_in, cin, cout, kv, VC_OPERAND_F32),
             "vc_conv_forward_epilogue: not available for this shape (vc_conv_epilogue_supported)");
  ConvEpilogue e{stats_partial, mean, var, gamma, beta, eps, relu};
  const int tr = trace_open(0, cin, cout, (hipStream_t)stream);
  const int rc = dispatch_ck<false>(cin, cout, x, nullptr, n_in, pair_fwd, weight, y, nullptr, row_order, n_out, kv, -1, 0,
                                    VC_OPERAND_F32, epilogue, e, flags, (hipStream_t)stream);
  if (tr >= 0) trace_close(tr, 0, cin, cout, pair_fwd, kv, n_in, n_out, (hipStream_t)stream);
  return rc;
}

int vc_conv_backward_input(const float* dy, const float* dy_centre, int64_t n_src, const int32_t* tbl, int64_t n_in,
                           int kv, const float* weight, int cin, int cout, int mirror, int centre, const int32_t* rep,
                           const int32_t* row_order, int operand_type, int flags, float* dx, void* stream) {
  VC_REQUIRE(n_src >= 0 && n_in >= 0 && kv >= 1 && weight, "vc_conv_backward_input: null/invalid argument");
  if (n_in == 0) return VC_OK;
  VC_REQUIRE(tbl && dx && (dy || n_src == 0), "vc_conv_backward_input: null argument");
  VC_REQUIRE(centre >= -1 && centre < kv, "vc_conv_backward_input: centre out of range");
  VC_REQUIRE(operand_type >= VC_OPERAND_F32 && operand_type <= VC_OPERAND_BF16,
             "vc_conv_backward_input: unknown operand_type");
  const int tr = trace_open(1, cout, cin, (hipStream_t)stream);
  const int rc = dispatch_ck<true>(cout, cin, dy, dy_centre, n_src, tbl, weight, dx, rep, row_order, n_in, kv, centre,
                                   mirror ? 1 : 0, operand_type, VC_EPI_NONE, ConvEpilogue{}, flags, (hipStream_t)stream);
  if (tr >= 0) trace_close(tr, 1, cout, cin, tbl, kv, n_src, n_in, (hipStream_t)stream);
  return rc;
}

size_t vc_conv_bwd_stats_partial_floats(int64_t n_in, int cin, int cout, int row_ordered) {
  if (n_in < 0 || cin < 1 || cout < 1) return 0;
  // one partial row per 16-row wave tile of the backward-input kernel <CK = cout, CN = cin> that this launch will get
  const int nw = conv_block_waves(cout, cin, true, n_in, row_ordered != 0);
  return (size_t)cdiv(n_in, 16 * nw) * nw * 2 * cin;
}

int vc_conv_backward_input_epilogue(const float* dy, const float* dy_centre, int64_t n_src, const int32_t* tbl, int64_t n_in,
                                    int kv, const float* weight, int cin, int cout, int mirror, int centre, const int32_t* rep,
                                    const int32_t* row_order, int flags, const float* addend, int add_stride, int add_col0,
                                    const float* y_raw, const float* mean, const float* var, const float* gamma,
                                    const float* beta, float eps, int relu, float* stats_partial, float* dx, void* stream) {
  VC_REQUIRE(n_src >= 0 && n_in >= 1 && kv >= 1 && weight && tbl && dx && (dy || n_src == 0),
             "vc_conv_backward_input_epilogue: null/invalid argument");
  VC_REQUIRE(centre >= -1 && centre < kv, "vc_conv_backward_input_epilogue: centre out of range");
  VC_REQUIRE(!addend || (add_stride >= cin && add_col0 >= 0 && add_col0 + cin <= add_stride),
             "vc_conv_backward_input_epilogue: bad addend view");
  VC_REQUIRE(!y_raw || (mean && var && stats_partial), "vc_conv_backward_input_epilogue: statistics requested without mean/var/partial");
  VC_REQUIRE(vc_conv_epilogue_supported(n_src, cout, cin, kv, VC_OPERAND_F32),
             "vc_conv_backward_input_epilogue: not available for this shape (vc_conv_epilogue_supported(n_src, cout, cin, kv))");
  ConvEpilogue e{stats_partial, mean, var, gamma, beta, eps, relu};
  e.y_raw = y_raw;
  e.addend = addend;
  e.add_stride = add_stride;
  e.add_col0 = add_col0;
  const int tr = trace_open(1, cout, cin, (hipStream_t)stream);
  const int rc = dispatch_ck<true>(cout, cin, dy, dy_centre, n_src, tbl, weight, dx, rep, row_order, n_in, kv, centre,
                                   mirror ? 1 : 0, VC_OPERAND_F32, VC_EPI_BWD, e, flags, (hipStream_t)stream);
  if (tr >= 0) trace_close(tr, 1, cout, cin, tbl, kv, n_src, n_in, (hipStream_t)stream);
  return rc;
}

size_t vc_conv_backward_weight_workspace_bytes(int64_t n_out, int kv, int cin, int cout) {
  (void)n_out;
  if (kv < 1 || cin < 1 || cout < 1) return 0;
  return (size_t)bw_max_split(kv, cin, cout) * kv * cin * cout * sizeof(float);
}

static int conv_backward_weight_impl(const float* x, const float* dy, const int32_t* pair_fwd, int64_t n_out, int kv, int cin,
                                     int cout, int operand_type, float* dweight, void* ws, size_t ws_bytes, void* stream,
                                     const int32_t* rep, int centre, const float* dy_grp);

int vc_conv_backward_weight(const float* x, const float* dy, const int32_t* pair_fwd, int64_t n_out, int kv, int cin,
                            int cout, int operand_type, float* dweight, void* ws, size_t ws_bytes, void* stream) {
  return conv_backward_weight_impl(x, dy, pair_fwd, n_out, kv, cin, cout, operand_type, dweight, ws, ws_bytes, stream, nullptr, -1,
                                   nullptr);
}

int vc_conv_backward_weight_dup(const float* x, const float* dy, const float* dy_grp, const int32_t* rep, int centre,
                                const int32_t* pair_fwd, int64_t n_out, int kv, int cin, int cout, int operand_type,
                                float* dweight, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(dy_grp && rep && centre >= 0 && centre < kv, "vc_conv_backward_weight_dup: null/invalid duplicate-pixel arguments");
  return conv_backward_weight_impl(x, dy, pair_fwd, n_out, kv, cin, cout, operand_type, dweight, ws, ws_bytes, stream, rep, centre,
                                   dy_grp);
}

static int conv_backward_weight_impl(const float* x, const float* dy, const int32_t* pair_fwd, int64_t n_out, int kv, int cin,
                                     int cout, int operand_type, float* dweight, void* ws, size_t ws_bytes, void* stream,
                                     const int32_t* rep, int centre, const float* dy_grp) {
  VC_REQUIRE(n_out >= 0 && kv >= 1 && dweight && ws, "vc_conv_backward_weight: null/invalid argument");
  VC_REQUIRE(operand_type >= VC_OPERAND_F32 && operand_type <= VC_OPERAND_BF16,
             "vc_conv_backward_weight: unknown operand_type");
  hipStream_t st = (hipStream_t)stream;
  if (ws_bytes < vc_conv_backward_weight_workspace_bytes(n_out, kv, cin, cout)) {
    set_error("vc_conv_backward_weight: workspace too small");
    return VC_ECAPACITY;
  }
  if (n_out == 0) {
    VC_CHECK_HIP(hipMemsetAsync(dweight, 0, (size_t)kv * cin * cout * 4, st));
    return VC_OK;
  }
  VC_REQUIRE(x && dy && pair_fwd, "vc_conv_backward_weight: null argument");
  float* partial = (float*)ws;
  const int tr = trace_open(2, cin, cout, st);   // direction 2 (this shape) or the trace-everything mode (direction -1)
  int rc;
  switch (cin) {
    case 4: rc = dispatch_bw_co<4>(cout, x, dy, pair_fwd, n_out, kv, dweight, partial, operand_type, st, rep, centre, dy_grp); break;
    case 8: rc = dispatch_bw_co<8>(cout, x, dy, pair_fwd, n_out, kv, dweight, partial, operand_type, st, rep, centre, dy_grp); break;
    case 16: rc = dispatch_bw_co<16>(cout, x, dy, pair_fwd, n_out, kv, dweight, partial, operand_type, st, rep, centre, dy_grp); break;
    case 32: rc = dispatch_bw_co<32>(cout, x, dy, pair_fwd, n_out, kv, dweight, partial, operand_type, st, rep, centre, dy_grp); break;
    case 64: rc = dispatch_bw_co<64>(cout, x, dy, pair_fwd, n_out, kv, dweight, partial, operand_type, st, rep, centre, dy_grp); break;
    default:
      set_error("bwd-weight: unsupported input channel count %d", cin);
      rc = VC_EINVAL;
  }
  if (tr >= 0) trace_close(tr, 2, cin, cout, pair_fwd, kv, n_out, n_out, st);
  return rc;
}

#ifdef VC_EXPERIMENTS   // order-free 64-bit fixed-point group sum of rounds 1-2 (csrc/experiments/group_sum_fixed.inc); not in the header
size_t vc_group_sum_workspace_bytes(int64_t n, int c) {
  if (n < 0 || c < 1) return 0;
  return (size_t)n * c * sizeof(long long) + 64;
}

int vc_group_sum_prepare(void* ws, size_t ws_bytes, int64_t n, int c, void* stream) {
  VC_REQUIRE(n >= 0 && c > 0 && ws, "vc_group_sum_prepare: null/invalid argument");
  if (ws_bytes < vc_group_sum_workspace_bytes(n, c)) { set_error("vc_group_sum_prepare: workspace too small"); return VC_ECAPACITY; }
  VC_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)n * c * sizeof(long long) + 64, (hipStream_t)stream));
  return VC_OK;
}

int vc_group_sum(const float* dy, const int32_t* rep, int64_t n, int c, float* dy_grp, void* ws, size_t ws_bytes,
                 int prepared, void* stream) {
  VC_REQUIRE(n >= 0 && c > 0, "vc_group_sum: invalid argument");
  if (n == 0) return VC_OK;
  VC_REQUIRE(dy && rep && dy_grp && ws, "vc_group_sum: null argument");
  if (ws_bytes < vc_group_sum_workspace_bytes(n, c)) { set_error("vc_group_sum: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  unsigned* absmax = (unsigned*)ws;                 // [0..63] header, then the int64 accumulators
  long long* acc = (long long*)((char*)ws + 64);
  const int64_t total = n * c;
  if (!prepared) {  // prepared: vc_group_sum_prepare zeroed ws and the producer of dy (vc_bn_relu_backward) left max|dy| in it
    VC_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)total * sizeof(long long) + 64, st));
    int64_t nb = cdiv(total, 256 * 16);
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)nb), dim3(256), 0, st, dy, total, absmax);
    VC_CHECK_LAUNCH("absmax_kernel");
  }
  hipLaunchKernelGGL(group_sum_fixed_kernel, dim3((unsigned)cdiv(cdiv(n, kGsRows) * c, 256)), dim3(256), 0, st, dy, rep, n, c,
                     absmax, acc);
  VC_CHECK_LAUNCH("group_sum_fixed_kernel");
  hipLaunchKernelGGL(group_sum_convert_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, acc, total, c, rep, absmax,
                     dy_grp, prepared == 2 ? 1 : 0);
  VC_CHECK_LAUNCH("group_sum_convert_kernel");
  return VC_OK;
}

#endif

}  // extern "C"
