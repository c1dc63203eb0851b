// Micro-benchmark: what does a 16-byte-per-lane row gather cost on gfx950 as a function of the LANE -> (row, chunk) mapping?
//   mode 0  MFMA A-operand mapping: lane l reads row idx[l & 15], 16-B chunk (l >> 4) + 4 * ch   (what gather_gemm_v2/v4 issue:
//           every quad of consecutive lanes touches 4 different rows)
//   mode 1  quad mapping: lane l reads row idx[l >> 2], chunk (l & 3) + 4 * ch                   (a quad reads 64 contiguous bytes)
//   mode 2  quad mapping through the LDS-DMA path (buffer_load_dwordx4 ... lds), then ds_read_b128 in the MFMA layout
// Every wave owns 64 output rows (4 tiles of 16) and walks KOFF neighbour offsets; rows are 64 floats (256 B).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_ubench.hip -o tools/ubench/gather_ubench && tools/ubench/gather_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK 64
#define KOFF 16

__device__ __forceinline__ i32x4 bl(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0); }

template <int MODE>
__global__ void __launch_bounds__(64) gather_kernel(const float* __restrict__ src, int n, const int* __restrict__ tbl,
                                                    float* __restrict__ out) {
  const int lane = threadIdx.x;
  const int row0 = blockIdx.x * 64;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * CK * 4, 0x00020000);
  __shared__ int s_idx[KOFF * 64];
  for (int k = 0; k < KOFF; ++k) s_idx[k * 64 + lane] = (row0 + lane < n) ? tbl[(size_t)k * n + row0 + lane] : -1;
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < KOFF; ++k) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int id; unsigned base;
      if (MODE == 0) { id = s_idx[k * 64 + t * 16 + (lane & 15)]; base = (unsigned)id * 256u + (unsigned)(lane >> 4) * 16u; }
      else           { id = s_idx[k * 64 + t * 16 + (lane >> 2)]; base = (unsigned)id * 256u + (unsigned)(lane & 3) * 16u; }
      i32x4 v[4];
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) v[ch] = bl(rs, base + ch * 64);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        acc[0] += __int_as_float(v[ch].x); acc[1] += __int_as_float(v[ch].y);
        acc[2] += __int_as_float(v[ch].z); acc[3] += __int_as_float(v[ch].w);
      }
    }
  }
  out[(size_t)blockIdx.x * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
}

// mode 2: rows fetched by the LDS-DMA engine with the quad mapping (lane 4r + c fetches 16 bytes of row r: one cache line per quad),
// 1 KB of LDS per instruction in lane order; the MFMA-layout fragment (row l & 15, chunk l >> 4) is then a conflict-free
// ds_read_b128 thanks to a chunk swizzle slot = chunk ^ ((row >> 3) << 1).  Index -1 reads a zero row that lives in a __device__
// array.  `check` != 0: write the gathered sums per (row, 16-B chunk element) for comparison with mode 0.
__device__ float g_zero_row[64];
template <int MODE>
__global__ void __launch_bounds__(64) gather_sum_kernel(const float* __restrict__ src, int n, const int* __restrict__ tbl,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x;
  const int row0 = blockIdx.x * 64;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * CK * 4, 0x00020000);
  __shared__ int s_idx[KOFF * 64];
  __shared__ __attribute__((aligned(16))) float s_a[2][4 * 16 * 16];  // [buffer][ch][row][16 floats]
  for (int k = 0; k < KOFF; ++k) s_idx[k * 64 + lane] = (row0 + lane < n) ? tbl[(size_t)k * n + row0 + lane] : -1;
  __syncthreads();
  const int i = lane & 15, q = lane >> 4;
  float acc[4][4][4];  // [tile][ch][j]: sum over offsets of the fragment element the MFMA lane (i, q) would hold
  for (int t = 0; t < 4; ++t) for (int ch = 0; ch < 4; ++ch) for (int j = 0; j < 4; ++j) acc[t][ch][j] = 0.f;
  if (MODE == 3) {
    // quad-mapped loads (lane 4r + c reads chunk c of row r: one cache line per quad), then the 16-byte values are moved to the
    // MFMA layout (lane r + 16c) in registers with ds_bpermute_b32 (LDS crossbar, no LDS memory): dst lane l pulls from lane
    // 4 * (l & 15) + (l >> 4)
    const int r = lane >> 2, c = lane & 3;
    const int pull = (4 * i + q) * 4;
    for (int k = 0; k < KOFF; ++k)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int id = s_idx[k * 64 + t * 16 + r];
        const unsigned base = (unsigned)id * 256u + (unsigned)c * 16u;
        i32x4 v[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) v[ch] = bl(rs, base + ch * 64);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          acc[t][ch][0] += __int_as_float(__builtin_amdgcn_ds_bpermute(pull, v[ch].x));
          acc[t][ch][1] += __int_as_float(__builtin_amdgcn_ds_bpermute(pull, v[ch].y));
          acc[t][ch][2] += __int_as_float(__builtin_amdgcn_ds_bpermute(pull, v[ch].z));
          acc[t][ch][3] += __int_as_float(__builtin_amdgcn_ds_bpermute(pull, v[ch].w));
        }
      }
  } else if (MODE == 0) {
    for (int k = 0; k < KOFF; ++k)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int id = s_idx[k * 64 + t * 16 + i];
        const unsigned base = (unsigned)id * 256u + (unsigned)q * 16u;
        i32x4 v[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) v[ch] = bl(rs, base + ch * 64);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          acc[t][ch][0] += __int_as_float(v[ch].x); acc[t][ch][1] += __int_as_float(v[ch].y);
          acc[t][ch][2] += __int_as_float(v[ch].z); acc[t][ch][3] += __int_as_float(v[ch].w);
        }
      }
  } else {
    const int r = lane >> 2, c = lane & 3;
    const int cq = c ^ ((r >> 3) << 1);              // the chunk this lane fetches so that slot (r, c) holds chunk cq
    const int rd = (i * 16 + (q ^ ((i >> 3) << 1)) * 4);  // float offset of (row i, chunk q) inside one ch block
    int buf = 0;
    auto issue = [&](int k, int t, int b) {
      const int id = s_idx[k * 64 + t * 16 + r];
      const float* g = id >= 0 ? src + (size_t)id * CK + cq * 4 : g_zero_row + cq * 4;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ch * 16),
                                         (__attribute__((address_space(3))) void*)(&s_a[b][ch * 256]), 16, 0, 0);
    };
    issue(0, 0, 0);
    for (int s = 0; s < KOFF * 4; ++s) {
      const int k = s >> 2, t = s & 3;
      const int sn = (s + 1 < KOFF * 4) ? s + 1 : s;
      issue(sn >> 2, sn & 3, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const float4 v = *reinterpret_cast<const float4*>(&s_a[buf][ch * 256 + rd]);
        acc[t][ch][0] += v.x; acc[t][ch][1] += v.y; acc[t][ch][2] += v.z; acc[t][ch][3] += v.w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      buf ^= 1;
    }
  }
  float* o = out + ((size_t)blockIdx.x * 64 + lane) * 64;
  for (int t = 0; t < 4; ++t) for (int ch = 0; ch < 4; ++ch) for (int j = 0; j < 4; ++j) o[(t * 4 + ch) * 4 + j] = acc[t][ch][j];
}

// weight-fragment style loads: 8 x 16-B per lane per step; mode 0: lane (n = l & 15, q = l >> 4) reads w[n][k][q*4 + 16 ch] from the
// canonical (Cout, KV, Cin) layout (rows 6912 B apart); mode 1: fragment-ordered image (each instruction = 1 KB contiguous)
template <int MODE>
__global__ void __launch_bounds__(64) wload_kernel(const float* __restrict__ w, int kv, int steps, float* __restrict__ out) {
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, kv * 64 * 32 * 4, 0x00020000);
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    const int k = (s + blockIdx.x) % kv;
    i32x4 v[8];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        unsigned off;
        if (MODE == 0) off = (unsigned)((((nt * 16 + (lane & 15)) * kv + k) * 64 + ch * 16 + (lane >> 4) * 4) * 4);
        else off = (unsigned)(((k * 8 + ch * 2 + nt) * 64 + lane) * 16);
        v[ch * 2 + nt] = bl(rs, off);
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += __int_as_float(v[u].x) + __int_as_float(v[u].w);
  }
  out[(size_t)blockIdx.x * 64 + lane] = acc;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 194944;
  const int line = 176;            // rows per x-line; neighbours: dx in {-1,0,1}, dy in +-line, dz in +-(line*40)
  std::vector<int> tbl((size_t)KOFF * n);
  const int deltas[KOFF] = {-1, 0, 1, -line - 1, -line, -line + 1, line - 1, line, line + 1, -line * 40 - 1, -line * 40, -line * 40 + 1,
                            line * 40 - 1, line * 40, line * 40 + 1, 2};
  for (int k = 0; k < KOFF; ++k)
    for (int r = 0; r < n; ++r) {
      long s = (long)r + deltas[k];
      tbl[(size_t)k * n + r] = (s >= 0 && s < n) ? (int)s : -1;
    }
  float *src, *out, *w; int* dtbl;
  CHECK(hipMalloc(&src, (size_t)n * CK * 4)); CHECK(hipMalloc(&out, ((size_t)n + 64) * 64 * 4 * 2)); CHECK(hipMalloc(&dtbl, tbl.size() * 4));
  CHECK(hipMalloc(&w, 27 * 64 * 32 * 4));
  {
    std::vector<float> h((size_t)n * CK);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (float)((x >> 9) & 1023) / 64.f - 8.f; }   // small exactly-summable values
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CHECK(hipMemset(w, 0, 27 * 64 * 32 * 4));
  CHECK(hipMemcpy(dtbl, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int nb = (n + 63) / 64;
  auto time = [&](auto fn, const char* name, double bytes) {
    for (int i = 0; i < 3; ++i) fn();
    CHECK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) fn(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us   %7.2f TB/s L2->CU\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  const double gb = (double)n * KOFF * 256;
  time([&] { hipLaunchKernelGGL(gather_kernel<0>, dim3(nb), dim3(64), 0, 0, src, n, dtbl, out); }, "gather, MFMA lane mapping (i, q)", gb);
  time([&] { hipLaunchKernelGGL(gather_kernel<1>, dim3(nb), dim3(64), 0, 0, src, n, dtbl, out); }, "gather, quad mapping (64 B per quad)", gb);
  float* out2 = out + ((size_t)n + 64) * 64;
  time([&] { hipLaunchKernelGGL(gather_sum_kernel<0>, dim3(nb), dim3(64), 0, 0, src, n, dtbl, out); }, "gather + keep fragments, MFMA mapping", gb);
  time([&] { hipLaunchKernelGGL(gather_sum_kernel<3>, dim3(nb), dim3(64), 0, 0, src, n, dtbl, out2); }, "gather quad-mapped + ds_bpermute transpose", gb);
  {
    std::vector<float> a((size_t)nb * 64 * 64), b(a.size());
    CHECK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t e = 0; e < a.size(); ++e) bad += a[e] != b[e];
    printf("quad + bpermute gather vs direct gather: %zu of %zu fragment sums differ\n", bad, a.size());
  }
  time([&] { hipLaunchKernelGGL(gather_sum_kernel<2>, dim3(nb), dim3(64), 0, 0, src, n, dtbl, out2); }, "gather via LDS-DMA (quad mapping) + ds_read_b128", gb);
  {
    std::vector<float> a((size_t)nb * 64 * 64), b(a.size());
    CHECK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t e = 0; e < a.size(); ++e) bad += a[e] != b[e];
    printf("LDS-DMA gather vs direct gather: %zu of %zu fragment sums differ\n", bad, a.size());
  }
  const int steps = 16;
  const double wb = (double)nb * steps * 8192;
  time([&] { hipLaunchKernelGGL(wload_kernel<0>, dim3(nb), dim3(64), 0, 0, w, 27, steps, out); }, "W fragments, canonical layout", wb);
  time([&] { hipLaunchKernelGGL(wload_kernel<1>, dim3(nb), dim3(64), 0, 0, w, 27, steps, out); }, "W fragments, fragment-ordered image", wb);
  CHECK(hipDeviceSynchronize());
  return 0;
}
