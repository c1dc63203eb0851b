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
  CHECK(hipMalloc(&src, (size_t)n * CK * 4)); CHECK(hipMalloc(&out, (size_t)n * 4 + 65536)); CHECK(hipMalloc(&dtbl, tbl.size() * 4));
  CHECK(hipMalloc(&w, 27 * 64 * 32 * 4));
  CHECK(hipMemset(src, 0, (size_t)n * CK * 4)); CHECK(hipMemset(w, 0, 27 * 64 * 32 * 4));
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
  const int steps = 16;
  const double wb = (double)nb * steps * 8192;
  time([&] { hipLaunchKernelGGL(wload_kernel<0>, dim3(nb), dim3(64), 0, 0, w, 27, steps, out); }, "W fragments, canonical layout", wb);
  time([&] { hipLaunchKernelGGL(wload_kernel<1>, dim3(nb), dim3(64), 0, 0, w, 27, steps, out); }, "W fragments, fragment-ordered image", wb);
  CHECK(hipDeviceSynchronize());
  return 0;
}
