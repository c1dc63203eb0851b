// K11: training/eval BatchNorm1d(+ReLU) over the N active rows of a sparse tensor, HBM-bound elementwise/reduction
// work (replaces nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU after every conv, spconv_backbone.py:101-105,160).
// Reductions are two-stage with a fixed summation order (bit-stable run to run); accumulation across threads in fp64.
#include "common.h"

namespace vc {

static constexpr int kMaxBnBlocks = 1024;

// Per-block partial sums of (a, b) per channel, where for STATS: a = x, b = x*x;
// for BWD: a = dyr (relu-masked dy), b = dyr * xhat.
template <bool BWD>
__global__ void __launch_bounds__(256) bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        int dy_stride, int dy_col0, int64_t n, int c,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int relu, int64_t rows_per_block,
                                                        double* __restrict__ partial, unsigned* __restrict__ zero_word) {
  __shared__ double lds[256 * 8];
  // the max|dx| word bn_bwd_dx_pow2_kernel (the next launch on this stream) accumulates into with atomicMax: cleared here
  if (zero_word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;
  const int c4 = c >> 2;
  const int R = 256 / c4;           // row-threads per block (power of two)
  const int cq = threadIdx.x % c4;  // which float4 of the row
  const int rt = threadIdx.x / c4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + rows_per_block, n);
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  float mu[4], istd[4], g[4], bt[4];
  if (BWD) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = cq * 4 + j;
      mu[j] = mean[ch];
      istd[j] = 1.0f / sqrtf(var[ch] + eps);
      g[j] = gamma ? gamma[ch] : 1.f;
      bt[j] = beta ? beta[ch] : 0.f;
    }
  }
  if (rt < R) {
    // rows are consumed in the same order as a plain loop (bit-identical sums); the 4 row loads of a trip are issued before
    // the first is used, so each thread keeps 4-8 16-byte loads in flight instead of one
    auto consume = [&](const float4& xv, const float4& dv) {
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sa[j] += xs[j]; sb[j] += xs[j] * xs[j]; }
      } else {
        const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xs[j] - mu[j]) * istd[j];
          float d = ds[j];
          if (relu && !(xh * g[j] + bt[j] > 0.f)) d = 0.f;
          sa[j] += d;
          sb[j] += d * xh;
        }
      }
    };
    int64_t r = r0 + rt;
    for (; r + 3 * (int64_t)R < r1; r += 4 * (int64_t)R) {
      float4 xv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *reinterpret_cast<const float4*>(x + (r + u * (int64_t)R) * c + cq * 4);
        dv[u] = BWD ? *reinterpret_cast<const float4*>(dy + (r + u * (int64_t)R) * dy_stride + dy_col0 + cq * 4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) consume(xv[u], dv[u]);
    }
    for (; r < r1; r += R) {
      const float4 xv = *reinterpret_cast<const float4*>(x + r * c + cq * 4);
      const float4 dv = BWD ? *reinterpret_cast<const float4*>(dy + r * dy_stride + dy_col0 + cq * 4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
      consume(xv, dv);
    }
  }
  double* my = lds + threadIdx.x * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) { my[j] = (double)sa[j]; my[4 + j] = (double)sb[j]; }
  __syncthreads();
  for (int s = R >> 1; s >= 1; s >>= 1) {
    if (rt < s) {
      const double* o = lds + (threadIdx.x + s * c4) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) my[j] += o[j];
    }
    __syncthreads();
  }
  if (rt == 0) {
    double* dst = partial + (int64_t)blockIdx.x * 2 * c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dst[cq * 4 + j] = my[j];
      dst[c + cq * 4 + j] = my[4 + j];
    }
  }
}

// One wave per channel: lane l sums partials l, l+64, ... (fixed order), then a fixed-order butterfly; 4 channels/block.
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
}

__global__ void __launch_bounds__(256) bn_stats_finalize_kernel(const double* __restrict__ partial, int nb, int64_t n,
                                                                int c, float* __restrict__ mean,
                                                                float* __restrict__ var,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var,
                                                                long long* __restrict__ num_batches_tracked,
                                                                float momentum) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;
  if (num_batches_tracked && ch == 0 && lane == 0) *num_batches_tracked += 1;
  double s = 0.0, ss = 0.0;
  {  // same addition order as a plain loop; four trips' loads are issued together (the kernel was a chain of dependent loads)
    int b = lane;
    for (; b + 192 < nb; b += 256) {
      double a[4], q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[(int64_t)(b + 64 * u) * 2 * c + ch];
        q[u] = partial[(int64_t)(b + 64 * u) * 2 * c + c + ch];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s += a[u]; ss += q[u]; }
    }
    for (; b < nb; b += 64) {
      s += partial[(int64_t)b * 2 * c + ch];
      ss += partial[(int64_t)b * 2 * c + c + ch];
    }
  }
  wave_sum2(s, ss);
  if (lane != 0) return;
  const double m = s / (double)n;
  double v = ss / (double)n - m * m;
  if (v < 0.0) v = 0.0;
  mean[ch] = (float)m;
  var[ch] = (float)v;
  if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
  if (running_var) {
    const double unb = (n > 1) ? v * (double)n / (double)(n - 1) : v;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
  }
}

// statistics from the conv epilogue's per-block fp32 partials [nb][2][c] (sum, sum of squares): same outputs as above
__global__ void __launch_bounds__(256) bn_stats_from_partial_kernel(const float* __restrict__ partial, int64_t nb, int64_t n,
                                                                    int c, float* __restrict__ mean,
                                                                    float* __restrict__ var,
                                                                    float* __restrict__ running_mean,
                                                                    float* __restrict__ running_var,
                                                                    long long* __restrict__ num_batches_tracked,
                                                                    float momentum) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;
  if (num_batches_tracked && ch == 0 && lane == 0) *num_batches_tracked += 1;
  double s = 0.0, ss = 0.0;
  for (int64_t b = lane; b < nb; b += 64) {
    s += (double)partial[(b * 2 + 0) * c + ch];
    ss += (double)partial[(b * 2 + 1) * c + ch];
  }
  wave_sum2(s, ss);
  if (lane != 0) return;
  const double m = s / (double)n;
  double v = ss / (double)n - m * m;
  if (v < 0.0) v = 0.0;
  mean[ch] = (float)m;
  var[ch] = (float)v;
  if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
  if (running_var) {
    const double unb = (n > 1) ? v * (double)n / (double)(n - 1) : v;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
  }
}

// First stage for MANY conv-epilogue partial rows (one per 16-row wave tile on the LDS-window kernel: N/16 rows): block g adds
// up a contiguous slab of rows with fully coalesced reads (a row = 2c consecutive floats; thread = (row sub-index, column)),
// fp64 accumulation, fixed order, and writes ONE fp64 partial row in the [g][2][c] layout of bn_stats_finalize_kernel.
__global__ void __launch_bounds__(256) bn_partial_reduce_kernel(const float* __restrict__ partial, int64_t nb,
                                                                int64_t rows_per_block, int c,
                                                                double* __restrict__ dpartial) {
  __shared__ double lds[256];
  const int c2 = 2 * c;
  const int RS = 256 / c2;  // row sub-threads (c2 <= 256, power of two)
  const int col = threadIdx.x % c2, rs = threadIdx.x / c2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, nb);
  double acc = 0.0;
  if (rs < RS) {
    int64_t r = r0 + rs;
    for (; r + 3 * (int64_t)RS < r1; r += 4 * (int64_t)RS) {  // four loads in flight, consumed in row order
      const float v0 = partial[r * c2 + col], v1 = partial[(r + RS) * c2 + col];
      const float v2 = partial[(r + 2 * RS) * c2 + col], v3 = partial[(r + 3 * RS) * c2 + col];
      acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
    }
    for (; r < r1; r += RS) acc += (double)partial[r * c2 + col];
  }
  lds[threadIdx.x] = acc;
  __syncthreads();
  if (rs == 0) {
    double t = lds[col];
    for (int u = 1; u < RS; ++u) t += lds[u * c2 + col];
    dpartial[(int64_t)blockIdx.x * c2 + col] = t;
  }
}

// ONE launch from the conv epilogue's per-wave fp32 partial rows [nb][2][c] to the finished per-channel numbers (round 3; it
// replaces bn_partial_reduce_kernel + bn_stats_finalize_kernel / bn_bwd_finalize_kernel, i.e. 40 launches and 40 kernel
// boundaries per train step).  Block = 4 channels (one 16-byte column of the sums and the matching one of the second
// moments) x 1024 threads; thread t adds up rows t, t + 1024, ... in that order in fp64 with up to 8 rows (16 loads of 16
// bytes) in flight, then a fixed-order butterfly per wave and a fixed-order sum over the 16 waves: run-to-run bit-stable.
//   BWD = false: (sum x, sum x^2)      -> mean, biased var, running statistics, num_batches_tracked
//   BWD = true : (sum d, sum d * xhat) -> dbeta, dgamma, sums[2][c] for the dx kernel; clears the max|dx| word
static constexpr int kBnFusedThreads = 1024;

template <bool BWD>
__global__ void __launch_bounds__(kBnFusedThreads) bn_partial_fused_kernel(const float* __restrict__ partial, int64_t nb, int64_t n,
                                                                         int c, float* __restrict__ o0, float* __restrict__ o1,
                                                                         float* __restrict__ r0, float* __restrict__ r1,
                                                                         long long* __restrict__ num_batches_tracked,
                                                                         float momentum, unsigned* __restrict__ zero_word) {
  __shared__ double red[kBnFusedThreads / 64][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = blockIdx.x;
  if (BWD && zero_word != nullptr && cq == 0 && tid == 0) *zero_word = 0u;
  if (!BWD && num_batches_tracked != nullptr && cq == 0 && tid == 0) *num_batches_tracked += 1;
  double acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0;
  const float* col = partial + cq * 4;
  const int64_t rs = (int64_t)2 * c;   // floats per partial row
  int64_t r = tid;
  for (; r + 7 * (int64_t)kBnFusedThreads < nb; r += 8 * (int64_t)kBnFusedThreads) {
    float4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* p = col + (r + (int64_t)u * kBnFusedThreads) * rs;
      a[u] = *reinterpret_cast<const float4*>(p);
      b[u] = *reinterpret_cast<const float4*>(p + c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] += (double)a[u].x; acc[1] += (double)a[u].y; acc[2] += (double)a[u].z; acc[3] += (double)a[u].w;
      acc[4] += (double)b[u].x; acc[5] += (double)b[u].y; acc[6] += (double)b[u].z; acc[7] += (double)b[u].w;
    }
  }
  for (; r < nb; r += kBnFusedThreads) {
    const float* p = col + r * rs;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + c);
    acc[0] += (double)a.x; acc[1] += (double)a.y; acc[2] += (double)a.z; acc[3] += (double)a.w;
    acc[4] += (double)b.x; acc[5] += (double)b.y; acc[6] += (double)b.z; acc[7] += (double)b.w;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][j] = acc[j];
  __syncthreads();
  if (tid >= 4) return;
  const int ch = cq * 4 + tid;
  double s = 0.0, ss = 0.0;
  for (int w = 0; w < kBnFusedThreads / 64; ++w) { s += red[w][tid]; ss += red[w][4 + tid]; }
  if (!BWD) {
    const double m = s / (double)n;
    double v = ss / (double)n - m * m;
    if (v < 0.0) v = 0.0;
    o0[ch] = (float)m;
    o1[ch] = (float)v;
    if (r0) r0[ch] = (1.f - momentum) * r0[ch] + momentum * (float)m;
    if (r1) {
      const double unb = (n > 1) ? v * (double)n / (double)(n - 1) : v;
      r1[ch] = (1.f - momentum) * r1[ch] + momentum * (float)unb;
    }
  } else {
    if (o0) o0[ch] = (float)s;    // dbeta
    if (o1) o1[ch] = (float)ss;   // dgamma
    r0[ch] = (float)s;            // sums[0][ch]
    r0[c + ch] = (float)ss;       // sums[1][ch]
  }
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const double* __restrict__ partial, int nb, int c,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ sums /* [2][c]: dbeta, dgamma */,
                                                              unsigned* __restrict__ zero_word) {
  // the max|dx| word the dx kernel (next launch on this stream) accumulates into: cleared here when the reduction pass that
  // normally clears it did not run (sums taken from a conv epilogue)
  if (zero_word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;
  double s = 0.0, ss = 0.0;
  {  // same addition order as a plain loop; four trips' loads are issued together (the kernel was a chain of dependent loads)
    int b = lane;
    for (; b + 192 < nb; b += 256) {
      double a[4], q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[(int64_t)(b + 64 * u) * 2 * c + ch];
        q[u] = partial[(int64_t)(b + 64 * u) * 2 * c + c + ch];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s += a[u]; ss += q[u]; }
    }
    for (; b < nb; b += 64) {
      s += partial[(int64_t)b * 2 * c + ch];
      ss += partial[(int64_t)b * 2 * c + c + ch];
    }
  }
  wave_sum2(s, ss);
  if (lane != 0) return;
  if (dbeta) dbeta[ch] = (float)s;
  if (dgamma) dgamma[ch] = (float)ss;
  sums[ch] = (float)s;
  sums[c + ch] = (float)ss;
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, int64_t n, int c,
                                                       const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int relu, float* __restrict__ y, int y_stride,
                                                       int y_col0) {
  const int c4 = c >> 2;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * c4) return;
  const int64_t r = t / c4;
  const int cq = (int)(t - r * c4);
  const float4 xv = *reinterpret_cast<const float4*>(x + r * c + cq * 4);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = cq * 4 + j;
    const float istd = 1.0f / sqrtf(var[ch] + eps);
    float v = (xs[j] - mean[ch]) * istd * (gamma ? gamma[ch] : 1.f) + (beta ? beta[ch] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    o[j] = v;
  }
  *reinterpret_cast<float4*>(y + r * y_stride + y_col0 + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// dx = gamma * istd * (dyr - dbeta/N - xhat * dgamma/N)         (training-mode BN)
__global__ void __launch_bounds__(256) bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        int dy_stride, int dy_col0, int64_t n, int c,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int relu, const float* __restrict__ sums,
                                                        float* __restrict__ dx) {
  const int c4 = c >> 2;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * c4) return;
  const int64_t r = t / c4;
  const int cq = (int)(t - r * c4);
  const float4 xv = *reinterpret_cast<const float4*>(x + r * c + cq * 4);
  const float4 dv = *reinterpret_cast<const float4*>(dy + r * dy_stride + dy_col0 + cq * 4);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
  const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
  const float inv_n = 1.0f / (float)n;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = cq * 4 + j;
    const float istd = 1.0f / sqrtf(var[ch] + eps);
    const float g = gamma ? gamma[ch] : 1.f, bt = beta ? beta[ch] : 0.f;
    const float xh = (xs[j] - mean[ch]) * istd;
    float d = ds[j];
    if (relu && !(xh * g + bt > 0.f)) d = 0.f;
    o[j] = g * istd * (d - sums[ch] * inv_n - xh * sums[c + ch] * inv_n);
  }
  *reinterpret_cast<float4*>(dx + r * c + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// Fast paths for power-of-two channel counts (every layer of this model): thread = (float4 column cq, row lane rt), the
// per-channel constants are computed ONCE per thread, the row index needs no 64-bit division (which made the generic
// kernels ALU-bound at ~2.7 TB/s), and each thread has 4 independent 16-byte loads in flight.  Per-element arithmetic is the
// same expression as in the generic kernels: results are bit-identical.
static constexpr int kBnRowsPerThread = 4;

__global__ void __launch_bounds__(256) bn_apply_pow2_kernel(const float* __restrict__ x, int64_t n, int c, int lg_c4,
                                                            const float* __restrict__ mean, const float* __restrict__ var,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, int relu, float* __restrict__ y, int y_stride,
                                                            int y_col0) {
  const int c4 = 1 << lg_c4, R = 256 >> lg_c4;
  const int cq = threadIdx.x & (c4 - 1), rt = threadIdx.x >> lg_c4;
  float mu[4], istd[4], g[4], bt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = cq * 4 + j;
    mu[j] = mean[ch];
    istd[j] = 1.0f / sqrtf(var[ch] + eps);
    g[j] = gamma ? gamma[ch] : 1.f;
    bt[j] = beta ? beta[ch] : 0.f;
  }
  const int64_t r0 = (int64_t)blockIdx.x * (R * kBnRowsPerThread) + rt;
  float4 xv[kBnRowsPerThread];
#pragma unroll
  for (int u = 0; u < kBnRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * R;
    if (r < n) xv[u] = *reinterpret_cast<const float4*>(x + r * c + cq * 4);
  }
#pragma unroll
  for (int u = 0; u < kBnRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * R;
    if (r >= n) continue;
    const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = (xs[j] - mu[j]) * istd[j] * g[j] + bt[j];
      if (relu) v = fmaxf(v, 0.f);
      o[j] = v;
    }
    *reinterpret_cast<float4*>(y + r * y_stride + y_col0 + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

__global__ void __launch_bounds__(256) bn_bwd_dx_pow2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             int dy_stride, int dy_col0, int64_t n, int c, int lg_c4,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int relu, const float* __restrict__ sums,
                                                             float* __restrict__ dx, unsigned* __restrict__ absmax_out) {
  __shared__ float s_max[4];
  float amax = 0.f;
  const int c4 = 1 << lg_c4, R = 256 >> lg_c4;
  const int cq = threadIdx.x & (c4 - 1), rt = threadIdx.x >> lg_c4;
  const float inv_n = 1.0f / (float)n;
  float mu[4], istd[4], g[4], bt[4], s0[4], s1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = cq * 4 + j;
    mu[j] = mean[ch];
    istd[j] = 1.0f / sqrtf(var[ch] + eps);
    g[j] = gamma ? gamma[ch] : 1.f;
    bt[j] = beta ? beta[ch] : 0.f;
    s0[j] = sums[ch];
    s1[j] = sums[c + ch];
  }
  const int64_t r0 = (int64_t)blockIdx.x * (R * kBnRowsPerThread) + rt;
  float4 xv[kBnRowsPerThread], dv[kBnRowsPerThread];
#pragma unroll
  for (int u = 0; u < kBnRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * R;
    if (r < n) {
      xv[u] = *reinterpret_cast<const float4*>(x + r * c + cq * 4);
      dv[u] = *reinterpret_cast<const float4*>(dy + r * dy_stride + dy_col0 + cq * 4);
    }
  }
#pragma unroll
  for (int u = 0; u < kBnRowsPerThread; ++u) {
    const int64_t r = r0 + (int64_t)u * R;
    if (r >= n) continue;
    const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
    const float ds[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - mu[j]) * istd[j];
      float d = ds[j];
      if (relu && !(xh * g[j] + bt[j] > 0.f)) d = 0.f;
      o[j] = g[j] * istd[j] * (d - s0[j] * inv_n - xh * s1[j] * inv_n);
      amax = fmaxf(amax, fabsf(o[j]));
    }
    *reinterpret_cast<float4*>(dx + r * c + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
  // max |dx| of the whole tensor for the consumer's fixed-point group sum (saves it a pass over dx): one atomic per block
  if (absmax_out != nullptr) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
      if (amax > 0.f) atomicMax(absmax_out, __float_as_uint(amax));
    }
  }
}

// log2(c / 4) when c / 4 is a power of two <= 256, else -1
static inline int bn_lg_c4(int c) {
  const int c4 = c >> 2;
  if (c % 4 != 0 || c4 < 1 || c4 > 256 || (c4 & (c4 - 1)) != 0) return -1;
  int lg = 0;
  while ((1 << lg) < c4) ++lg;
  return lg;
}

static inline void bn_split(int64_t n, int c, int& nb, int64_t& rpb) {
  const int R = 256 / (c >> 2);
  int64_t want = cdiv(n, (int64_t)R * 8);
  if (want < 1) want = 1;
  if (want > kMaxBnBlocks) want = kMaxBnBlocks;
  rpb = cdiv(n, want);
  nb = (int)cdiv(n, rpb);
}

// vc_debug_set "bn_fused_partial": conv-epilogue partial rows -> statistics in ONE launch (bn_partial_fused_kernel) instead of
// slab reduce + finalize.  MEASURED SLOWER and therefore off (round 3, profiles/r03_bn_fused_partial_rejected.txt): a block per 4
// channels means 2-16 blocks pulling 0.5-5 MB through one CU each -- 24 us (statistics) / 31 us (backward sums) per launch against
// 6.5 + 5 us for the two-launch route, +0.5 ms per train step.  Kept for A/B runs only.
int g_bn_fused_partial = 0;

static inline bool bn_c_ok(int c) { return c == 4 || c == 8 || c == 16 || c == 32 || c == 64 || c == 128; }

int bn_bwd_dx_launch(const float* x, const float* dy, int dy_stride, int dy_col0, int64_t n, int c, const float* mean,
                     const float* var, const float* gamma, const float* beta, float eps, int relu, const float* sums, float* dx,
                     hipStream_t st) {
  VC_REQUIRE(bn_c_ok(c) && n >= 1 && x && dy && mean && var && sums && dx, "bn_bwd_dx_launch: null/invalid argument");
  VC_REQUIRE(dy_stride >= c && dy_stride % 4 == 0 && dy_col0 % 4 == 0 && dy_col0 + c <= dy_stride, "bn_bwd_dx_launch: bad stride arguments");
  const int lg = bn_lg_c4(c);
  VC_REQUIRE(lg >= 0, "bn_bwd_dx_launch: channel count must be a power of two");
  const int rows_per_block = (256 >> lg) * kBnRowsPerThread;
  const int tr = trace_open_aux(3, c, st);
  VC_LAUNCH_WITH_STOP_EVENT(bn_bwd_dx_pow2_kernel, dim3((unsigned)cdiv(n, rows_per_block)), dim3(256), 0, st, x, dy, dy_stride, dy_col0,
                     n, c, lg, mean, var, gamma, beta, eps, relu, sums, dx, (unsigned*)nullptr);
  if (tr >= 0) trace_close_aux(tr, 3, c, n, st);
  VC_CHECK_LAUNCH("bn_bwd_dx_kernel");
  return VC_OK;
}

}  // namespace vc

using namespace vc;

extern "C" {

size_t vc_bn_workspace_bytes(int64_t n, int c) {
  (void)n;
  if (c < 1) return 0;
  return (size_t)kMaxBnBlocks * 2 * c * sizeof(double) + 2 * c * sizeof(float) + 64;
}

int vc_bn_stats(const float* x, int64_t n, int c, float* mean, float* var, float* running_mean, float* running_var,
                int64_t* num_batches_tracked, float momentum, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(bn_c_ok(c), "vc_bn_stats: unsupported channel count %d", c);
  VC_REQUIRE(n >= 1 && x && mean && var && ws, "vc_bn_stats: null/invalid argument (n=%lld)", (long long)n);
  if (ws_bytes < vc_bn_workspace_bytes(n, c)) { set_error("vc_bn_stats: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  int nb;
  int64_t rpb;
  bn_split(n, c, nb, rpb);
  double* partial = (double*)ws;
  hipLaunchKernelGGL((bn_reduce_kernel<false>), dim3(nb), dim3(256), 0, st, x, (const float*)nullptr, 0, 0, n, c,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, 0,
                     rpb, partial, (unsigned*)nullptr);
  VC_CHECK_LAUNCH("bn_reduce_kernel<stats>");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, partial, nb, n, c, mean, var,
                     running_mean, running_var, (long long*)num_batches_tracked, momentum);
  VC_CHECK_LAUNCH("bn_stats_finalize_kernel");
  return VC_OK;
}

int vc_bn_stats_from_partial(const float* partial, int64_t nblocks, int64_t n, int c, float* mean, float* var,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                             void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(c >= 1 && n >= 1 && nblocks >= 1 && partial && mean && var, "vc_bn_stats_from_partial: null/invalid argument");
  hipStream_t st = (hipStream_t)stream;
  if (g_bn_fused_partial && c % 4 == 0) {  // one launch (round 3)
    hipLaunchKernelGGL((bn_partial_fused_kernel<false>), dim3(c / 4), dim3(kBnFusedThreads), 0, st, partial, nblocks, n, c, mean, var,
                       running_mean, running_var, (long long*)num_batches_tracked, momentum, (unsigned*)nullptr);
    VC_CHECK_LAUNCH("bn_partial_fused_kernel<stats>");
    return VC_OK;
  }
  if (nblocks > 512 && ws != nullptr && 2 * c <= 256 && (c & (c - 1)) == 0) {
    // many partial rows: coalesced slab reduce to <= 256 fp64 rows, then the ordinary finalize
    if (ws_bytes < vc_bn_workspace_bytes(n, c)) { set_error("vc_bn_stats_from_partial: workspace too small"); return VC_ECAPACITY; }
    const int64_t rpb = bn_partial_rows_per_group(nblocks);
    const int g = (int)cdiv(nblocks, rpb);
    double* dpartial = (double*)ws;
    hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(g), dim3(256), 0, st, partial, nblocks, rpb, c, dpartial);
    VC_CHECK_LAUNCH("bn_partial_reduce_kernel");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, dpartial, g, n, c, mean, var,
                       running_mean, running_var, (long long*)num_batches_tracked, momentum);
    VC_CHECK_LAUNCH("bn_stats_finalize_kernel");
    return VC_OK;
  }
  hipLaunchKernelGGL(bn_stats_from_partial_kernel, dim3((c + 3) / 4), dim3(256), 0, st, partial, nblocks, n,
                     c, mean, var, running_mean, running_var, (long long*)num_batches_tracked, momentum);
  VC_CHECK_LAUNCH("bn_stats_from_partial_kernel");
  return VC_OK;
}

int vc_bn_apply_relu(const float* x, int64_t n, int c, const float* mean, const float* var, const float* gamma,
                     const float* beta, float eps, int relu, float* y, int y_stride, int y_col0, void* stream) {
  VC_REQUIRE(c > 0 && c % 4 == 0 && y_stride >= c && y_stride % 4 == 0 && y_col0 % 4 == 0 && y_col0 + c <= y_stride,
             "vc_bn_apply_relu: bad channel/stride arguments (c=%d stride=%d col0=%d)", c, y_stride, y_col0);
  if (n == 0) return VC_OK;
  VC_REQUIRE(n > 0 && x && mean && var && y, "vc_bn_apply_relu: null argument");
  const int lg = bn_lg_c4(c);
  if (lg >= 0) {
    const int rows_per_block = (256 >> lg) * kBnRowsPerThread;
    hipLaunchKernelGGL(bn_apply_pow2_kernel, dim3((unsigned)cdiv(n, rows_per_block)), dim3(256), 0, (hipStream_t)stream, x,
                       n, c, lg, mean, var, gamma, beta, eps, relu, y, y_stride, y_col0);
  } else {
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)cdiv(n * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, n, c,
                       mean, var, gamma, beta, eps, relu, y, y_stride, y_col0);
  }
  VC_CHECK_LAUNCH("bn_apply_kernel");
  return VC_OK;
}

int vc_bn_relu_backward(const float* x, const float* dy, int dy_stride, int dy_col0, int64_t n, int c, const float* mean,
                        const float* var, const float* gamma, const float* beta, float eps, int relu, float* dx,
                        float* dgamma, float* dbeta, unsigned* absmax_out, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(bn_c_ok(c), "vc_bn_relu_backward: unsupported channel count %d", c);
  VC_REQUIRE(dy_stride >= c && dy_stride % 4 == 0 && dy_col0 % 4 == 0 && dy_col0 + c <= dy_stride,
             "vc_bn_relu_backward: bad stride arguments");
  VC_REQUIRE(n >= 1 && x && dy && mean && var && dx && ws, "vc_bn_relu_backward: null/invalid argument");
  if (ws_bytes < vc_bn_workspace_bytes(n, c)) { set_error("vc_bn_relu_backward: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  int nb;
  int64_t rpb;
  bn_split(n, c, nb, rpb);
  double* partial = (double*)ws;
  float* sums = (float*)(partial + (size_t)kMaxBnBlocks * 2 * c);
  hipLaunchKernelGGL((bn_reduce_kernel<true>), dim3(nb), dim3(256), 0, st, x, dy, dy_stride, dy_col0, n, c, mean, var,
                     gamma, beta, eps, relu, rpb, partial, absmax_out);
  VC_CHECK_LAUNCH("bn_reduce_kernel<bwd>");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, partial, nb, c, dgamma, dbeta, sums,
                     (unsigned*)nullptr);
  VC_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  const int lg = bn_lg_c4(c);
  if (lg >= 0) {
    const int rows_per_block = (256 >> lg) * kBnRowsPerThread;
    const int tr = trace_open_aux(3, c, st);
    VC_LAUNCH_WITH_STOP_EVENT(bn_bwd_dx_pow2_kernel, dim3((unsigned)cdiv(n, rows_per_block)), dim3(256), 0, st, x, dy, dy_stride,
                       dy_col0, n, c, lg, mean, var, gamma, beta, eps, relu, sums, dx, absmax_out);
    if (tr >= 0) trace_close_aux(tr, 3, c, n, st);
  } else {
    VC_REQUIRE(absmax_out == nullptr, "vc_bn_relu_backward: absmax_out needs a power-of-two channel count");
    VC_LAUNCH_WITH_STOP_EVENT(bn_bwd_dx_kernel, dim3((unsigned)cdiv(n * (c / 4), 256)), dim3(256), 0, st, x, dy, dy_stride,
                       dy_col0, n, c, mean, var, gamma, beta, eps, relu, sums, dx);
  }
  VC_CHECK_LAUNCH("bn_bwd_dx_kernel");
  return VC_OK;
}

int vc_bn_relu_backward_from_partial(const float* x, const float* dy, int dy_stride, int dy_col0, int64_t n, int c,
                                     const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                     int relu, const float* fpartial, int64_t nblocks, float* dx, float* dgamma, float* dbeta,
                                     unsigned* absmax_out, void* ws, size_t ws_bytes, void* stream) {
  VC_REQUIRE(bn_c_ok(c) && 2 * c <= 256, "vc_bn_relu_backward_from_partial: unsupported channel count %d", c);
  VC_REQUIRE(dy_stride >= c && dy_stride % 4 == 0 && dy_col0 % 4 == 0 && dy_col0 + c <= dy_stride,
             "vc_bn_relu_backward_from_partial: bad stride arguments");
  VC_REQUIRE(n >= 1 && nblocks >= 1 && x && dy && mean && var && dx && ws && fpartial,
             "vc_bn_relu_backward_from_partial: null/invalid argument");
  if (ws_bytes < vc_bn_workspace_bytes(n, c)) { set_error("vc_bn_relu_backward_from_partial: workspace too small"); return VC_ECAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  float* sums = (float*)(partial + (size_t)kMaxBnBlocks * 2 * c);
  if (g_bn_fused_partial) {  // one launch (round 3)
    hipLaunchKernelGGL((bn_partial_fused_kernel<true>), dim3(c / 4), dim3(kBnFusedThreads), 0, st, fpartial, nblocks, n, c, dbeta, dgamma,
                       sums, (float*)nullptr, (long long*)nullptr, 0.f, absmax_out);
    VC_CHECK_LAUNCH("bn_partial_fused_kernel<bwd>");
  } else {
    const int64_t rpb = bn_partial_rows_per_group(nblocks);
    const int g = (int)cdiv(nblocks, rpb);
    hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3(g), dim3(256), 0, st, fpartial, nblocks, rpb, c, partial);
    VC_CHECK_LAUNCH("bn_partial_reduce_kernel");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, partial, g, c, dgamma, dbeta, sums, absmax_out);
    VC_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  }
  const int lg = bn_lg_c4(c);
  VC_REQUIRE(lg >= 0, "vc_bn_relu_backward_from_partial: channel count must be a power of two");
  const int rows_per_block = (256 >> lg) * kBnRowsPerThread;
  const int tr = trace_open_aux(3, c, st);
  VC_LAUNCH_WITH_STOP_EVENT(bn_bwd_dx_pow2_kernel, dim3((unsigned)cdiv(n, rows_per_block)), dim3(256), 0, st, x, dy, dy_stride, dy_col0,
                     n, c, lg, mean, var, gamma, beta, eps, relu, sums, dx, absmax_out);
  if (tr >= 0) trace_close_aux(tr, 3, c, n, st);
  VC_CHECK_LAUNCH("bn_bwd_dx_kernel");
  return VC_OK;
}

}  // extern "C"
