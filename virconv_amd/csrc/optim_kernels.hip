// Gradient clip + Adam step with true weight decay over a few flat fp32 parameter vectors, in two launches.
//
// Reference semantics: tools/train_utils/train_utils.py:50-51 (`clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP)` then
// `optimizer.step()`), the optimizer being tools/train_utils/optimization/__init__.py:19-32 (`adam_onecycle`: torch Adam with betas
// (0.9, 0.99) inside fastai's OptimWrapper with true_wd = bn_wd = True) whose step is fastai_optim.py:132-149: every parameter is
// multiplied by (1 - wd * lr), then the plain Adam update runs with weight_decay 0 -- which is torch.optim.AdamW.  Learning rate and
// beta1 change every iteration under the one-cycle schedule, so both are arguments of the call, not state.
//
// Why it exists: with the backbone's parameters living in one flat tensor (feature_pass.flatten_parameters) the stock route is
// `clip_grad_norm_` (11 small launches) + the fused multi-tensor AdamW, which gets 7 blocks for one 430 k-element tensor (43 us): 0.10 ms of
// launches behind every backward pass for 1.7 MB of state.  Here:
//   * grad_sqsum_kernel   kNormBlocks blocks per tensor, each a fixed strided slice of its gradient -> one fp32 partial of sum g^2 per
//                         block (a fixed tree: per-thread chain in element order, wave shuffle, four wave sums in order);
//   * clip_adamw_kernel   every block first adds all partials in a fixed order (in double: 128 values per tensor from L2) and so holds
//                         the same total norm and the same clip coefficient -- no ticket, no atomics, no zeroed workspace --
//                         then updates its elements: g' = g * coef; p *= 1 - lr * wd; m += (1 - b1) * (g' - m);
//                         v = b2 * v + (1 - b2) * g'^2; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
// HBM-bound in principle (7 floats per element moved: 12 MB at 430 k elements, 1.5 us at 8 TB/s); in practice two launch latencies.
// The gradient is rescaled in memory (what torch's clip_grad_norm_ leaves in .grad) only if the caller asks for it (scale_grads).
#include "common.h"

#include <math.h>

namespace vc {

static constexpr int kNormBlocks = 128;

struct AdamBatch {
  vc_adam_tensor t[VC_ADAM_MAX_TENSORS];
  unsigned block0[VC_ADAM_MAX_TENSORS + 1];   // first block of tensor i in clip_adamw_kernel's grid
  int vec[VC_ADAM_MAX_TENSORS];               // all four pointers 16-byte aligned: float4 body + scalar tail
  int n;
};

__global__ void __launch_bounds__(256) grad_sqsum_kernel(AdamBatch B, float* __restrict__ partial) {
  __shared__ float wsum[4];
  const vc_adam_tensor& S = B.t[blockIdx.y];
  const float* __restrict__ g = S.grad;
  const int64_t n = S.n;
  float acc = 0.f;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, T = (int64_t)kNormBlocks * 256;
  if (B.vec[blockIdx.y]) {
    const int64_t n4 = n >> 2;
    for (int64_t i = t; i < n4; i += T) {
      const float4 x = reinterpret_cast<const float4*>(g)[i];
      acc += x.x * x.x; acc += x.y * x.y; acc += x.z * x.z; acc += x.w * x.w;
    }
    if (t < (n & 3)) { const float x = g[(n4 << 2) + t]; acc += x * x; }
  } else {
    for (int64_t i = t; i < n; i += T) { const float x = g[i]; acc += x * x; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * kNormBlocks + blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

struct AdamArgs {
  float beta1, beta2, eps, decay;   // decay = 1 - lr * weight_decay
  float step_size, inv_bc2_sqrt;    // lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)
  float max_norm;                   // <= 0: no clipping (the norm is still reported)
  int scale_grads;                  // write grad * coef back (torch's clip_grad_norm_ does; nothing on this path reads it)
};

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
  g *= coef;
  p *= a.decay;
  m = fmaf(1.0f - a.beta1, g - m, m);
  v = fmaf(a.beta2, v, (1.0f - a.beta2) * g * g);
  const float denom = fmaf(sqrtf(v), a.inv_bc2_sqrt, a.eps);
  p -= a.step_size * (m / denom);
  return g;
}

__global__ void __launch_bounds__(256) clip_adamw_kernel(AdamBatch B, AdamArgs a, const float* __restrict__ partial,
                                                         float* __restrict__ total_norm) {
  __shared__ float s_coef;
  if (threadIdx.x < 64) {   // one wave adds every partial in a fixed order -> the same value in every block
    double s = 0.0;
    for (int i = threadIdx.x; i < B.n * kNormBlocks; i += 64) s += (double)partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) {
      const float norm = sqrtf((float)s);
      float coef = 1.0f;
      if (a.max_norm > 0.f) {
        const float c = a.max_norm / (norm + 1e-6f);   // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (total_norm + 1e-6), max = 1)
        coef = c < 1.0f ? c : (c != c ? c : 1.0f);     // a NaN norm propagates, as torch.clamp does
      }
      s_coef = coef;
      if (blockIdx.x == 0 && total_norm) *total_norm = norm;
    }
  }
  __syncthreads();
  const float coef = s_coef;
  int s = 0;
#pragma unroll
  for (int i = 1; i < VC_ADAM_MAX_TENSORS; ++i)
    if (i < B.n && blockIdx.x >= B.block0[i]) s = i;
  const vc_adam_tensor& S = B.t[s];
  float* __restrict__ p = S.param;
  float* __restrict__ g = S.grad;
  float* __restrict__ m = S.exp_avg;
  float* __restrict__ v = S.exp_avg_sq;
  const int64_t n = S.n;
  const int64_t t = (int64_t)(blockIdx.x - B.block0[s]) * 256 + threadIdx.x;
  if (B.vec[s]) {
    const int64_t n4 = n >> 2;
    if (t < n4) {
      float4 P = reinterpret_cast<float4*>(p)[t], M = reinterpret_cast<float4*>(m)[t], V = reinterpret_cast<float4*>(v)[t];
      float4 G = reinterpret_cast<const float4*>(g)[t];
      G.x = adam_one(P.x, G.x, M.x, V.x, a, coef); G.y = adam_one(P.y, G.y, M.y, V.y, a, coef);
      G.z = adam_one(P.z, G.z, M.z, V.z, a, coef); G.w = adam_one(P.w, G.w, M.w, V.w, a, coef);
      reinterpret_cast<float4*>(p)[t] = P; reinterpret_cast<float4*>(m)[t] = M; reinterpret_cast<float4*>(v)[t] = V;
      if (a.scale_grads) reinterpret_cast<float4*>(g)[t] = G;
    } else if (t - n4 < (n & 3)) {
      const int64_t i = (n4 << 2) + (t - n4);
      float P = p[i], M = m[i], V = v[i];
      const float G = adam_one(P, g[i], M, V, a, coef);
      p[i] = P; m[i] = M; v[i] = V;
      if (a.scale_grads) g[i] = G;
    }
  } else if (t < n) {
    float P = p[t], M = m[t], V = v[t];
    const float G = adam_one(P, g[t], M, V, a, coef);
    p[t] = P; m[t] = M; v[t] = V;
    if (a.scale_grads) g[t] = G;
  }
}

}  // namespace vc

using namespace vc;

extern "C" {

size_t vc_clip_adamw_workspace_bytes(int n_tensors) {
  return (size_t)(n_tensors > 0 ? n_tensors : 0) * kNormBlocks * sizeof(float);
}

int vc_clip_adamw(const vc_adam_tensor* tensors, int n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, float max_norm, int scale_grads, float* total_norm, void* workspace, size_t workspace_bytes, void* stream) {
  VC_REQUIRE(n_tensors >= 0 && n_tensors <= VC_ADAM_MAX_TENSORS, "vc_clip_adamw: %d tensors (at most %d: flatten the parameters)", n_tensors,
             VC_ADAM_MAX_TENSORS);
  VC_REQUIRE(step >= 1, "vc_clip_adamw: step = %lld (the first step is 1)", (long long)step);
  VC_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "vc_clip_adamw: betas (%g, %g) outside [0, 1)", beta1, beta2);
  hipStream_t st = (hipStream_t)stream;
  AdamBatch B{};
  unsigned blocks = 0;
  for (int i = 0; i < n_tensors; ++i) {
    const vc_adam_tensor& T = tensors[i];
    VC_REQUIRE(T.n >= 0, "vc_clip_adamw: tensor %d has n = %lld", i, (long long)T.n);
    if (T.n == 0) continue;
    VC_REQUIRE(T.param && T.grad && T.exp_avg && T.exp_avg_sq, "vc_clip_adamw: tensor %d has a NULL pointer", i);
    const uintptr_t bits = (uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg | (uintptr_t)T.exp_avg_sq;
    const int k = B.n++;
    B.t[k] = T;
    B.vec[k] = (bits & 15) == 0;
    B.block0[k] = blocks;
    blocks += (unsigned)cdiv(B.vec[k] ? (T.n >> 2) + (T.n & 3) : T.n, 256);
  }
  B.block0[B.n] = blocks;
  if (B.n == 0) {
    if (total_norm) VC_CHECK_HIP(hipMemsetAsync(total_norm, 0, sizeof(float), st));
    return VC_OK;
  }
  VC_REQUIRE(workspace != nullptr && workspace_bytes >= vc_clip_adamw_workspace_bytes(B.n), "vc_clip_adamw: workspace of %zu bytes, need %zu",
             workspace_bytes, vc_clip_adamw_workspace_bytes(B.n));
  AdamArgs a;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.decay = (float)(1.0 - (double)lr * (double)weight_decay);
  a.step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  a.max_norm = max_norm;
  a.scale_grads = scale_grads != 0;
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(grad_sqsum_kernel, dim3(kNormBlocks, B.n), dim3(256), 0, st, B, partial);
  VC_CHECK_LAUNCH("grad_sqsum_kernel");
  hipLaunchKernelGGL(clip_adamw_kernel, dim3(blocks), dim3(256), 0, st, B, a, (const float*)partial, total_norm);
  VC_CHECK_LAUNCH("clip_adamw_kernel");
  return VC_OK;
}

}  // extern "C"
