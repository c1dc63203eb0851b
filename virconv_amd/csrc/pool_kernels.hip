// RoI-grid-pooling operators on the backbone outputs for gfx950:
//   voxel index   occupancy bitmap + the locality-preserving coordinate hash of K3, instead of the reference's dense
//                 (B, Z, Y, X) int32 index volume (pcdet/utils/spconv_utils.py:4-21: 12 MB per frame at x_conv3)
//   voxel query   pointnet2_stack/src/voxel_query_gpu.cu:10-90, one WAVE per query point: lane = one (dz, dy) line of the
//                 neighbourhood (its 2*x_range+1 cells are one bit-field of the bitmap), hits are ordered with a wave
//                 prefix sum so the result is exactly the reference's sequential dz, dy, dx scan order
//   group points  pointnet2_stack/src/group_points_gpu.cu:15-100, one wave per point, rows read coalesced and transposed
//                 through LDS into the reference's (M, C, nsample) layout
#include "common.h"

namespace vc {

// ------------------------------------------------------------------------------------------------- voxel index
__global__ void __launch_bounds__(256) vq_mark_kernel(const int32_t* __restrict__ indices, int64_t n, int D, int H, int W,
                                                      unsigned long long* __restrict__ bitmap) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t* c = indices + i * 4;
  const uint64_t L = (((uint64_t)c[0] * D + c[1]) * H + c[2]) * W + c[3];
  atomicOr(&bitmap[L >> 6], 1ULL << (L & 63));
}

// bits [L, L + nbits) of the bitmap as the low bits of a 64-bit word (nbits <= 63)
__device__ __forceinline__ unsigned long long vq_bits(const unsigned long long* __restrict__ bitmap, uint64_t L, int nbits,
                                                      uint64_t nwords) {
  const uint64_t w = L >> 6;
  const int sh = (int)(L & 63);
  unsigned long long v = bitmap[w] >> sh;
  if (sh + nbits > 64 && w + 1 < nwords) v |= bitmap[w + 1] << (64 - sh);
  return v & ((1ULL << nbits) - 1ULL);
}

// Which cells of the line starting at linear cell L hold a voxel whose centre is within the radius?  Returns a bit mask of
// positions in the line (ascending x), tested in float32 with one rounding per operation, left to right.
__device__ __forceinline__ unsigned long long vq_hits(unsigned long long occ, uint64_t L, const uint64_t* __restrict__ keys,
                                                      const int32_t* __restrict__ vals, uint64_t mask,
                                                      const float* __restrict__ xyz, float qx, float qy, float qz,
                                                      float r2) {
  unsigned long long hit = 0ULL;
  while (occ) {
    const int p = __ffsll((long long)occ) - 1;
    occ &= occ - 1;
    const int row = hash_lookup(keys, vals, mask, L + (uint64_t)p);
    if (row < 0) continue;  // cannot happen for a consistent index; keeps a corrupted one from faulting
    const float dx = __fsub_rn(xyz[(int64_t)row * 3 + 0], qx);
    const float dy = __fsub_rn(xyz[(int64_t)row * 3 + 1], qy);
    const float dz = __fsub_rn(xyz[(int64_t)row * 3 + 2], qz);
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    if (!(d2 > r2)) hit |= 1ULL << p;
  }
  return hit;
}

__global__ void __launch_bounds__(256) voxel_query_kernel(const unsigned long long* __restrict__ bitmap, uint64_t nwords,
                                                          const uint64_t* __restrict__ keys,
                                                          const int32_t* __restrict__ vals, uint64_t mask, int B, int D,
                                                          int H, int W, const float* __restrict__ xyz,
                                                          const float* __restrict__ new_xyz,
                                                          const int32_t* __restrict__ new_coords, int64_t m, int zr,
                                                          int yr, int xr, float radius, int nsample,
                                                          int32_t* __restrict__ idx, uint8_t* __restrict__ empty) {
  const int lane = threadIdx.x & 63;
  const int64_t pt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= m) return;  // whole wave
  const float qx = new_xyz[pt * 3 + 0], qy = new_xyz[pt * 3 + 1], qz = new_xyz[pt * 3 + 2];
  const int b = new_coords[pt * 4 + 0], cz = new_coords[pt * 4 + 1], cy = new_coords[pt * 4 + 2],
            cx = new_coords[pt * 4 + 3];
  const float r2 = __fmul_rn(radius, radius);
  const int ny = 2 * yr + 1, nlines = (2 * zr + 1) * ny;
  const int xs = max(cx - xr, 0), xe = min(cx + xr, W - 1);
  // <= 63 (checked on the host); <= 0 when the query column is outside the grid (or the batch index is invalid)
  const int nbits = (b >= 0 && b < B) ? xe - xs + 1 : 0;
  int32_t* out = idx + pt * nsample;
  int cnt = 0;        // wave-uniform: hits written so far
  int first = -1;     // wave-uniform: row of the first hit
  for (int base = 0; base < nlines && cnt < nsample; base += 64) {
    const int l = base + lane;
    unsigned long long hit = 0ULL;
    uint64_t L = 0;
    if (l < nlines && nbits > 0) {
      const int z = cz + l / ny - zr, y = cy + l % ny - yr;
      if (z >= 0 && z < D && y >= 0 && y < H) {
        L = (((uint64_t)b * D + z) * H + y) * W + xs;
        const unsigned long long occ = vq_bits(bitmap, L, nbits, nwords);
        if (occ) hit = vq_hits(occ, L, keys, vals, mask, xyz, qx, qy, qz, r2);
      }
    }
    // ordered compaction: lanes are lines in scan order, bits inside a lane are ascending x
    const int c = __popcll(hit);
    int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    const int total = __shfl(inc, 63, 64);
    int pos = cnt + inc - c;
    unsigned long long h = hit;
    while (h && pos < nsample) {
      const int p = __ffsll((long long)h) - 1;
      h &= h - 1;
      out[pos++] = hash_lookup(keys, vals, mask, L + (uint64_t)p);
    }
    if (first < 0 && total > 0) {
      const unsigned long long any = __ballot(c > 0);
      const int src = __ffsll((long long)any) - 1;
      const int mine = (c > 0) ? hash_lookup(keys, vals, mask, L + (uint64_t)(__ffsll((long long)hit) - 1)) : -1;
      first = __shfl(mine, src, 64);
    }
    cnt += total;
  }
  cnt = min(cnt, nsample);
  // slots beyond the hits hold the first hit (the reference fills all slots when it finds the first one); empty ball: the
  // Python wrapper of the reference zeroes the row and returns the mask (voxel_query_utils.py:39-44)
  for (int s = cnt + lane; s < nsample; s += 64) out[s] = (first >= 0) ? first : 0;
  if (lane == 0) empty[pt] = (first < 0) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------- group points
__device__ __forceinline__ int64_t gp_batch_start(int B, int64_t pt, const int32_t* __restrict__ idx_batch_cnt,
                                                  const int32_t* __restrict__ features_batch_cnt) {
  int bs = 0;
  int64_t acc = idx_batch_cnt[0];
  for (int k = 1; k < B; ++k) {
    if (pt < acc) break;
    acc += idx_batch_cnt[k];
    bs = k;
  }
  int64_t start = 0;
  for (int k = 0; k < bs; ++k) start += features_batch_cnt[k];
  return start;
}

// one wave per point; LDS tile [min(C,64)][nsample + 1] per wave, channels in chunks of 64
template <bool GRAD>
__global__ void __launch_bounds__(256) group_points_kernel(int B, int64_t m, int C, int nsample,
                                                           float* __restrict__ features /* GRAD: grad_features */,
                                                           const int32_t* __restrict__ features_batch_cnt,
                                                           const int32_t* __restrict__ idx,
                                                           const int32_t* __restrict__ idx_batch_cnt,
                                                           float* __restrict__ out /* GRAD: grad_out (read) */) {
  extern __shared__ float tile_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pt = (int64_t)blockIdx.x * 4 + wave;
  if (pt >= m) return;
  const int ld = nsample + 1;
  float* tile = tile_all + (size_t)wave * 64 * ld;
  const int64_t start = gp_batch_start(B, pt, idx_batch_cnt, features_batch_cnt);
  const int32_t* id = idx + pt * nsample;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int cw = min(64, C - c0);
    float* blk = out + (pt * C + c0) * nsample;  // cw x nsample contiguous floats
    if (!GRAD) {
      for (int s = 0; s < nsample; ++s) {
        const int64_t row = start + id[s];
        if (lane < cw) tile[lane * ld + s] = features[row * C + c0 + lane];
      }
      __builtin_amdgcn_wave_barrier();
      for (int e = lane; e < cw * nsample; e += 64) blk[e] = tile[(e / nsample) * ld + e % nsample];
    } else {
      for (int e = lane; e < cw * nsample; e += 64) tile[(e / nsample) * ld + e % nsample] = blk[e];
      __builtin_amdgcn_wave_barrier();
      // consecutive samples that repeat a row (the unused slots of a ball all hold its first hit) are summed in registers
      // and flushed with one atomic; exact zeros (the gradient of an emptied ball) are not sent at all -- with the
      // reference's kernel those are tens of thousands of same-address atomics on row 0
      int64_t cur = start + id[0];
      float acc = (lane < cw) ? tile[lane * ld] : 0.f;
      for (int s = 1; s < nsample; ++s) {
        const int64_t row = start + id[s];
        const float v = (lane < cw) ? tile[lane * ld + s] : 0.f;
        if (row != cur) {
          if (lane < cw && acc != 0.f) atomicAdd(&features[cur * C + c0 + lane], acc);
          cur = row;
          acc = v;
        } else {
          acc += v;
        }
      }
      if (lane < cw && acc != 0.f) atomicAdd(&features[cur * C + c0 + lane], acc);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

static inline void vq_layout(int batch, const int32_t* shape, int64_t n, uint64_t& nwords, uint64_t& cap) {
  const uint64_t cells = (uint64_t)batch * shape[0] * shape[1] * shape[2];
  nwords = (cells + 63) / 64 + 1;
  cap = coord_hash_capacity(n);
}

}  // namespace vc

using namespace vc;

extern "C" {

size_t vc_voxel_index_workspace_bytes(int64_t n, int batch_size, const int32_t* spatial_shape) {
  if (n < 0 || batch_size < 1 || !spatial_shape) return 0;
  uint64_t nwords, cap;
  vq_layout(batch_size, spatial_shape, n, nwords, cap);
  return (size_t)(nwords * 8 + cap * 12);
}

int vc_voxel_index_build(const int32_t* indices, int64_t n, int batch_size, const int32_t* spatial_shape, void* ws,
                         size_t ws_bytes, void* stream) {
  VC_REQUIRE(n >= 0 && batch_size >= 1 && spatial_shape && ws && (indices || n == 0), "vc_voxel_index_build: null/invalid argument");
  if (ws_bytes < vc_voxel_index_workspace_bytes(n, batch_size, spatial_shape)) {
    set_error("vc_voxel_index_build: workspace too small");
    return VC_ECAPACITY;
  }
  uint64_t nwords, cap;
  vq_layout(batch_size, spatial_shape, n, nwords, cap);
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* bitmap = (unsigned long long*)ws;
  VC_CHECK_HIP(hipMemsetAsync(bitmap, 0, nwords * 8, st));
  const int rc = vc_hash_build(indices, n, 3, spatial_shape, (char*)ws + nwords * 8, cap * 12, stream);
  if (rc != VC_OK) return rc;
  if (n > 0) {
    hipLaunchKernelGGL(vq_mark_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, indices, n, spatial_shape[0],
                       spatial_shape[1], spatial_shape[2], bitmap);
    VC_CHECK_LAUNCH("vq_mark_kernel");
  }
  return VC_OK;
}

int vc_voxel_query(const void* ws, size_t ws_bytes, int64_t n, int batch_size, const int32_t* spatial_shape,
                   const float* xyz, const float* new_xyz, const int32_t* new_coords, int64_t m, int z_range, int y_range,
                   int x_range, float radius, int nsample, int32_t* idx, uint8_t* empty_mask, void* stream) {
  VC_REQUIRE(n >= 0 && m >= 0 && batch_size >= 1 && spatial_shape && ws, "vc_voxel_query: null/invalid argument");
  VC_REQUIRE(z_range >= 0 && y_range >= 0 && x_range >= 0 && x_range <= 31, "vc_voxel_query: ranges must be >= 0 and x_range <= 31");
  VC_REQUIRE(nsample >= 1, "vc_voxel_query: nsample must be >= 1");
  if (m == 0) return VC_OK;
  VC_REQUIRE(new_xyz && new_coords && idx && empty_mask && (xyz || n == 0), "vc_voxel_query: null argument");
  if (ws_bytes < vc_voxel_index_workspace_bytes(n, batch_size, spatial_shape)) {
    set_error("vc_voxel_query: workspace too small");
    return VC_ECAPACITY;
  }
  uint64_t nwords, cap;
  vq_layout(batch_size, spatial_shape, n, nwords, cap);
  const unsigned long long* bitmap = (const unsigned long long*)ws;
  const uint64_t* keys = (const uint64_t*)((const char*)ws + nwords * 8);
  const int32_t* vals = (const int32_t*)(keys + cap);
  hipLaunchKernelGGL(voxel_query_kernel, dim3((unsigned)cdiv(m, 4)), dim3(256), 0, (hipStream_t)stream, bitmap, nwords, keys,
                     vals, cap - 1, batch_size, spatial_shape[0], spatial_shape[1], spatial_shape[2], xyz, new_xyz, new_coords, m,
                     z_range, y_range, x_range, radius, nsample, idx, empty_mask);
  VC_CHECK_LAUNCH("voxel_query_kernel");
  return VC_OK;
}

int vc_group_points(int batch_size, int64_t m, int c, int nsample, const float* features,
                    const int32_t* features_batch_cnt, const int32_t* idx, const int32_t* idx_batch_cnt, float* out,
                    void* stream) {
  VC_REQUIRE(batch_size >= 1 && m >= 0 && c >= 1 && nsample >= 1, "vc_group_points: invalid argument");
  if (m == 0) return VC_OK;
  VC_REQUIRE(features && features_batch_cnt && idx && idx_batch_cnt && out, "vc_group_points: null argument");
  const size_t lds = (size_t)4 * 64 * (nsample + 1) * sizeof(float);
  VC_REQUIRE(lds <= 64 * 1024, "vc_group_points: nsample too large");
  hipLaunchKernelGGL((group_points_kernel<false>), dim3((unsigned)cdiv(m, 4)), dim3(256), lds, (hipStream_t)stream,
                     batch_size, m, c, nsample, const_cast<float*>(features), features_batch_cnt, idx, idx_batch_cnt, out);
  VC_CHECK_LAUNCH("group_points_kernel");
  return VC_OK;
}

int vc_group_points_grad(int batch_size, int64_t m, int c, int64_t n, int nsample, const float* grad_out,
                         const int32_t* idx, const int32_t* idx_batch_cnt, const int32_t* features_batch_cnt,
                         float* grad_features, void* stream) {
  VC_REQUIRE(batch_size >= 1 && m >= 0 && c >= 1 && n >= 0 && nsample >= 1, "vc_group_points_grad: invalid argument");
  if (n > 0) {
    VC_REQUIRE(grad_features, "vc_group_points_grad: null argument");
    VC_CHECK_HIP(hipMemsetAsync(grad_features, 0, (size_t)n * c * sizeof(float), (hipStream_t)stream));
  }
  if (m == 0) return VC_OK;
  VC_REQUIRE(grad_out && features_batch_cnt && idx && idx_batch_cnt && grad_features, "vc_group_points_grad: null argument");
  const size_t lds = (size_t)4 * 64 * (nsample + 1) * sizeof(float);
  VC_REQUIRE(lds <= 64 * 1024, "vc_group_points_grad: nsample too large");
  hipLaunchKernelGGL((group_points_kernel<true>), dim3((unsigned)cdiv(m, 4)), dim3(256), lds, (hipStream_t)stream,
                     batch_size, m, c, nsample, grad_features, features_batch_cnt, idx, idx_batch_cnt,
                     const_cast<float*>(grad_out));
  VC_CHECK_LAUNCH("group_points_grad_kernel");
  return VC_OK;
}

}  // extern "C"
