// Data front-end of the VirConv hot path on the GPU (SURVEY §8 a2 + f-rank-2):
//   vc_input_discard            StVD input point discard, bin-based  (pcdet/datasets/dataset.py:120-189)
//   vc_frontend_voxelize_mean   raw LiDAR + raw virtual points (fp32 or the fp16 of the .npy files) -> input discard ->
//                               LiDAR-first concat (dataset.py:270-294, data_processor.py:152-155) -> first-touch voxeliser
//                               + fused MeanVFE (data_processor.py:14-59, mean_vfe.py:39-49); nothing touches the host.
//
// The reference's algorithm (App-A.12), restated for a device that cannot "points[mask]":
//   bins over x, far -> near; bin i = [inter*i, inter*(i+1)), the last bin is open-ended, x < 0 belongs to no bin;
//   running retain test over the bin COUNTS decides `position` / `distant_acc`; per_bin = int((int(N*retain) - distant_acc) /
//   (position + 1e-4)); the nearest `position` bins that hold more than per_bin points keep perm_i[:per_bin] of their
//   points (in that order), every other bin keeps all its points in input order; output = bins concatenated far -> near.
// Kernels: (1) one ballot per bin and wave -> per-wave bin counts; (2) one block scans them per bin (wave per bin) and one
// thread replays the reference's integer/double arithmetic on the counts; (3) every point takes its stable in-bin rank from
// the scanned counts + a ballot prefix and is either copied to its output row (kept-whole bins) or listed in its bin's row
// list; (4) the reduced bins emit rows list[perm(j)], j < per_bin, perm = an injected permutation (parity tests) or a
// point-wise Feistel permutation (common.h), and the unused tail of the output is filled with out-of-range sentinel rows so
// that the voxeliser can run over the whole capacity without knowing the device-side count.
#include <hip/hip_fp16.h>

#include "common.h"

namespace vc {

static constexpr int kMaxBins = 16;

struct DiscardHeader {      // lives at the start of the workspace
  int32_t count[kMaxBins];  // points per bin (index = bin id, near -> far)
  int32_t kept[kMaxBins];   // rows the bin contributes to the output
  int32_t start[kMaxBins];  // first output row of the bin (bins are emitted far -> near)
  int32_t list0[kMaxBins];  // first entry of the bin's row list (reduced bins only use it)
  int32_t reduced[kMaxBins];
  int32_t n_out, per_bin, position, distant_acc;
};

struct DiscardEdges {
  float lo[kMaxBins];  // fp32 bin edges inter*i: the reference compares its float32 x against them in float32 (numpy)
  int nb;
};

struct DiscardPerms {
  const int64_t* p[kMaxBins];
};

template <bool HALF>
__device__ __forceinline__ float load_x(const void* pts, int64_t i, int f) {
  if (HALF) return __half2float(reinterpret_cast<const __half*>(pts)[i * f]);
  return reinterpret_cast<const float*>(pts)[i * f];
}

__device__ __forceinline__ int bin_of(float x, const DiscardEdges& e) {
  // i with lo[i] <= x < lo[i+1]; last bin open-ended; x < lo[0] = 0 (or NaN): none
  int b = -1;
#pragma unroll
  for (int i = 0; i < kMaxBins; ++i)
    if (i < e.nb && x >= e.lo[i]) b = i;
  return b;
}

template <bool HALF>
__global__ void __launch_bounds__(256) idisc_count_kernel(const void* __restrict__ pts, int64_t p, int f, DiscardEdges e,
                                                          int32_t* __restrict__ wavecnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t w = i >> 6;
  const int b = (i < p) ? bin_of(load_x<HALF>(pts, i, f), e) : -1;
  int mine = 0;
  for (int k = 0; k < e.nb; ++k) {
    const int c = __popcll(__ballot(b == k));
    if (lane == k) mine = c;
  }
  if (lane < kMaxBins && (w << 6) < p) wavecnt[w * kMaxBins + lane] = (lane < e.nb) ? mine : 0;
}

// one block, 16 waves: wave b turns the per-wave counts of bin b into exclusive prefixes (in place); then thread 0 replays
// dataset.py:120-189 on the totals.  Python semantics: `/` is float64 division, int() truncates toward zero.
__global__ void __launch_bounds__(1024) idisc_scan_kernel(int32_t* __restrict__ wavecnt, int64_t nwaves, int64_t p, int nb,
                                                          double retain, DiscardHeader* __restrict__ hdr) {
  __shared__ int total[kMaxBins];
  const int lane = threadIdx.x & 63, b = threadIdx.x >> 6;
  if (b < nb) {
    int run = 0;
    for (int64_t base = 0; base < nwaves; base += 64) {
      const int64_t w = base + lane;
      const int v = (w < nwaves) ? wavecnt[w * kMaxBins + b] : 0;
      int inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
      }
      if (w < nwaves) wavecnt[w * kMaxBins + b] = run + inc - v;
      run += __shfl(inc, 63, 64);
    }
    if (lane == 0) total[b] = run;
  } else if (lane == 0 && b < kMaxBins) {
    total[b] = 0;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  long long acc = 0, distant_acc = 0;
  int position = nb - 1;
  for (int j = 0; j < nb; ++j) {  // far -> near (dataset.py:137-164)
    const int i = nb - 1 - j;
    acc += total[i];
    const long long sampled = acc + (long long)i * total[i];
    if (p > 0 && (double)sampled / (double)p < retain) { position = i; distant_acc = acc; }
  }
  if (position < 0) position = 0;
  const long long out_n = (long long)((double)p * retain);                                         // int(N * retain)
  const long long per_bin = (long long)((double)(out_n - distant_acc) / ((double)position + 0.0001));  // int(.../(pos+1e-4))
  int row = 0, list = 0;
  for (int j = 0; j < nb; ++j) {  // parts[] order = far -> near; parts[len - pos:] = the nearest `position` bins
    const int i = nb - 1 - j;
    const bool red = (j >= nb - position) && ((long long)total[i] > per_bin);
    const int kept = red ? (int)(per_bin > 0 ? per_bin : 0) : total[i];
    hdr->count[i] = total[i];
    hdr->kept[i] = kept;
    hdr->start[i] = row;
    hdr->reduced[i] = red ? 1 : 0;
    hdr->list0[i] = list;
    row += kept;
    if (red) list += total[i];
  }
  for (int i = nb; i < kMaxBins; ++i) { hdr->count[i] = hdr->kept[i] = hdr->start[i] = hdr->list0[i] = hdr->reduced[i] = 0; }
  hdr->n_out = row;
  hdr->per_bin = (int)(per_bin > 0x7fffffffLL ? 0x7fffffffLL : (per_bin < 0 ? 0 : per_bin));
  hdr->position = position;
  hdr->distant_acc = (int)distant_acc;
}

template <bool HALF>
__device__ __forceinline__ void copy_row(const void* __restrict__ pts, int64_t src, float* __restrict__ out, int64_t dst,
                                         int f, float intensity_div) {
  for (int c = 0; c < f; ++c) {
    float v = HALF ? __half2float(reinterpret_cast<const __half*>(pts)[src * f + c])
                   : reinterpret_cast<const float*>(pts)[src * f + c];
    if (c == 3 && intensity_div != 0.0f) v = __fdiv_rn(v, intensity_div);  // points[:, 3] /= 10 (dataset.py:292)
    out[dst * f + c] = v;
  }
}

template <bool HALF>
__global__ void __launch_bounds__(256) idisc_place_kernel(const void* __restrict__ pts, int64_t p, int f, DiscardEdges e,
                                                          const int32_t* __restrict__ wavebase,
                                                          const DiscardHeader* __restrict__ hdr, float intensity_div,
                                                          float* __restrict__ out, int32_t* __restrict__ lists) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t w = i >> 6;
  const int b = (i < p) ? bin_of(load_x<HALF>(pts, i, f), e) : -1;
  int rank = 0;
  for (int k = 0; k < e.nb; ++k) {
    const unsigned long long m = __ballot(b == k);
    if (b == k) rank = __popcll(m & ((1ULL << lane) - 1ULL));
  }
  if (b < 0) return;
  rank += wavebase[w * kMaxBins + b];
  if (hdr->reduced[b]) lists[hdr->list0[b] + rank] = (int32_t)i;
  else copy_row<HALF>(pts, i, out, (int64_t)hdr->start[b] + rank, f, intensity_div);
}

// thread = output row t of the capacity: rows of reduced bins are fetched through the permutation, rows beyond n_out become
// out-of-range sentinels (fill_tail) so that a consumer can process the whole capacity.
template <bool HALF>
__global__ void __launch_bounds__(256) idisc_emit_kernel(const void* __restrict__ pts, int64_t cap, int f, int nb,
                                                         const DiscardHeader* __restrict__ hdr,
                                                         const int32_t* __restrict__ lists, DiscardPerms perms,
                                                         uint64_t seed, float intensity_div, int fill_tail,
                                                         float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= cap) return;
  if (t >= hdr->n_out) {
    if (fill_tail)
      for (int c = 0; c < f; ++c) out[t * f + c] = (c < 3) ? -3.0e38f : 0.0f;  // far outside any point-cloud range
    return;
  }
  int b = -1;
#pragma unroll
  for (int i = 0; i < kMaxBins; ++i)
    if (i < nb && hdr->kept[i] > 0 && t >= hdr->start[i] && t < (int64_t)hdr->start[i] + hdr->kept[i]) b = i;
  if (b < 0 || !hdr->reduced[b]) return;  // kept-whole bins were placed by idisc_place_kernel
  const int64_t j = t - hdr->start[b];
  const uint64_t n_b = (uint64_t)hdr->count[b];
  uint64_t src;
  if (perms.p[b] != nullptr) src = (uint64_t)perms.p[b][j];
  else src = feistel_perm((uint64_t)j, n_b, feistel_half_bits(n_b), seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(b + 1)));
  if (src >= n_b) src = n_b - 1;  // a malformed injected permutation must not read out of bounds
  copy_row<HALF>(pts, (int64_t)lists[hdr->list0[b] + (int64_t)src], out, t, f, intensity_div);
}

__global__ void __launch_bounds__(256) copy_rows_div_kernel(const float* __restrict__ src, int64_t n, int f,
                                                            float intensity_div, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * f) return;
  float v = src[e];
  if (intensity_div != 0.0f && (e % f) == 3) v = __fdiv_rn(v, intensity_div);
  dst[e] = v;
}

__global__ void idisc_count_out_kernel(const DiscardHeader* hdr, int32_t add, int32_t* n_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = hdr->n_out + add;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static inline size_t discard_ws_bytes(int64_t p) {
  const int64_t nw = cdiv(p > 0 ? p : 1, 64);
  return align256(sizeof(DiscardHeader)) + align256((size_t)nw * kMaxBins * 4) + align256((size_t)(p > 0 ? p : 1) * 4);
}

static int input_discard_impl(const void* points, int is_half, int64_t p, int f, int bin_num, double rate, double max_dis,
                              const int64_t* const* host_perms, uint64_t seed, float intensity_div, int fill_tail,
                              void* ws, size_t ws_bytes, float* out, int32_t* n_out_dev, int32_t n_out_add,
                              hipStream_t st) {
  VC_REQUIRE(p >= 0 && f >= 1 && bin_num >= 1 && bin_num <= kMaxBins, "vc_input_discard: invalid p / f / bin_num (1..16)");
  VC_REQUIRE(rate >= 0.0 && rate < 1.0 && max_dis > 0.0, "vc_input_discard: need 0 <= rate < 1 and max_dis > 0");
  VC_REQUIRE(ws && (out || p == 0) && (points || p == 0), "vc_input_discard: null argument");
  VC_REQUIRE(!is_half || f % 2 == 0, "vc_input_discard: fp16 points need an even feature count");
  if (ws_bytes < discard_ws_bytes(p)) { set_error("vc_input_discard: workspace too small"); return VC_ECAPACITY; }
  DiscardHeader* hdr = (DiscardHeader*)ws;
  int32_t* wavecnt = (int32_t*)((char*)ws + align256(sizeof(DiscardHeader)));
  const int64_t nw = cdiv(p > 0 ? p : 1, 64);
  int32_t* lists = (int32_t*)((char*)wavecnt + align256((size_t)nw * kMaxBins * 4));
  DiscardEdges e;
  e.nb = bin_num;
  const double inter = max_dis / bin_num;
  for (int i = 0; i < kMaxBins; ++i) e.lo[i] = (float)(inter * i);
  DiscardPerms pm;
  for (int i = 0; i < kMaxBins; ++i) pm.p[i] = (host_perms && i < bin_num) ? host_perms[i] : nullptr;
  const double retain = 1.0 - rate;
  const unsigned nblk = (unsigned)cdiv(p > 0 ? p : 1, 256);
  if (p > 0) {
    if (is_half) hipLaunchKernelGGL(idisc_count_kernel<true>, dim3(nblk), dim3(256), 0, st, points, p, f, e, wavecnt);
    else hipLaunchKernelGGL(idisc_count_kernel<false>, dim3(nblk), dim3(256), 0, st, points, p, f, e, wavecnt);
    VC_CHECK_LAUNCH("idisc_count_kernel");
  }
  hipLaunchKernelGGL(idisc_scan_kernel, dim3(1), dim3(1024), 0, st, wavecnt, p > 0 ? nw : 0, p, bin_num, retain, hdr);
  VC_CHECK_LAUNCH("idisc_scan_kernel");
  if (p > 0) {
    if (is_half) {
      hipLaunchKernelGGL(idisc_place_kernel<true>, dim3(nblk), dim3(256), 0, st, points, p, f, e, wavecnt, hdr,
                         intensity_div, out, lists);
      hipLaunchKernelGGL(idisc_emit_kernel<true>, dim3(nblk), dim3(256), 0, st, points, p, f, bin_num, hdr, lists, pm, seed,
                         intensity_div, fill_tail, out);
    } else {
      hipLaunchKernelGGL(idisc_place_kernel<false>, dim3(nblk), dim3(256), 0, st, points, p, f, e, wavecnt, hdr,
                         intensity_div, out, lists);
      hipLaunchKernelGGL(idisc_emit_kernel<false>, dim3(nblk), dim3(256), 0, st, points, p, f, bin_num, hdr, lists, pm,
                         seed, intensity_div, fill_tail, out);
    }
    VC_CHECK_LAUNCH("idisc_place/emit_kernel");
  }
  if (n_out_dev) {
    hipLaunchKernelGGL(idisc_count_out_kernel, dim3(1), dim3(64), 0, st, hdr, n_out_add, n_out_dev);
    VC_CHECK_LAUNCH("idisc_count_out_kernel");
  }
  return VC_OK;
}

}  // namespace vc

using namespace vc;

extern "C" {

size_t vc_input_discard_workspace_bytes(int64_t p) { return p < 0 ? 0 : discard_ws_bytes(p); }

int vc_input_discard(const void* points, int points_are_f16, int64_t p, int f, int bin_num, double rate, double max_dis,
                     const int64_t* const* host_perms, uint64_t seed, void* ws, size_t ws_bytes, float* out,
                     int32_t* n_out_dev, void* stream) {
  VC_REQUIRE(n_out_dev, "vc_input_discard: n_out_dev is null");
  return input_discard_impl(points, points_are_f16, p, f, bin_num, rate, max_dis, host_perms, seed, 0.0f, 0, ws, ws_bytes, out,
                            n_out_dev, 0, (hipStream_t)stream);
}

size_t vc_frontend_workspace_bytes(int64_t p_lidar, int64_t p_virtual, int f, int max_points) {
  if (p_lidar < 0 || p_virtual < 0 || f < 1 || max_points < 1) return 0;
  const int64_t p = p_lidar + p_virtual;
  return align256((size_t)(p > 0 ? p : 1) * f * 4) + align256(discard_ws_bytes(p_virtual)) +
         align256(vc_voxelize_workspace_bytes(p, max_points));
}

int vc_frontend_voxelize_mean(const float* lidar, int64_t p_lidar, const void* virt, int virt_is_f16, int64_t p_virtual,
                              int f, int bin_num, double rate, double max_dis, const int64_t* const* host_perms,
                              uint64_t seed, float intensity_div, const float* host_range, const float* host_vsize,
                              int max_points, int max_voxels, int vfe_max_last, void* ws, size_t ws_bytes, float* features,
                              int32_t* coords, int32_t* num_points, int32_t* n_voxels_dev, int32_t* n_points_dev,
                              void* stream) {
  VC_REQUIRE(p_lidar >= 0 && p_virtual >= 0 && f >= 3 && ws && (lidar || p_lidar == 0) && (virt || p_virtual == 0),
             "vc_frontend_voxelize_mean: null/invalid argument");
  if (ws_bytes < vc_frontend_workspace_bytes(p_lidar, p_virtual, f, max_points)) {
    set_error("vc_frontend_voxelize_mean: workspace too small");
    return VC_ECAPACITY;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t p = p_lidar + p_virtual;
  float* pts = (float*)ws;                                         // fused point list: LiDAR rows first (LIDAR_FIRST)
  char* dws = (char*)ws + align256((size_t)(p > 0 ? p : 1) * f * 4);
  char* vws = dws + align256(discard_ws_bytes(p_virtual));
  if (p_lidar > 0) {
    hipLaunchKernelGGL(copy_rows_div_kernel, dim3((unsigned)cdiv(p_lidar * f, 256)), dim3(256), 0, st, lidar, p_lidar, f,
                       intensity_div, pts);
    VC_CHECK_LAUNCH("copy_rows_div_kernel");
  }
  int rc = input_discard_impl(virt, virt_is_f16, p_virtual, f, bin_num, rate, max_dis, host_perms, seed, intensity_div,
                              /*fill_tail=*/1, dws, discard_ws_bytes(p_virtual), pts + p_lidar * f, n_points_dev,
                              (int32_t)p_lidar, st);
  if (rc != VC_OK) return rc;
  // rows beyond the kept virtual points are out-of-range sentinels: the voxeliser drops them like any point outside the
  // range, and they come AFTER every real point, so first-touch voxel ids and slot order are those of the exact list
  return vc_voxelize_mean(pts, p, f, host_range, host_vsize, max_points, max_voxels, vfe_max_last, vws,
                          vc_voxelize_workspace_bytes(p, max_points), features, coords, num_points, n_voxels_dev, st);
}

}  // extern "C"
