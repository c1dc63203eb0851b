// Rotated-box BEV overlap / IoU and NMS for gfx950.  Reference path: pcdet/ops/iou3d_nms --
// iou3d_nms_utils.py:31-135 (boxes_iou_bev, boxes_iou3d_gpu, nms_gpu, nms_normal_gpu) over src/iou3d_nms_kernel.cu:30-414 and
// the host loop of src/iou3d_nms.cpp:98-187.
//
// The per-pair geometry follows the reference's published algorithm step by step, because its quirks are part of the result
// (corners count as "inside" up to a 1e-2 margin, edge pairs that merely touch do not intersect, the polygon is ordered by the
// angle around the mean of its vertices): oriented corners -> edge x edge intersections -> contained corners -> angular order ->
// shoelace area.  What is different is everything around it:
//   * overlap / IoU matrices: one lane per pair, the "b" box of a lane is loaded once per block column, and boxes_iou3d is ONE
//     launch (the reference multiplies the BEV overlap with the height overlap in five torch kernels);
//   * NMS mask: a wave per (64 rows x 64 columns) tile, only tiles on or above the diagonal (the reference launches the full
//     square and never reads the lower half); lane j keeps column box j in registers, the row box is broadcast, and the 64-bit
//     suppression word of a row IS the wave ballot of (IoU > thresh);
//   * NMS selection: on the device, one wave.  The reference copies the N x N/64 mask to the host and walks it there
//     (iou3d_nms.cpp:125-150); here each 64-box block first resolves itself from its diagonal words in registers
//     (v_readlane chain, no memory), then the kept rows' mask words are OR-ed into the lanes' removal words with independent,
//     coalesced loads.  The kept indices and their count stay on the device (capacity + device-side count).
#include "common.h"

namespace vc {

namespace {

struct P2 {
  float x, y;
};

__device__ __forceinline__ float cross3(const P2& p1, const P2& p2, const P2& p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

// iou3d_nms_kernel.cu:62-72: point inside the box, with the reference's 1e-2 margin
__device__ __forceinline__ bool inside_with_margin(const float* box, const P2& p) {
  const float c = cosf(-box[6]), s = sinf(-box[6]);
  const float dx = p.x - box[0], dy = p.y - box[1];
  const float rx = dx * c + dy * (-s), ry = dx * s + dy * c;
  return fabsf(rx) < box[3] / 2 + 1e-2f && fabsf(ry) < box[4] / 2 + 1e-2f;
}

// iou3d_nms_kernel.cu:74-106: proper crossing of segments (p0,p1) x (q0,q1)
__device__ __forceinline__ bool seg_intersection(const P2& p1, const P2& p0, const P2& q1, const P2& q0, P2& ans) {
  const bool boxes_meet = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!boxes_meet) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float d = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / d;
    ans.y = (a1 * c0 - a0 * c1) / d;
  }
  return true;
}

__device__ __forceinline__ void oriented_corners(const float* box, P2* c) {  // 5 entries, c[4] = c[0]
  const float hx = box[3] / 2, hy = box[4] / 2;
  const float cs = cosf(box[6]), sn = sinf(box[6]);
  const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // the reference rotates the axis-aligned corner (box[0] +- hx, box[1] +- hy) around the centre: same expression tree
    const float px = box[0] + lx[k], py = box[1] + ly[k];
    c[k].x = (px - box[0]) * cs + (py - box[1]) * (-sn) + box[0];
    c[k].y = (px - box[0]) * sn + (py - box[1]) * cs + box[1];
  }
  c[4] = c[0];
}

// iou3d_nms_kernel.cu:127-225
__device__ float rbox_overlap(const float* a, const float* b) {
  {  // far-apart boxes: no edge can cross and no corner can fall inside the other box even with the reference's 1e-2 margin, so
     // the reference's polygon is empty and its area exactly 0 -- skip the clipping (most pairs of a proposal set)
    const float dx = a[0] - b[0], dy = a[1] - b[1];
    const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]), rb = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
    const float reach = ra + rb + 0.05f;
    if (dx * dx + dy * dy > reach * reach) return 0.f;
  }
  P2 ca[5], cb[5];
  oriented_corners(a, ca);
  oriented_corners(b, cb);
  P2 pts[16];
  float sx = 0.f, sy = 0.f;
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], x)) {
        sx += x.x; sy += x.y;
        pts[cnt++] = x;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (inside_with_margin(a, cb[k])) { sx += cb[k].x; sy += cb[k].y; pts[cnt++] = cb[k]; }
    if (inside_with_margin(b, ca[k])) { sx += ca[k].x; sy += ca[k].y; pts[cnt++] = ca[k]; }
  }
  if (cnt < 3) return 0.f;  // the reference's fan over fewer than three points sums nothing (cnt <= 1) or one zero cross product (cnt = 2)
  const float mx = sx / cnt, my = sy / cnt;
  // angular order around the mean: the reference bubble-sorts with atan2 evaluated inside the comparator (a stable ascending
  // sort); a stable insertion sort on the angles computed once gives the same permutation
  float ang[16];
  for (int k = 0; k < cnt; ++k) ang[k] = atan2f(pts[k].y - my, pts[k].x - mx);
  for (int k = 1; k < cnt; ++k) {
    const P2 p = pts[k];
    const float t = ang[k];
    int m = k - 1;
    while (m >= 0 && ang[m] > t) { pts[m + 1] = pts[m]; ang[m + 1] = ang[m]; --m; }
    pts[m + 1] = p; ang[m + 1] = t;
  }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    const float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    const float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float rbox_iou_bev(const float* a, const float* b) {
  const float sa = a[3] * a[4], sb = b[3] * b[4];
  const float ov = rbox_overlap(a, b);
  return ov / fmaxf(sa + sb - ov, 1e-8f);
}

// iou3d_nms_kernel.cu:321-331: axis-aligned BEV IoU (heading ignored)
__device__ __forceinline__ float abox_iou_bev(const float* a, const float* b) {
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  const float inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

}  // namespace

// MODE 0: BEV overlap area, 1: BEV IoU, 2: 3-D IoU (iou3d_nms_utils.py:67-99)
// block (64, 4): lane = column box b (kept in registers), 4 row boxes a per block
template <int MODE>
__global__ void __launch_bounds__(256) pair_matrix_kernel(const float* __restrict__ boxes_a, int n_a,
                                                          const float* __restrict__ boxes_b, int n_b, float* __restrict__ out) {
  const int jb = blockIdx.x * 64 + threadIdx.x;
  const int ia0 = (blockIdx.y * 4 + threadIdx.y) * 16;
  float b[7];
  if (jb < n_b) {
#pragma unroll
    for (int c = 0; c < 7; ++c) b[c] = boxes_b[(int64_t)jb * 7 + c];
  }
  for (int r = 0; r < 16; ++r) {
    const int ia = ia0 + r;
    if (ia >= n_a) break;  // uniform per wave
    float a[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) a[c] = boxes_a[(int64_t)ia * 7 + c];  // same address for the whole wave: one broadcast load
    if (jb >= n_b) continue;
    float v;
    if (MODE == 1) {
      v = rbox_iou_bev(a, b);
    } else {
      v = rbox_overlap(a, b);
      if (MODE == 2) {
        const float hmax = fminf(a[2] + a[5] / 2, b[2] + b[5] / 2), hmin = fmaxf(a[2] - a[5] / 2, b[2] - b[5] / 2);
        const float ov3 = v * fmaxf(hmax - hmin, 0.f);
        const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
        v = ov3 / fmaxf(va + vb - ov3, 1e-6f);
      }
    }
    out[(int64_t)ia * n_b + jb] = v;
  }
}

// One wave per (row block rb, column block cb >= rb).  mask[row * n_blocks + cb] bit j <=> IoU(row, 64*cb + j) > thresh and
// (on the diagonal) j > row's position.
template <bool ROTATED>
__global__ void __launch_bounds__(64) nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh, int n_blocks,
                                                      unsigned long long* __restrict__ mask) {
  // upper-triangular tile index -> (rb, cb)
  int t = blockIdx.x, rb = 0;
  {  // rows of the triangle have n_blocks, n_blocks - 1, ... tiles
    int rem = n_blocks;
    while (t >= rem) { t -= rem; --rem; ++rb; }
  }
  const int cb = rb + t;
  const int lane = threadIdx.x;
  __shared__ float s_row[64 * 7];
  const int row0 = rb * 64, col = cb * 64 + lane;
  for (int e = lane; e < 64 * 7; e += 64) s_row[e] = (row0 * 7 + e < n * 7) ? boxes[(int64_t)row0 * 7 + e] : 0.f;
  float b[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) b[c] = (col < n) ? boxes[(int64_t)col * 7 + c] : 0.f;
  __syncthreads();
  unsigned long long mine = 0ULL;
  const int rows = min(64, n - row0);
  for (int r = 0; r < rows; ++r) {
    bool hit = false;
    if (col < n && (cb != rb || lane > r)) {
      const float iou = ROTATED ? rbox_iou_bev(&s_row[r * 7], b) : abox_iou_bev(&s_row[r * 7], b);
      hit = iou > thresh;
    }
    const unsigned long long bits = __ballot(hit);
    if (lane == r) mine = bits;
  }
  if (lane < rows) mask[(int64_t)(row0 + lane) * n_blocks + cb] = mine;
}

static constexpr int kNmsMaxWordsPerLane = 16;  // n <= 64 * 64 * 16 = 65536 boxes

// One wave.  Lane l owns the removal words w = l, l + 64, ... (WPL of them).
__global__ void __launch_bounds__(64) nms_select_kernel(const unsigned long long* __restrict__ mask, int n, int n_blocks,
                                                        int64_t* __restrict__ keep, int64_t* __restrict__ num_out) {
  const int lane = threadIdx.x;
  unsigned long long remv[kNmsMaxWordsPerLane];
#pragma unroll
  for (int s = 0; s < kNmsMaxWordsPerLane; ++s) remv[s] = 0ULL;
  int64_t cnt = 0;
  for (int blk = 0; blk < n_blocks; ++blk) {
    const int row0 = blk * 64, rows = min(64, n - row0);
    // the block's own removal word (held by lane blk & 63, slot blk >> 6) and its diagonal mask words (lane r: row r)
    unsigned long long cur_l = 0ULL;
#pragma unroll
    for (int s = 0; s < kNmsMaxWordsPerLane; ++s)
      if (s == (blk >> 6)) cur_l = remv[s];
    unsigned long long cur = __shfl(cur_l, blk & 63, 64);
    const unsigned long long diag = (lane < rows) ? mask[(int64_t)(row0 + lane) * n_blocks + blk] : 0ULL;
    unsigned long long kept = 0ULL;
    for (int r = 0; r < rows; ++r) {
      const unsigned long long d = __shfl(diag, r, 64);
      if (!((cur >> r) & 1ULL)) {
        kept |= 1ULL << r;
        cur |= d;
      }
    }
    // kept rows -> output (lane r writes its own index at its rank), in ascending order = descending score
    if ((kept >> lane) & 1ULL) keep[cnt + __popcll(kept & ((1ULL << lane) - 1ULL))] = row0 + lane;
    cnt += __popcll(kept);
    // suppress everything the kept rows overlap in the blocks to the right: independent coalesced loads
    unsigned long long k2 = kept;
    while (k2) {
      const int r = __ffsll((long long)k2) - 1;
      k2 &= k2 - 1;
      const unsigned long long* mrow = mask + (int64_t)(row0 + r) * n_blocks;
#pragma unroll
      for (int s = 0; s < kNmsMaxWordsPerLane; ++s) {
        const int w = lane + 64 * s;
        if (w > blk && w < n_blocks) remv[s] |= mrow[w];
      }
    }
  }
  if (lane == 0) *num_out = cnt;
}

}  // namespace vc

using namespace vc;

extern "C" {

static int launch_pairs(int mode, const float* boxes_a, int64_t n_a, const float* boxes_b, int64_t n_b, float* out, void* stream,
                        const char* who) {
  VC_REQUIRE(n_a >= 0 && n_b >= 0 && n_a < (1LL << 31) && n_b < (1LL << 31), "%s: invalid box count", who);
  if (n_a == 0 || n_b == 0) return VC_OK;
  VC_REQUIRE(boxes_a && boxes_b && out, "%s: null argument", who);
  const dim3 grid((unsigned)cdiv(n_b, 64), (unsigned)cdiv(n_a, 64)), block(64, 4);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(pair_matrix_kernel<0>, grid, block, 0, st, boxes_a, (int)n_a, boxes_b, (int)n_b, out);
  else if (mode == 1) hipLaunchKernelGGL(pair_matrix_kernel<1>, grid, block, 0, st, boxes_a, (int)n_a, boxes_b, (int)n_b, out);
  else hipLaunchKernelGGL(pair_matrix_kernel<2>, grid, block, 0, st, boxes_a, (int)n_a, boxes_b, (int)n_b, out);
  VC_CHECK_LAUNCH("pair_matrix_kernel");
  return VC_OK;
}

int vc_boxes_overlap_bev(const float* boxes_a, int64_t n_a, const float* boxes_b, int64_t n_b, float* overlap, void* stream) {
  return launch_pairs(0, boxes_a, n_a, boxes_b, n_b, overlap, stream, "vc_boxes_overlap_bev");
}

int vc_boxes_iou_bev(const float* boxes_a, int64_t n_a, const float* boxes_b, int64_t n_b, float* iou, void* stream) {
  return launch_pairs(1, boxes_a, n_a, boxes_b, n_b, iou, stream, "vc_boxes_iou_bev");
}

int vc_boxes_iou3d(const float* boxes_a, int64_t n_a, const float* boxes_b, int64_t n_b, float* iou, void* stream) {
  return launch_pairs(2, boxes_a, n_a, boxes_b, n_b, iou, stream, "vc_boxes_iou3d");
}

size_t vc_nms_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const int64_t nb = cdiv(n, 64);
  return (size_t)n * (size_t)nb * sizeof(unsigned long long) + 256;
}

int vc_nms(const float* boxes, int64_t n, float thresh, int rotated, int64_t* keep, int64_t* num_out, void* ws, size_t ws_bytes,
           void* stream) {
  VC_REQUIRE(n >= 0 && n <= 64 * 64 * kNmsMaxWordsPerLane, "vc_nms: %lld boxes (supported: up to %d)", (long long)n,
             64 * 64 * kNmsMaxWordsPerLane);
  VC_REQUIRE(num_out && (n == 0 || (boxes && keep && ws)), "vc_nms: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    VC_CHECK_HIP(hipMemsetAsync(num_out, 0, sizeof(int64_t), st));
    return VC_OK;
  }
  VC_REQUIRE(ws_bytes >= vc_nms_workspace_bytes(n), "vc_nms: workspace too small");
  const int nb = (int)cdiv(n, 64);
  unsigned long long* mask = (unsigned long long*)ws;
  const unsigned tiles = (unsigned)((int64_t)nb * (nb + 1) / 2);
  if (rotated) hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(tiles), dim3(64), 0, st, boxes, (int)n, thresh, nb, mask);
  else hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(tiles), dim3(64), 0, st, boxes, (int)n, thresh, nb, mask);
  VC_CHECK_LAUNCH("nms_mask_kernel");
  hipLaunchKernelGGL(nms_select_kernel, dim3(1), dim3(64), 0, st, mask, (int)n, nb, keep, num_out);
  VC_CHECK_LAUNCH("nms_select_kernel");
  return VC_OK;
}

}  // extern "C"
