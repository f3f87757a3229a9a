// Depth-aware crack filling of the stage-1 warper (vggt/modules/utils_warp.py:386-706 as run by warp_single_img :954-985), all warped views
// of a camera path in a handful of launches.  HBM-bound 3 x 3 stencils on u8 / f32 images; nothing is copied to the host.
//
// Per view (reference functions in brackets):
//   1. k_cf_stats     min / max of the depth over the valid pixels, number of non-NaN depths            [segment_depth_map :506-518]
//   2. k_cf_segment   depth segment 0..4 of every valid pixel (boundaries = linspace(min, max, 6) in f64), outlier test of the pixel inside
//                     its own segment (3 x 3 count incl. the centre < min_neighbors, BORDER_REFLECT_101), per-segment "has outliers" flag
//                                                                                                        [segment_depth_map, fill_segment_cracks :586-600]
//   3. k_cf_fillmask  for every segment that lost an outlier: 3 x 3 closing of its cleaned mask, the pixels newly covered with >=
//                     min_valid_neighbors valid 8-neighbours are filled (bit s of the pixel's fill mask); per-segment depth sum / count for
//                     the far-to-near order of the merge                                                [fill_small_cracks :403-430, merge :650-662]
//   4. k_cf_merge     far-to-near merge: every pixel takes colour / depth from the NEAREST segment that covers it -- its own value where the
//                     segment holds the pixel, the mean of the segment's valid 8-neighbours (colour) / of the valid depths around it (depth,
//                     BORDER_REFLECT) where the segment filled it                                       [fill_small_cracks, vectorized_depth_estimation :539-564, merge :664-676]
// OpenCV semantics restated as in oracle/crackfill.py (filter2D = correlation with BORDER_REFLECT_101, morphologyEx ignores the border);
// parity with a real cv2 is unpinned (opencv-python is not in the image).
#include "common.h"

using namespace wf;

namespace {

constexpr int NSEG = 5;

struct CFArgs {
  const uint8_t* img;   // [n, H, W, 3] splatted views
  const uint8_t* mask;  // [n, H, W]
  const float* depth;   // [n, H, W], NaN = empty
  uint8_t* out_img;
  uint8_t* out_mask;
  float* out_depth;
  uint8_t* seg;        // [n, H, W]: segment 0..4, 255 = not valid; bit 7 clear + bit 6 set = outlier of its segment (seg | 64)
  uint8_t* fill;       // [n, H, W]: bit s = filled by segment s
  unsigned* stats;     // [n][4]: ordered(min), ordered(max), count of non-NaN depths, count of valid pixels
  unsigned* segflag;   // [n][NSEG]: segment has outliers
  double* segsum;      // [n][NSEG][2]: sum of depths, count (for the far-to-near order)
  int n, H, W;
  int min_neighbors, min_valid_neighbors, num_segments;
};

__device__ __forceinline__ unsigned ordered(float f) {  // monotone float -> unsigned
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__device__ __forceinline__ int refl101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }

__global__ void k_cf_stats(CFArgs a) {
  const int f = blockIdx.y;
  const size_t np = (size_t)a.H * a.W;
  unsigned lo = 0xffffffffu, hi = 0u, cnt = 0u, nv = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a.depth[f * np + i];
    const bool nn = d == d;
    cnt += nn;
    if (a.mask[f * np + i] && nn) {
      const unsigned o = ordered(d);
      lo = min(lo, o);
      hi = max(hi, o);
      ++nv;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
    cnt += (unsigned)__shfl_xor((int)cnt, o, 64);
    nv += (unsigned)__shfl_xor((int)nv, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(a.stats + f * 4 + 0, lo);
    atomicMax(a.stats + f * 4 + 1, hi);
    atomicAdd(a.stats + f * 4 + 2, cnt);
    atomicAdd(a.stats + f * 4 + 3, nv);
  }
}

// segment of a depth value: boundaries b_i = lo + i * (hi - lo) / S in f64 (np.linspace), last one = hi; [b_i, b_{i+1}) and [b_{S-1}, b_S]
__device__ __forceinline__ int segment_of(float d, double lo, double hi, int S) {
  if (!(d == d)) return 255;
  if (lo == hi) return 0;
  const double x = (double)d, step = (hi - lo) / S;
  for (int i = 0; i < S; ++i) {
    const double b0 = lo + i * step, b1 = i == S - 1 ? hi : lo + (i + 1) * step;
    if (x >= b0 && (i == S - 1 ? x <= b1 : x < b1)) return i;
  }
  return 255;
}

__global__ void k_cf_segid(CFArgs a) {
  const int f = blockIdx.y;
  const size_t np = (size_t)a.H * a.W;
  if (a.stats[f * 4 + 3] == 0) return;
  const double lo = unordered(a.stats[f * 4 + 0]), hi = unordered(a.stats[f * 4 + 1]);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x)
    a.seg[f * np + i] = a.mask[f * np + i] ? (uint8_t)segment_of(a.depth[f * np + i], lo, hi, a.num_segments) : (uint8_t)255;
}

__global__ void k_cf_outlier(CFArgs a) {
  const int f = blockIdx.y;
  const size_t np = (size_t)a.H * a.W;
  if (a.stats[f * 4 + 3] == 0) return;
  const uint8_t* sg = a.seg + f * np;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / a.W), x = (int)(i % a.W);
    const int s = sg[i] & 63;
    if (sg[i] == 255) continue;
    int c = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const uint8_t q = sg[(size_t)refl101(y + dy, a.H) * a.W + refl101(x + dx, a.W)];
        c += (q != 255 && (q & 63) == s);
      }
    if (c < a.min_neighbors) {
      a.fill[f * np + i] = 128;  // marker: outlier of its own segment (turned into seg | 64 by the next kernel; fill is reset there)
      atomicOr(a.segflag + f * NSEG + s, 1u);
    }
  }
}

__global__ void k_cf_mark(CFArgs a) {  // seg |= 64 for outliers (a separate pass: the outlier test reads its neighbours' plain ids)
  const int f = blockIdx.y;
  const size_t np = (size_t)a.H * a.W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x) {
    if (a.fill[f * np + i] == 128) a.seg[f * np + i] |= 64;
    a.fill[f * np + i] = 0;
  }
}

// cleaned mask of segment s at (y, x); outside the image -> 0
__device__ __forceinline__ bool cleaned(const uint8_t* sg, int y, int x, int H, int W, int s) {
  if (y < 0 || y >= H || x < 0 || x >= W) return false;
  return sg[(size_t)y * W + x] == (uint8_t)s;  // 255 = invalid, s | 64 = outlier: both differ from s
}
__device__ __forceinline__ bool cleaned101(const uint8_t* sg, int y, int x, int H, int W, int s) {
  return sg[(size_t)refl101(y, H) * W + refl101(x, W)] == (uint8_t)s;
}

// is pixel (y, x) filled by segment s?  closing (border ignored) newly covers it and it has >= mvn cleaned 8-neighbours (reflect 101)
__device__ __forceinline__ bool fills(const uint8_t* sg, int y, int x, int H, int W, int s, int mvn) {
  if (cleaned(sg, y, x, H, W, s)) return false;
  // erode(dilate(m)): every in-image 3 x 3 neighbour q of p must have a cleaned pixel in ITS in-image 3 x 3 neighbourhood
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int qy = y + dy, qx = x + dx;
      if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
      bool any = false;
      for (int ey = -1; ey <= 1 && !any; ++ey)
        for (int ex = -1; ex <= 1 && !any; ++ex) any = cleaned(sg, qy + ey, qx + ex, H, W, s);
      if (!any) return false;
    }
  int c = 0;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx)
      if (dy || dx) c += cleaned101(sg, y + dy, x + dx, H, W, s);
  return c >= mvn;
}

__device__ __forceinline__ float est_depth(const float* dp, int y, int x, int H, int W) {  // :539-564, BORDER_REFLECT, centre excluded
  float s = 0.f, c = 0.f;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      if (!dy && !dx) continue;
      const float d = dp[(size_t)refl(y + dy, H) * W + refl(x + dx, W)];
      if (d == d) {
        s += d;
        c += 1.f;
      }
    }
  return s / fmaxf(c, 1e-6f);
}

__global__ void k_cf_fillmask(CFArgs a) {
  const int f = blockIdx.y;
  const size_t np = (size_t)a.H * a.W;
  if (a.stats[f * 4 + 3] == 0) return;
  const uint8_t* sg = a.seg + f * np;
  const float* dp = a.depth + f * np;
  unsigned flags = 0;
  for (int s = 0; s < NSEG; ++s) flags |= (a.segflag[f * NSEG + s] ? 1u : 0u) << s;
  double sum[NSEG] = {0, 0, 0, 0, 0}, cnt[NSEG] = {0, 0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / a.W), x = (int)(i % a.W);
    unsigned fm = 0;
    for (int s = 0; s < NSEG; ++s) {
      if (sg[i] == (uint8_t)s) {  // the segment holds the pixel (not an outlier)
        sum[s] += dp[i];
        cnt[s] += 1;
      } else if (((flags >> s) & 1u) && fills(sg, y, x, a.H, a.W, s, a.min_valid_neighbors)) {
        fm |= 1u << s;
        sum[s] += est_depth(dp, y, x, a.H, a.W);
        cnt[s] += 1;
      }
    }
    a.fill[f * np + i] = (uint8_t)fm;
  }
  for (int s = 0; s < NSEG; ++s) {
    const double ts = wave_sum_d(sum[s]), tc = wave_sum_d(cnt[s]);
    if ((threadIdx.x & 63) == 0 && tc > 0) {
      atomicAdd(a.segsum + (f * NSEG + s) * 2, ts);
      atomicAdd(a.segsum + (f * NSEG + s) * 2 + 1, tc);
    }
  }
}

__global__ void k_cf_merge(CFArgs a) {
  const int f = blockIdx.y;
  const size_t np = (size_t)a.H * a.W;
  const uint8_t* sg = a.seg + f * np;
  const uint8_t* im = a.img + f * np * 3;
  const float* dp = a.depth + f * np;
  // far-to-near order: segments by mean depth, descending; the LAST writer (nearest) wins -> walk ascending and stop at the first cover
  int order[NSEG];
  double avg[NSEG];
  for (int s = 0; s < NSEG; ++s) {
    order[s] = s;
    const double c = a.segsum[(f * NSEG + s) * 2 + 1];
    avg[s] = c > 0 ? (double)(float)(a.segsum[(f * NSEG + s) * 2] / c) : 1e300;
  }
  for (int i = 1; i < NSEG; ++i)  // insertion sort, ascending mean depth; ties keep the later segment first (stable descending sort reversed)
    for (int j = i; j > 0 && (avg[order[j]] < avg[order[j - 1]] || (avg[order[j]] == avg[order[j - 1]] && order[j] > order[j - 1])); --j) {
      const int t = order[j];
      order[j] = order[j - 1];
      order[j - 1] = t;
    }
  const bool have = a.stats[f * 4 + 3] != 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / a.W), x = (int)(i % a.W);
    uint8_t r = 0, g = 0, b = 0, m = 0;
    float d = __uint_as_float(0x7fc00000u);
    if (have) {
      const unsigned fm = a.fill[f * np + i];
      for (int k = 0; k < NSEG; ++k) {
        const int s = order[k];
        if (a.segsum[(f * NSEG + s) * 2 + 1] <= 0) continue;
        if (sg[i] == (uint8_t)s) {
          r = im[i * 3];
          g = im[i * 3 + 1];
          b = im[i * 3 + 2];
          d = dp[i];
          m = 1;
          break;
        }
        if ((fm >> s) & 1u) {
          float acc[3] = {0.f, 0.f, 0.f}, c = 0.f;
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              if (!dy && !dx) continue;
              const size_t q = (size_t)refl101(y + dy, a.H) * a.W + refl101(x + dx, a.W);
              if (sg[q] == (uint8_t)s) {
                acc[0] += (float)im[q * 3] / 255.0f;
                acc[1] += (float)im[q * 3 + 1] / 255.0f;
                acc[2] += (float)im[q * 3 + 2] / 255.0f;
                c += 1.f;
              }
            }
          const float sc = fmaxf(c, 1e-6f);
          r = (uint8_t)(int)(acc[0] / sc * 255.0f);
          g = (uint8_t)(int)(acc[1] / sc * 255.0f);
          b = (uint8_t)(int)(acc[2] / sc * 255.0f);
          d = est_depth(dp, y, x, a.H, a.W);
          m = 1;
          break;
        }
      }
    }
    a.out_img[(f * np + i) * 3] = r;
    a.out_img[(f * np + i) * 3 + 1] = g;
    a.out_img[(f * np + i) * 3 + 2] = b;
    a.out_mask[f * np + i] = m;
    a.out_depth[f * np + i] = d;
  }
}

}  // namespace

extern "C" size_t wf_crack_fill_workspace_bytes(int n, int H, int W) {
  if (n <= 0 || H <= 0 || W <= 0) return 0;
  const size_t np = (size_t)n * H * W;
  return 2 * np + (size_t)n * (4 + NSEG) * sizeof(unsigned) + 64 + (size_t)n * NSEG * 2 * sizeof(double);
}

extern "C" int wf_crack_fill(const void* img, const void* mask, const float* depth, void* out_img, void* out_mask, float* out_depth, int n,
                             int H, int W, int min_neighbors, int min_valid_neighbors, int num_segments, void* workspace, void* stream) {
  WF_CHECK_ARG(img && mask && depth && out_img && out_mask && out_depth && workspace, "wf_crack_fill: null pointer");
  WF_CHECK_ARG(n > 0 && H >= 2 && W >= 2, "wf_crack_fill: bad sizes n=%d H=%d W=%d", n, H, W);
  WF_CHECK_ARG(num_segments >= 1 && num_segments <= NSEG, "wf_crack_fill: num_segments (%d) must be 1..%d", num_segments, NSEG);
  hipStream_t st = (hipStream_t)stream;
  const size_t np = (size_t)n * H * W;
  unsigned char* ws = (unsigned char*)workspace;
  CFArgs a;
  a.img = (const uint8_t*)img; a.mask = (const uint8_t*)mask; a.depth = depth;
  a.out_img = (uint8_t*)out_img; a.out_mask = (uint8_t*)out_mask; a.out_depth = out_depth;
  a.seg = ws;
  a.fill = ws + np;
  size_t off = (2 * np + 63) / 64 * 64;
  a.segsum = (double*)(ws + off);
  off += (size_t)n * NSEG * 2 * sizeof(double);
  a.stats = (unsigned*)(ws + off);
  a.segflag = a.stats + (size_t)n * 4;
  a.n = n; a.H = H; a.W = W;
  a.min_neighbors = min_neighbors; a.min_valid_neighbors = min_valid_neighbors; a.num_segments = num_segments;
  // seg = 255, fill = 0, sums / flags = 0, stats = {~0, 0, 0, 0}
  if (hipMemsetAsync(a.seg, 0xff, np, st) != hipSuccess || hipMemsetAsync(a.fill, 0, np, st) != hipSuccess ||
      hipMemsetAsync(a.segsum, 0, (size_t)n * NSEG * 2 * sizeof(double) + (size_t)n * (4 + NSEG) * sizeof(unsigned), st) != hipSuccess)
    return check_hip(hipGetLastError(), "wf_crack_fill: memset");
  for (int f = 0; f < n; ++f)
    if (hipMemsetAsync(a.stats + f * 4, 0xff, sizeof(unsigned), st) != hipSuccess) return check_hip(hipGetLastError(), "wf_crack_fill: memset");
  const dim3 grid((unsigned)std::min<size_t>(((size_t)H * W + 255) / 256, 1024), (unsigned)n), blk(256);
  hipLaunchKernelGGL(k_cf_stats, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_cf_segid, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_cf_outlier, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_cf_mark, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_cf_fillmask, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_cf_merge, grid, blk, 0, st, a);
  WF_LAUNCH_CHECK("wf_crack_fill");
  return WF_OK;
}
