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

// ------------------------------------------------------------------------------------------------------------------------------------
// fill_small_cracks in full (utils_warp.py:386-455), the path warp_single_img takes for a view with <= 100 splatted depths (:973-981):
//   step 1 (parallel)  3 x 3 closing of the mask; a newly covered pixel with >= min_valid_neighbors valid 8-neighbours takes their mean;
//   step 2 (only with a confidence map and when step 1 filled fewer than half of the holes): the 4-connected hole components of
//          <= min(max_crack_size, 4) pixels, in scipy.ndimage.label order (raster order of their first pixel), their pixels in np.where
//          (raster) order; a pixel is filled with the mean of the valid 3 x 3 neighbours whose ORIGINAL depth (the source view's map, indexed
//          at the target pixel: the reference's own indexing) is within depth_threshold of its own, if >= min_valid_neighbors of them
//          qualify; every fill is visible to the pixels after it -- a sequential rule, so ONE lane walks the (few) small components.
// The image is carried as float32 (u8 / 255) through both steps and quantised once at the end, as the reference does.
// ------------------------------------------------------------------------------------------------------------------------------------
struct FSArgs {
  const uint8_t* img;    // [H, W, 3]
  const uint8_t* mask;   // [H, W]
  const float* odepth;   // [H, W] original (source-view) depth
  uint8_t* out_img;
  uint8_t* out_mask;
  float* fimg;           // [H, W, 3] f32 working image
  uint8_t* small;        // [H, W]: bit 0 = pixel of a small hole component, bit 1 = its first pixel in raster order
  unsigned* cnt;         // [0] holes of the input mask, [1] pixels filled by step 1
  int H, W, mvn, max_crack, use_depth;
  float thr;
};

__device__ __forceinline__ bool fs_valid(const uint8_t* m, int y, int x, int H, int W) {
  return y >= 0 && y < H && x >= 0 && x < W && m[(size_t)y * W + x] != 0;
}

__global__ void k_fs_step1(FSArgs a) {
  const size_t np = (size_t)a.H * a.W;
  unsigned holes = 0, morph = 0;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / a.W), x = (int)(p % a.W);
    float v[3];
    for (int c = 0; c < 3; ++c) v[c] = (float)a.img[p * 3 + c] / 255.0f;
    uint8_t mo = a.mask[p];
    if (mo == 0) {
      ++holes;
      // closing (border ignored by dilate and erode): every in-image 3 x 3 neighbour has a valid pixel in its own in-image neighbourhood
      bool closed = true;
      for (int dy = -1; dy <= 1 && closed; ++dy)
        for (int dx = -1; dx <= 1 && closed; ++dx) {
          const int qy = y + dy, qx = x + dx;
          if (qy < 0 || qy >= a.H || qx < 0 || qx >= a.W) continue;
          bool any = false;
          for (int ey = -1; ey <= 1 && !any; ++ey)
            for (int ex = -1; ex <= 1 && !any; ++ex) any = fs_valid(a.mask, qy + ey, qx + ex, a.H, a.W);
          closed = any;
        }
      if (closed) {
        // cv2.filter2D with the 8-neighbour kernel, BORDER_REFLECT_101; sums in double, stored as float (as ndimage.correlate does)
        double cnt = 0.0, s[3] = {0.0, 0.0, 0.0};
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            if (!dy && !dx) continue;
            const size_t q = (size_t)refl101(y + dy, a.H) * a.W + refl101(x + dx, a.W);
            if (a.mask[q]) {
              cnt += 1.0;
              for (int c = 0; c < 3; ++c) s[c] += (double)((float)a.img[q * 3 + c] / 255.0f);
            }
          }
        if ((float)cnt >= (float)a.mvn) {
          const float safe = fmaxf((float)cnt, 1e-6f);
          for (int c = 0; c < 3; ++c) v[c] = (float)s[c] / safe;
          mo = 1;
          ++morph;
        }
      }
    }
    for (int c = 0; c < 3; ++c) a.fimg[p * 3 + c] = v[c];
    a.out_mask[p] = mo;
    a.small[p] = 0;
  }
  if (holes) atomicAdd(&a.cnt[0], holes);
  if (morph) atomicAdd(&a.cnt[1], morph);
}

// the 4-connected component of the hole pixel p0 in mask m, explored up to `cap` pixels: returns its size (cap + 1 = larger) and its pixels
__device__ __forceinline__ int fs_component(const uint8_t* m, int H, int W, int p0, int* px, int cap) {
  int n = 1;
  px[0] = p0;
  for (int i = 0; i < n && n <= cap; ++i) {
    const int y = px[i] / W, x = px[i] % W;
    const int ny[4] = {y - 1, y + 1, y, y}, nx[4] = {x, x, x - 1, x + 1};
    for (int k = 0; k < 4 && n <= cap; ++k) {
      if (ny[k] < 0 || ny[k] >= H || nx[k] < 0 || nx[k] >= W) continue;
      const int q = ny[k] * W + nx[k];
      if (m[q]) continue;
      bool seen = false;
      for (int j = 0; j < n; ++j) seen |= px[j] == q;
      if (!seen) px[n++] = q;
    }
  }
  return n;
}

__global__ void k_fs_small(FSArgs a) {
  if (!a.use_depth || !((float)a.cnt[1] < (float)a.cnt[0] * 0.5f)) return;  // :433
  const int lim = a.max_crack < 4 ? a.max_crack : 4;
  const size_t np = (size_t)a.H * a.W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (size_t)gridDim.x * blockDim.x) {
    if (a.out_mask[p]) continue;
    int px[6];
    const int n = fs_component(a.out_mask, a.H, a.W, (int)p, px, 4);
    if (n > lim) continue;
    int first = px[0];
    for (int j = 1; j < n; ++j) first = min(first, px[j]);
    a.small[p] = (uint8_t)(1 | (first == (int)p ? 2 : 0));
  }
}

__global__ void k_fs_step2(FSArgs a) {  // one lane: the rule is sequential; the components it visits are few and tiny
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!a.use_depth || !((float)a.cnt[1] < (float)a.cnt[0] * 0.5f)) return;
  const int np = a.H * a.W;
  for (int p = 0; p < np; ++p) {
    if (!(a.small[p] & 2)) continue;
    int px[6];
    const int n = fs_component(a.out_mask, a.H, a.W, p, px, 4);  // (unchanged since k_fs_small: fills never merge or split hole components' pixels of OTHER components)
    for (int i = 1; i < n; ++i) {  // np.where order = raster order
      const int key = px[i];
      int j = i - 1;
      while (j >= 0 && px[j] > key) { px[j + 1] = px[j]; --j; }
      px[j + 1] = key;
    }
    for (int i = 0; i < n; ++i) {
      const int y = px[i] / a.W, x = px[i] % a.W;
      if (a.out_mask[px[i]]) continue;
      const float dc = a.odepth[px[i]];
      int nvalid = 0, nok = 0;
      float s[3] = {0.f, 0.f, 0.f};
      for (int yy = max(0, y - 1); yy < min(a.H, y + 2); ++yy)
        for (int xx = max(0, x - 1); xx < min(a.W, x + 2); ++xx) {
          const size_t q = (size_t)yy * a.W + xx;
          if (!a.out_mask[q]) continue;
          ++nvalid;
          if (fabsf(a.odepth[q] - dc) <= a.thr) {
            ++nok;
            for (int c = 0; c < 3; ++c) s[c] += a.fimg[q * 3 + c];
          }
        }
      if (nvalid >= a.mvn && nok >= a.mvn) {
        for (int c = 0; c < 3; ++c) a.fimg[(size_t)px[i] * 3 + c] = s[c] / (float)nok;
        a.out_mask[px[i]] = 1;
      }
    }
  }
}

__global__ void k_fs_out(FSArgs a) {
  const size_t n = (size_t)a.H * a.W * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a.out_img[i] = (uint8_t)(a.fimg[i] * 255.0f);
}

}  // namespace

extern "C" size_t wf_fill_small_cracks_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  return (size_t)H * W * 3 * sizeof(float) + (size_t)H * W + 64;
}

extern "C" int wf_fill_small_cracks(const void* img, const void* mask, const float* original_depth, int has_depth_conf, void* out_img,
                                    void* out_mask, int H, int W, float depth_threshold, int max_crack_size, int min_valid_neighbors,
                                    void* workspace, void* stream) {
  WF_CHECK_ARG(img && mask && out_img && out_mask && workspace, "wf_fill_small_cracks: null pointer");
  WF_CHECK_ARG(!has_depth_conf || original_depth, "wf_fill_small_cracks: the depth-guided step needs the source view's depth map");
  WF_CHECK_ARG(H >= 2 && W >= 2 && (size_t)H * W < (1u << 31), "wf_fill_small_cracks: bad sizes H=%d W=%d", H, W);
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  FSArgs a;
  a.img = (const uint8_t*)img; a.mask = (const uint8_t*)mask; a.odepth = original_depth;
  a.out_img = (uint8_t*)out_img; a.out_mask = (uint8_t*)out_mask;
  a.fimg = (float*)ws;
  a.small = ws + (size_t)H * W * 3 * sizeof(float);
  a.cnt = (unsigned*)(ws + ((size_t)H * W * 3 * sizeof(float) + (size_t)H * W + 15) / 16 * 16);
  a.H = H; a.W = W; a.mvn = min_valid_neighbors; a.max_crack = max_crack_size; a.use_depth = has_depth_conf ? 1 : 0; a.thr = depth_threshold;
  if (hipMemsetAsync(a.cnt, 0, 2 * sizeof(unsigned), st) != hipSuccess) return check_hip(hipGetLastError(), "wf_fill_small_cracks: memset");
  const dim3 grid((unsigned)std::min<size_t>(((size_t)H * W + 255) / 256, 1024)), blk(256);
  hipLaunchKernelGGL(k_fs_step1, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_fs_small, grid, blk, 0, st, a);
  hipLaunchKernelGGL(k_fs_step2, dim3(1), dim3(64), 0, st, a);
  hipLaunchKernelGGL(k_fs_out, grid, blk, 0, st, a);
  WF_LAUNCH_CHECK("wf_fill_small_cracks");
  return WF_OK;
}

namespace {
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
