// Point-cloud renderer + depth-edge point filter of the DepthCrafter stage-1 warper (dynamic scenes: one point cloud per video frame).
//
// Replaces DepthCrafter/utils.py project_points_to_image_pytorch (:103-171: pytorch3d PointsRasterizer with radius 0.005, nearest point per
// pixel, 5 x 5 opening of the coverage mask) and filter_edge_points / detect_depth_edges (:495-567: Sobel magnitude, 7 x 7 dilation, 5 x 5
// min / max depth-jump test) as warp_depthcrafter.py:255-288 runs them for every frame.  HBM / atomic bound: a 64-bit atomicMin z-buffer
// (view depth bits << 32 | point index: nearest wins, ties to the smaller index), a handful of 5 x 5 / 7 x 7 stencils on byte masks.
// pytorch3d and OpenCV are third-party packages absent from the reference tree: their behaviour is restated (oracle/pointrender.py lists the
// statements) and parity with them is UNPINNED.  Compiled with -ffp-contract=off: the projection is the oracle's float32 op sequence.
#include <stdint.h>

#include "common.h"

using namespace wf;

namespace {

struct PRCam {
  float R[9];   // pytorch3d rotation (row vectors: view = p R + T)
  float T[3];
  float focal[2], p0[2];
};

// rasterization_utils.cuh PixToNonSquareNdc
__device__ __forceinline__ float pix_to_ndc(int i, int S1, int S2) {
  const float r = S1 <= S2 ? 2.0f : ((float)S1 * 2.0f) / (float)S2;
  const float o = r / 2.0f;
  return -o + (r * (float)i + o) / (float)S1;
}

__global__ void k_pr_splat(const float* __restrict__ pts, const unsigned char* __restrict__ drop, int n, PRCam c, int H, int W, float radius,
                           unsigned long long* __restrict__ zbuf) {
  const float r2 = radius * radius;
  const float rx = W <= H ? 2.0f : ((float)W * 2.0f) / (float)H, ry = H <= W ? 2.0f : ((float)H * 2.0f) / (float)W;
  const int reach = (int)ceilf(radius * fmaxf((float)W / rx, (float)H / ry)) + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (drop && drop[i]) continue;
    const float px = pts[3 * (size_t)i], py = pts[3 * (size_t)i + 1], pz = pts[3 * (size_t)i + 2];
    float v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = ((px * c.R[j] + py * c.R[3 + j]) + pz * c.R[6 + j]) + c.T[j];
    const float z = v[2];
    if (!(z >= 0.0f)) continue;  // behind the camera (rasterize_points: pz < 0), NaN
    const float x = (c.focal[0] * v[0] + c.p0[0] * z) / z, y = (c.focal[1] * v[1] + c.p0[1] * z) / z;
    if (!isfinite(x) || !isfinite(y)) continue;
    // flipped pixel indices i' = S - 1 - i of the centres around the point
    const int bx = (int)floorf((x + rx * 0.5f) * (float)W / rx - 0.5f), by = (int)floorf((y + ry * 0.5f) * (float)H / ry - 0.5f);
    const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)i;
    for (int dy = -reach; dy <= reach + 1; ++dy) {
      const int iy = by + dy;
      if (iy < 0 || iy >= H) continue;
      const float ddy = pix_to_ndc(iy, H, W) - y;
      for (int dx = -reach; dx <= reach + 1; ++dx) {
        const int ix = bx + dx;
        if (ix < 0 || ix >= W) continue;
        const float ddx = pix_to_ndc(ix, W, H) - x;
        if (ddx * ddx + ddy * ddy < r2) atomicMin(&zbuf[(size_t)(H - 1 - iy) * W + (W - 1 - ix)], key);
      }
    }
  }
}

__global__ void k_pr_mask(const unsigned long long* __restrict__ zbuf, unsigned char* __restrict__ m, size_t hw) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (size_t)gridDim.x * blockDim.x) m[i] = zbuf[i] != ~0ull;
}

// K x K erosion (MIN) or dilation (MAX) of a byte mask, anchor at the centre, pixels outside the image ignored (OpenCV's default border
// value for morphology: +inf for erode, -inf for dilate)
template <bool ERODE>
__global__ void k_morph(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W, int K) {
  const int r = K >> 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)H * W; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    unsigned char v = ERODE ? 1 : 0;
    for (int dy = -r; dy <= r; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = -r; dx <= r; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= W) continue;
        const unsigned char s = in[(size_t)yy * W + xx] != 0;
        v = ERODE ? (v & s) : (v | s);
      }
    }
    out[i] = v;
  }
}

// image[p] = features[idx[p]] where the final mask is set, else 0 (utils.py:149-169)
__global__ void k_pr_resolve(const unsigned long long* __restrict__ zbuf, const unsigned char* __restrict__ mask,
                             const float* __restrict__ feat, int F, float* __restrict__ img, size_t hw) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (size_t)gridDim.x * blockDim.x) {
    const bool on = mask[i] != 0;
    const unsigned idx = (unsigned)(zbuf[i] & 0xffffffffull);
    for (int k = 0; k < F; ++k) img[i * F + k] = on ? feat[(size_t)idx * F + k] : 0.0f;
  }
}

// ---- filter_edge_points ----
__device__ __forceinline__ int reflect101(int i, int n) {  // BORDER_REFLECT_101 (n >= 2)
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : i;
}
__device__ __forceinline__ int reflect_edge(int i, int n) {  // scipy.ndimage mode="reflect": d c b a | a b c d | d c b a
  if (i < 0) i = -i - 1;
  if (i >= n) i = 2 * n - 1 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
// Sobel (ksize 3, CV_64F) gradient magnitude, and its maximum as the bits of a non-negative double (atomicMax on the integer image)
__global__ void k_sobel_mag(const float* __restrict__ d, double* __restrict__ mag, unsigned long long* __restrict__ gmax, int H, int W) {
  unsigned long long lm = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)H * W; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H), xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
    auto at = [&](int yy, int xx) { return (double)d[(size_t)yy * W + xx]; };
    // correlation in row-major tap order (dy, dx), zero taps skipped
    const double gx = ((((-at(ym, xm) + at(ym, xp)) - 2.0 * at(y, xm)) + 2.0 * at(y, xp)) - at(yp, xm)) + at(yp, xp);
    const double gy = ((((-at(ym, xm) - 2.0 * at(ym, x)) - at(ym, xp)) + at(yp, xm)) + 2.0 * at(yp, x)) + at(yp, xp);
    const double m = sqrt(gx * gx + gy * gy);
    mag[i] = m;
    const unsigned long long b = (unsigned long long)__double_as_longlong(m);
    if (m == m && b > lm) lm = b;
  }
  if (lm && lm > __builtin_nontemporal_load(gmax)) atomicMax(gmax, lm);  // (same-address atomics serialise: only while the maximum still grows)
}
__global__ void k_edge_thresh(const double* __restrict__ mag, const unsigned long long* __restrict__ gmax, double thr, unsigned char* __restrict__ e,
                              size_t hw) {
  const double mx = __longlong_as_double((long long)*gmax);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (size_t)gridDim.x * blockDim.x) {
    const double m = mx > 0.0 ? mag[i] / mx : mag[i];
    e[i] = m > thr;
  }
}
// drop |= (max - min over the (2 r + 1)^2 window, scipy "reflect" border) > jump
__global__ void k_depth_jump(const float* __restrict__ d, unsigned char* __restrict__ drop, int H, int W, int r, float jump) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)H * W; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    float mn = INFINITY, mx = -INFINITY;
    for (int dy = -r; dy <= r; ++dy) {
      const int yy = reflect_edge(y + dy, H);
      for (int dx = -r; dx <= r; ++dx) {
        const float v = d[(size_t)yy * W + reflect_edge(x + dx, W)];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
    }
    if (mx - mn > jump) drop[i] = 1;
  }
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t wf_points_render_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const size_t hw = (size_t)H * W;
  return al256(hw * 8) + 2 * al256(hw);  // z-buffer, two byte masks
}

extern "C" int wf_points_render(const float* points, const float* features, const void* drop_u8, int n, int F, const float* camera,
                                int H, int W, float radius, int morph, float* out_image, void* out_mask_u8, void* workspace,
                                void* stream) {
  const unsigned char* drop = (const unsigned char*)drop_u8;
  unsigned char* out_mask = (unsigned char*)out_mask_u8;
  WF_CHECK_ARG(points && features && camera && out_image && out_mask && workspace, "wf_points_render: null pointer");
  WF_CHECK_ARG(n > 0 && F > 0 && H > 0 && W > 0 && radius > 0.0f, "wf_points_render: empty problem");
  hipStream_t s = (hipStream_t)stream;
  const size_t hw = (size_t)H * W;
  unsigned char* base = (unsigned char*)workspace;
  unsigned long long* zbuf = (unsigned long long*)base;
  unsigned char* m0 = base + al256(hw * 8);
  unsigned char* m1 = m0 + al256(hw);
  PRCam c;
  for (int i = 0; i < 9; ++i) c.R[i] = camera[i];
  for (int i = 0; i < 3; ++i) c.T[i] = camera[9 + i];
  c.focal[0] = camera[12]; c.focal[1] = camera[13]; c.p0[0] = camera[14]; c.p0[1] = camera[15];
  if (hipMemsetAsync(zbuf, 0xff, hw * 8, s) != hipSuccess) return check_hip(hipGetLastError(), "wf_points_render");
  hipLaunchKernelGGL(k_pr_splat, dim3(grid_for((size_t)n, 256, 16384)), dim3(256), 0, s, points, drop, n, c, H, W, radius, zbuf);
  const int g = grid_for(hw, 256, 16384);
  hipLaunchKernelGGL(k_pr_mask, dim3(g), dim3(256), 0, s, (const unsigned long long*)zbuf, morph ? m0 : out_mask, hw);
  if (morph) {  // cv2.morphologyEx(mask, MORPH_OPEN, ones(5, 5))
    hipLaunchKernelGGL(k_morph<true>, dim3(g), dim3(256), 0, s, (const unsigned char*)m0, m1, H, W, 5);
    hipLaunchKernelGGL(k_morph<false>, dim3(g), dim3(256), 0, s, (const unsigned char*)m1, out_mask, H, W, 5);
  }
  hipLaunchKernelGGL(k_pr_resolve, dim3(g), dim3(256), 0, s, (const unsigned long long*)zbuf, (const unsigned char*)out_mask, features, F,
                     out_image, hw);
  WF_LAUNCH_CHECK("wf_points_render");
  return WF_OK;
}

extern "C" size_t wf_depth_edge_mask_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const size_t hw = (size_t)H * W;
  return 256 + al256(hw * 8) + al256(hw);  // maximum, magnitudes, thresholded edges
}

extern "C" int wf_depth_edge_mask(const float* depth, int H, int W, double edge_threshold, int edge_dilation, float jump_threshold,
                                  int neighbor_radius, void* out_drop_u8, void* workspace, void* stream) {
  unsigned char* out_drop = (unsigned char*)out_drop_u8;
  WF_CHECK_ARG(depth && out_drop && workspace, "wf_depth_edge_mask: null pointer");
  WF_CHECK_ARG(H >= 2 && W >= 2, "wf_depth_edge_mask: the image must be at least 2 x 2");
  WF_CHECK_ARG(edge_dilation >= 0 && edge_dilation <= 15 && neighbor_radius >= 0 && neighbor_radius <= 15, "wf_depth_edge_mask: radii in 0..15");
  hipStream_t s = (hipStream_t)stream;
  const size_t hw = (size_t)H * W;
  unsigned char* base = (unsigned char*)workspace;
  unsigned long long* gmax = (unsigned long long*)base;
  double* mag = (double*)(base + 256);
  unsigned char* e = base + 256 + al256(hw * 8);
  if (hipMemsetAsync(gmax, 0, 8, s) != hipSuccess) return check_hip(hipGetLastError(), "wf_depth_edge_mask");
  const int g = grid_for(hw, 256, 16384);
  hipLaunchKernelGGL(k_sobel_mag, dim3(g), dim3(256), 0, s, depth, mag, gmax, H, W);
  hipLaunchKernelGGL(k_edge_thresh, dim3(g), dim3(256), 0, s, (const double*)mag, (const unsigned long long*)gmax, edge_threshold,
                     edge_dilation > 0 ? e : out_drop, hw);
  if (edge_dilation > 0)
    hipLaunchKernelGGL(k_morph<false>, dim3(g), dim3(256), 0, s, (const unsigned char*)e, out_drop, H, W, 2 * edge_dilation + 1);
  if (jump_threshold > 0.0f && neighbor_radius > 0)
    hipLaunchKernelGGL(k_depth_jump, dim3(g), dim3(256), 0, s, depth, out_drop, H, W, neighbor_radius, jump_threshold);
  WF_LAUNCH_CHECK("wf_depth_edge_mask");
  return WF_OK;
}
