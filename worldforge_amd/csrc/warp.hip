// Stage-1 depth-guided forward warping (vggt/modules/utils_warp.py:863-945, warp_single_img): every source pixel with a valid depth is
// un-projected, moved to each new camera, projected and rounded to the nearest target pixel; the NEAREST source (smallest camera-space z:
// the reference sorts far-to-near and lets the last write win) owns the target pixel.  Integer / byte scatter work, HBM- and atomic-bound:
// pass 1 scatters a sortable 64-bit image of z with atomicMin into a z-buffer, pass 2 lets the winner write colour, mask and depth.
// All geometry in fp64, as the reference's numpy (float64 matrices, float32 depth).
#include "common.h"

using namespace wf;

namespace {

struct WarpGeom {      // doubles, row-major
  double Kinv[9], K[9], Rinv[9], tinv[3];
};

__device__ __forceinline__ unsigned long long zkey(double z) {  // order-preserving map double -> uint64
  unsigned long long u = (unsigned long long)__double_as_longlong(z);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

// returns false when the source pixel does not land in the target image of camera `cam` (R|t rows in cams[12])
__device__ __forceinline__ bool project(const WarpGeom& g, const double* __restrict__ cam, int x, int y, float depth, int H, int W,
                                        int& tu, int& tv, double& z) {
  if (!(depth > 0.0f)) return false;  // NaN or non-positive depth (utils_warp.py:868)
  const double d = (double)depth;
  // 3-term dot products as one FMA chain over k (what a BLAS dgemm micro-kernel does; the reference's numpy matmul goes through
  // whatever BLAS it is linked with).  In flat-depth regions neighbouring sources land on the same target with z equal up to the last
  // few fp64 bits, so WHICH of them wins there is decided by the evaluation order of the host library: the tests accept either.
#define WF_DOT3(a0, a1, a2, b0, b1, b2) fma((a2), (b2), fma((a1), (b1), (a0) * (b0)))
  const double cx = WF_DOT3(g.Kinv[0], g.Kinv[1], g.Kinv[2], (double)x, (double)y, 1.0) * d;
  const double cy = WF_DOT3(g.Kinv[3], g.Kinv[4], g.Kinv[5], (double)x, (double)y, 1.0) * d;
  const double cz = WF_DOT3(g.Kinv[6], g.Kinv[7], g.Kinv[8], (double)x, (double)y, 1.0) * d;
  const double wx = WF_DOT3(g.Rinv[0], g.Rinv[1], g.Rinv[2], cx, cy, cz) + g.tinv[0];
  const double wy = WF_DOT3(g.Rinv[3], g.Rinv[4], g.Rinv[5], cx, cy, cz) + g.tinv[1];
  const double wz = WF_DOT3(g.Rinv[6], g.Rinv[7], g.Rinv[8], cx, cy, cz) + g.tinv[2];
  const double px = WF_DOT3(cam[0], cam[1], cam[2], wx, wy, wz) + cam[3];
  const double py = WF_DOT3(cam[4], cam[5], cam[6], wx, wy, wz) + cam[7];
  z = WF_DOT3(cam[8], cam[9], cam[10], wx, wy, wz) + cam[11];
  if (!(fabs(z) > 1e-6)) return false;
  const double nx = px / z, ny = py / z;
  const double u = WF_DOT3(g.K[0], g.K[1], g.K[2], nx, ny, z / z);
  const double v = WF_DOT3(g.K[3], g.K[4], g.K[5], nx, ny, z / z);
#undef WF_DOT3
  if (!(u >= 0.0 && u < (double)W && v >= 0.0 && v < (double)H)) return false;
  tu = min(max((int)rint(u), 0), W - 1);  // np.round (half to even) + clip
  tv = min(max((int)rint(v), 0), H - 1);
  return true;
}

__global__ void k_warp_zpass(const float* __restrict__ depth, const double* __restrict__ params, const double* __restrict__ cams,
                             unsigned long long* __restrict__ zbuf, int n, int H, int W) {
  const WarpGeom& g = *reinterpret_cast<const WarpGeom*>(params);
  const size_t hw = (size_t)H * W, total = hw * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / hw);
    const size_t p = i - (size_t)c * hw;
    int tu, tv;
    double z;
    if (project(g, cams + 12 * c, (int)(p % W), (int)(p / W), depth[p], H, W, tu, tv, z))
      atomicMin(&zbuf[(size_t)c * hw + (size_t)tv * W + tu], zkey(z));
  }
}
__global__ void k_warp_write(const float* __restrict__ image, const float* __restrict__ depth, const double* __restrict__ params,
                             const double* __restrict__ cams, const unsigned long long* __restrict__ zbuf,
                             unsigned char* __restrict__ out_img, unsigned char* __restrict__ out_mask, float* __restrict__ out_depth, int n,
                             int H, int W) {
  const WarpGeom& g = *reinterpret_cast<const WarpGeom*>(params);
  const size_t hw = (size_t)H * W, total = hw * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / hw);
    const size_t p = i - (size_t)c * hw;
    int tu, tv;
    double z;
    if (!project(g, cams + 12 * c, (int)(p % W), (int)(p / W), depth[p], H, W, tu, tv, z)) continue;
    const size_t t = (size_t)c * hw + (size_t)tv * W + tu;
    if (zbuf[t] != zkey(z)) continue;  // a nearer source owns this target pixel
#pragma unroll
    for (int k = 0; k < 3; ++k) out_img[t * 3 + k] = (unsigned char)(image[p * 3 + k] * 255.0f);  // (img * 255).astype(uint8), :931-932
    out_mask[t] = 1;
    out_depth[t] = (float)z;
  }
}

}  // namespace

extern "C" int wf_warp_splat(const float* image, const float* depth, const double* geometry, const double* cameras, void* out_images,
                             void* out_masks, float* out_depth, void* zbuffer, int n_cameras, int H, int W, void* stream) {
  WF_CHECK_ARG(image && depth && geometry && cameras && out_images && out_masks && out_depth && zbuffer, "wf_warp_splat: null pointer");
  WF_CHECK_ARG(n_cameras > 0 && H > 0 && W > 0, "wf_warp_splat: empty problem");
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)n_cameras * H * W;
  // z-buffer = +inf keys, images / masks = 0, depth = NaN (utils_warp.py:879-892)
  if (hipMemsetAsync(zbuffer, 0xff, total * 8, s) != hipSuccess || hipMemsetAsync(out_images, 0, total * 3, s) != hipSuccess ||
      hipMemsetAsync(out_masks, 0, total, s) != hipSuccess || hipMemsetAsync(out_depth, 0xff, total * 4, s) != hipSuccess)
    return check_hip(hipGetLastError(), "wf_warp_splat");
  const int grid = grid_for(total, 256, 16384);
  hipLaunchKernelGGL(k_warp_zpass, dim3(grid), dim3(256), 0, s, depth, geometry, cameras, (unsigned long long*)zbuffer, n_cameras, H, W);
  hipLaunchKernelGGL(k_warp_write, dim3(grid), dim3(256), 0, s, image, depth, geometry, cameras, (const unsigned long long*)zbuffer,
                     (unsigned char*)out_images, (unsigned char*)out_masks, out_depth, n_cameras, H, W);
  WF_LAUNCH_CHECK("wf_warp_splat");
  return WF_OK;
}
