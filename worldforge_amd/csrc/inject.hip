// Trajectory-injection reductions and resizes: DSG (PIPE:664-681), FLF flow metric (SCHED:497-607),
// temporal-difference motion (SCHED:391-392), bilinear / nearest resize (SCHED:1316-1324, 1355-1362).
// HBM-bound; reductions use a fixed two-level tree (per-thread -> wave shuffle -> LDS -> per-block partial -> one
// finishing block), no floating-point atomics, so results are bit-identical run to run.
// Compiled with -ffp-contract=off (every torch op rounds separately).
#include "common.h"

using namespace wf;

#define RED_BLOCK 256
#define RED_NBLK 256  // per-block partials of the first level

// Block-level sum of K floats per thread -> thread 0 holds the totals.
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float* smem /* K * 4 floats */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) smem[k * 4 + wid] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = (smem[k * 4 + 0] + smem[k * 4 + 1]) + (smem[k * 4 + 2] + smem[k * 4 + 3]);
  }
}

// ------------------------------------------------------------------------------------------------
// DSG
// ------------------------------------------------------------------------------------------------
template <bool RB>
__global__ void k_dsg_reduce(TView g, TView w, float* ws, size_t n) {
  __shared__ float sm[12];
  float acc[3] = {0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float a = tload(g, i), b = tload(w, i);
    acc[0] += rnd<RB>(a * b);
    acc[1] += rnd<RB>(a * a);
    acc[2] += rnd<RB>(b * b);
  }
  block_sum<3>(acc, sm);
  if (threadIdx.x == 0) {
    ws[8 + 3 * blockIdx.x + 0] = acc[0];
    ws[8 + 3 * blockIdx.x + 1] = acc[1];
    ws[8 + 3 * blockIdx.x + 2] = acc[2];
  }
}
// PIPE:669-676 on three scalars.  One block; fixed-order tree over the RED_NBLK partials.
template <bool RB>
__global__ void k_dsg_coeff(float* ws, int nblk) {
  __shared__ float sm[12];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    acc[0] += ws[8 + 3 * i + 0];
    acc[1] += ws[8 + 3 * i + 1];
    acc[2] += ws[8 + 3 * i + 2];
  }
  block_sum<3>(acc, sm);
  if (threadIdx.x == 0) {
    float dot = rnd<RB>(acc[0]), ng2 = rnd<RB>(acc[1]), nw2 = rnd<RB>(acc[2]);
    float ng = rnd<RB>(sqrtf(ng2)), nw = rnd<RB>(sqrtf(nw2));
    float den = rnd<RB>(rnd<RB>(ng * nw) + 1e-8f);
    float c = rnd<RB>(dot / den);
    float cl = fminf(fmaxf(c, -1.0f), 1.0f);
    float ang = rnd<RB>(acosf(cl));
    float s = rnd<RB>(sinf(ang));
    float ratio = rnd<RB>(ng / rnd<RB>(nw + 1e-8f));
    ws[0] = dot;
    ws[1] = ng2;
    ws[2] = nw2;
    ws[3] = c;
    ws[4] = s;
    ws[5] = ratio;
    ws[6] = 0.f;
    ws[7] = 0.f;
  }
}
// PIPE:681  good + omega*sin_theta * (good - (magnitude_ratio*cos_theta) * worse)
template <bool RB>
__global__ void k_dsg_apply(TView g, TView w, TView o, const float* ws, float omega, size_t n) {
  const float A = rnd<RB>(omega * ws[4]);
  const float Bc = rnd<RB>(ws[5] * ws[3]);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float a = tload(g, i), b = tload(w, i);
    float t = rnd<RB>(Bc * b);
    t = rnd<RB>(a - t);
    t = rnd<RB>(A * t);
    tstore(o, i, rnd<RB>(a + t));
  }
}
extern "C" size_t wf_dsg_workspace_floats(void) { return 8 + 3 * RED_NBLK; }
extern "C" int wf_dsg(const void* g, const void* w, void* out, int dt, float omega, size_t n, float* ws, void* stream) {
  WF_CHECK_ARG(g && w && out && ws, "wf_dsg: null pointer");
  WF_CHECK_ARG(dt == WF_F32 || dt == WF_BF16, "wf_dsg: bad dtype %d", dt);
  WF_CHECK_ARG(n > 0, "wf_dsg: empty tensor");
  TView gv{(void*)g, dt}, wv{(void*)w, dt}, ov{out, dt};
  hipStream_t s = (hipStream_t)stream;
  int nblk = grid_for(n, RED_BLOCK, RED_NBLK);
  if (dt == WF_BF16) {
    hipLaunchKernelGGL(k_dsg_reduce<true>, dim3(nblk), dim3(RED_BLOCK), 0, s, gv, wv, ws, n);
    hipLaunchKernelGGL(k_dsg_coeff<true>, dim3(1), dim3(RED_BLOCK), 0, s, ws, nblk);
    hipLaunchKernelGGL(k_dsg_apply<true>, dim3(grid_for(n, RED_BLOCK)), dim3(RED_BLOCK), 0, s, gv, wv, ov, ws, omega, n);
  } else {
    hipLaunchKernelGGL(k_dsg_reduce<false>, dim3(nblk), dim3(RED_BLOCK), 0, s, gv, wv, ws, n);
    hipLaunchKernelGGL(k_dsg_coeff<false>, dim3(1), dim3(RED_BLOCK), 0, s, ws, nblk);
    hipLaunchKernelGGL(k_dsg_apply<false>, dim3(grid_for(n, RED_BLOCK)), dim3(RED_BLOCK), 0, s, gv, wv, ov, ws, omega, n);
  }
  WF_LAUNCH_CHECK("wf_dsg");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:391-392 / 478-479: channel_motion = x[:, :, 1:] - x[:, :, :-1] (then permuted to [T-1, 1, h, w] per channel)
// ------------------------------------------------------------------------------------------------
__global__ void k_tdiff(TView x, float* out, int C, int T, size_t hw, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    size_t p = i % hw;
    size_t ct = i / hw;
    int t = (int)(ct % (T - 1));
    size_t c = ct / (T - 1);
    size_t src = (c * T + t) * hw + p;
    out[i] = tload(x, src + hw) - tload(x, src);
  }
}
extern "C" int wf_temporal_diff(const void* x, int dt, float* out, int C, int T, size_t hw, void* stream) {
  WF_CHECK_ARG(x && out, "wf_temporal_diff: null pointer");
  WF_CHECK_ARG(T >= 2, "wf_temporal_diff: needs at least 2 frames, got %d", T);
  size_t n = (size_t)C * (T - 1) * hw;
  if (n == 0) return WF_OK;
  TView xv{(void*)x, dt};
  hipLaunchKernelGGL(k_tdiff, dim3(grid_for(n, RED_BLOCK)), dim3(RED_BLOCK), 0, (hipStream_t)stream, xv, out, C, T, hw, n);
  WF_LAUNCH_CHECK("wf_temporal_diff");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// CFG-zero (LongCat pipeline_longcat_video.py:374-383, 875-888): st = <c,u> / (|u|^2 + 1e-8);
// out = +-(u*st + g*(c - u*st)), every product / sum rounded separately as the reference's torch ops.  fp32.
// ------------------------------------------------------------------------------------------------
__global__ void k_cfgz_coeff(float* ws, int nblk) {
  __shared__ float sm[12];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    acc[0] += ws[8 + 3 * i + 0];
    acc[1] += ws[8 + 3 * i + 1];
    acc[2] += ws[8 + 3 * i + 2];
  }
  block_sum<3>(acc, sm);
  if (threadIdx.x == 0) {
    ws[0] = acc[0];
    ws[1] = acc[1];
    ws[2] = acc[2];
    ws[3] = acc[0] / (acc[2] + 1e-8f);
  }
}
__global__ void k_cfgz_apply(const float* __restrict__ c, const float* __restrict__ u, float* __restrict__ o, const float* ws, float g,
                             int negate, size_t n) {
  const float st = ws[3];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float us = u[i] * st;
    const float d = c[i] - us;
    const float r = us + g * d;
    o[i] = negate ? -r : r;
  }
}
extern "C" int wf_cfg_zero(const float* cond, const float* uncond, float* out, float guidance, int negate, size_t n, float* ws,
                           void* stream) {
  WF_CHECK_ARG(cond && uncond && out && ws, "wf_cfg_zero: null pointer");
  WF_CHECK_ARG(n > 0, "wf_cfg_zero: empty tensor");
  TView cv{(void*)cond, WF_F32}, uv{(void*)uncond, WF_F32};
  hipStream_t s = (hipStream_t)stream;
  const int nblk = grid_for(n, RED_BLOCK, RED_NBLK);
  hipLaunchKernelGGL(k_dsg_reduce<false>, dim3(nblk), dim3(RED_BLOCK), 0, s, cv, uv, ws, n);
  hipLaunchKernelGGL(k_cfgz_coeff, dim3(1), dim3(RED_BLOCK), 0, s, ws, nblk);
  hipLaunchKernelGGL(k_cfgz_apply, dim3(grid_for(n, RED_BLOCK)), dim3(RED_BLOCK), 0, s, cond, uncond, out, ws, guidance, negate, n);
  WF_LAUNCH_CHECK("wf_cfg_zero");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:497-607 _compute_flow_metrics (mask=None path): per channel three means -> similarity
// ------------------------------------------------------------------------------------------------
#define FM_NBLK 32
__global__ void k_flow_partial(const float* __restrict__ ref, const float* __restrict__ chan, float* ws, int Tm, int Cr,
                               int Cc, size_t hw, int outlier_or) {
  __shared__ float sm[12];
  const int ch = blockIdx.y;
  const size_t npix = (size_t)Tm * hw;
  const float* r = ref + (size_t)ch * Tm * Cr * hw;
  const float* c = chan + (size_t)ch * Tm * Cc * hw;
  float acc[3] = {0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i / hw, p = i % hw;
    float r0 = r[(t * Cr + 0) * hw + p];
    float r1 = (Cr >= 2) ? r[(t * Cr + 1) * hw + p] : r0;
    float c0 = c[(t * Cc + 0) * hw + p];
    float c1 = (Cc >= 2) ? c[(t * Cc + 1) * hw + p] : c0;
    float d0 = r0 - c0, d1 = r1 - c1;
    float epe = sqrtf((d0 * d0 + d1 * d1) + 1e-8f);
    float dot = r0 * c0 + r1 * c1;
    float rn = sqrtf((r0 * r0 + r1 * r1) + 1e-8f);
    float cn = sqrtf((c0 * c0 + c1 * c1) + 1e-8f);
    float ca = dot / (rn * cn + 1e-8f);
    ca = fminf(fmaxf(ca, -1.0f), 1.0f);
    float ae = acosf(ca) * 180.0f / 3.14159265358979323846f;
    const bool o_abs = epe > 3.0f, o_rel = epe > rn * 0.05f;
    bool outl = outlier_or ? (o_abs || o_rel) : (o_abs && o_rel);
    acc[0] += epe;
    acc[1] += ae;
    acc[2] += outl ? 1.0f : 0.0f;
  }
  block_sum<3>(acc, sm);
  if (threadIdx.x == 0) {
    float* o = ws + ((size_t)ch * FM_NBLK + blockIdx.x) * 3;
    o[0] = acc[0];
    o[1] = acc[1];
    o[2] = acc[2];
  }
}
__global__ void k_flow_final(const float* ws, float* sim, int Tm, size_t hw, float w_epe, float w_fl, float w_ae) {
  const int ch = blockIdx.x;
  // 64 threads, FM_NBLK (=32) partials: fixed-order wave tree
  float a[3] = {0.f, 0.f, 0.f};
  if (threadIdx.x < FM_NBLK) {
    const float* p = ws + ((size_t)ch * FM_NBLK + threadIdx.x) * 3;
    a[0] = p[0];
    a[1] = p[1];
    a[2] = p[2];
  }
  a[0] = wave_sum(a[0]);
  a[1] = wave_sum(a[1]);
  a[2] = wave_sum(a[2]);
  if (threadIdx.x == 0) {
    float N = (float)((size_t)Tm * hw);
    float m_epe = a[0] / N, m_ae = a[1] / N, fl = a[2] / N;
    float ne = fminf(fmaxf(m_epe / 10.0f, 0.f), 1.f);
    float nf = fminf(fmaxf(fl / 0.5f, 0.f), 1.f);
    float na = fminf(fmaxf(m_ae / 30.0f, 0.f), 1.f);
    float werr = (w_epe * ne + w_fl * nf) + w_ae * na;
    float s = 1.0f - werr;
    sim[ch] = fminf(fmaxf(s, 0.f), 1.f);
  }
}
extern "C" size_t wf_flow_metrics_workspace_floats(int n_channels) { return (size_t)n_channels * FM_NBLK * 3; }
extern "C" int wf_flow_metrics_variant(const float* ref_flow, const float* chan_flow, float* sim, int n_channels, int Tm, int Cr,
                                       int Cc, size_t hw, int variant, float* ws, void* stream) {
  WF_CHECK_ARG(ref_flow && chan_flow && sim && ws, "wf_flow_metrics: null pointer");
  WF_CHECK_ARG((Cr == 1 || Cr == 2) && (Cc == 1 || Cc == 2), "wf_flow_metrics: flow components must be 1 or 2");
  WF_CHECK_ARG(n_channels > 0 && Tm > 0 && hw > 0, "wf_flow_metrics: empty input");
  WF_CHECK_ARG(variant == 0 || variant == 1, "wf_flow_metrics: variant must be 0 (Wan) or 1 (LongCat)");
  hipStream_t s = (hipStream_t)stream;
  // variant 1 = the LongCat scheduler's metric (scheduling_flow_match_euler_discrete.py:172-243): outlier = abs OR rel, weights .4/.4/.2
  hipLaunchKernelGGL(k_flow_partial, dim3(FM_NBLK, n_channels), dim3(RED_BLOCK), 0, s, ref_flow, chan_flow, ws, Tm, Cr, Cc, hw, variant);
  if (variant == 0)
    hipLaunchKernelGGL(k_flow_final, dim3(n_channels), dim3(64), 0, s, ws, sim, Tm, hw, 0.45f, 0.45f, 0.1f);
  else
    hipLaunchKernelGGL(k_flow_final, dim3(n_channels), dim3(64), 0, s, ws, sim, Tm, hw, 0.4f, 0.4f, 0.2f);
  WF_LAUNCH_CHECK("wf_flow_metrics");
  return WF_OK;
}
extern "C" int wf_flow_metrics(const float* ref_flow, const float* chan_flow, float* sim, int n_channels, int Tm, int Cr,
                               int Cc, size_t hw, float* ws, void* stream) {
  return wf_flow_metrics_variant(ref_flow, chan_flow, sim, n_channels, Tm, Cr, Cc, hw, 0, ws, stream);
}

// ------------------------------------------------------------------------------------------------
// LongCat refine pass, stage-1 video -> model input (pipeline_longcat_video.py:1407-1413): uint8 frames [F][H0][W0][3]
//   -> bf16 -> F.interpolate(bilinear, align_corners=True) to (H, W)              (bf16 result)
//   -> / 255                                                                        (bf16 result)
//   -> F.interpolate(trilinear, align_corners=True) to (Fo, H, W): H, W unchanged, so only the frame axis interpolates (bf16 result)
//   -> * 2 - 1                                                                      (two bf16 roundings)
// out f32 [3][Fo][H][W] (bf16-valued).  Index / weight arithmetic as PyTorch's upsample kernels: scale = (in-1)/(out-1) in fp32,
// src = scale * dst, i0 = (int)src, lambda1 = src - i0.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float refine_tap(const unsigned char* __restrict__ frame, int H0, int W0, int c, int y0, int y1, float ly1,
                                            int x0, int x1, float lx1) {
  const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
  const float v00 = frame[((size_t)y0 * W0 + x0) * 3 + c], v01 = frame[((size_t)y0 * W0 + x1) * 3 + c];
  const float v10 = frame[((size_t)y1 * W0 + x0) * 3 + c], v11 = frame[((size_t)y1 * W0 + x1) * 3 + c];
  const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  return rbf(rbf(v) / 255.0f);
}
__global__ void k_refine_upsample(const unsigned char* __restrict__ in, float* __restrict__ out, int F, int H0, int W0, int Fo, int H,
                                  int W, float sf, float sy, float sx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), fo = (int)((i / ((size_t)W * H)) % Fo), c = (int)(i / ((size_t)W * H * Fo));
    const float fy = sy * y, fx = sx * x, ff = sf * fo;
    const int y0 = (int)fy, x0 = (int)fx, f0 = (int)ff;
    const int y1 = y0 + (y0 < H0 - 1 ? 1 : 0), x1 = x0 + (x0 < W0 - 1 ? 1 : 0), f1 = f0 + (f0 < F - 1 ? 1 : 0);
    const float ly1 = fy - y0, lx1 = fx - x0, lf1 = ff - f0;
    const float a = refine_tap(in + (size_t)f0 * H0 * W0 * 3, H0, W0, c, y0, y1, ly1, x0, x1, lx1);
    const float b = refine_tap(in + (size_t)f1 * H0 * W0 * 3, H0, W0, c, y0, y1, ly1, x0, x1, lx1);
    const float t = rbf((1.0f - lf1) * a + lf1 * b);
    out[i] = rbf(rbf(t * 2.0f) - 1.0f);
  }
}
extern "C" int wf_refine_upsample_u8(const void* frames_u8, float* out, int F, int H0, int W0, int Fo, int H, int W, void* stream) {
  WF_CHECK_ARG(frames_u8 && out, "wf_refine_upsample_u8: null pointer");
  WF_CHECK_ARG(F > 0 && H0 > 0 && W0 > 0 && Fo > 0 && H > 0 && W > 0, "wf_refine_upsample_u8: empty shape");
  const size_t n = (size_t)3 * Fo * H * W;
  const float sf = Fo > 1 ? (float)(F - 1) / (float)(Fo - 1) : 0.0f;
  const float sy = H > 1 ? (float)(H0 - 1) / (float)(H - 1) : 0.0f;
  const float sx = W > 1 ? (float)(W0 - 1) / (float)(W - 1) : 0.0f;
  hipLaunchKernelGGL(k_refine_upsample, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)frames_u8, out, F, H0, W0, Fo, H, W, sf, sy, sx, n);
  WF_LAUNCH_CHECK("wf_refine_upsample_u8");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// Resize: PyTorch upsample_bilinear2d (align_corners=False) and legacy 'nearest'
// ------------------------------------------------------------------------------------------------
__global__ void k_bilinear(const float* __restrict__ in, float* __restrict__ out, int Hi, int Wi, int Ho, int Wo, float rh,
                           float rw, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int x = (int)(i % Wo);
    int y = (int)((i / Wo) % Ho);
    size_t nimg = i / ((size_t)Wo * Ho);
    float sy = rh * ((float)y + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    int y0 = (int)sy;
    int yp = (y0 < Hi - 1) ? 1 : 0;
    float ly1 = sy - (float)y0, ly0 = 1.0f - ly1;
    float sx = rw * ((float)x + 0.5f) - 0.5f;
    sx = sx < 0.f ? 0.f : sx;
    int x0 = (int)sx;
    int xp = (x0 < Wi - 1) ? 1 : 0;
    float lx1 = sx - (float)x0, lx0 = 1.0f - lx1;
    const float* p = in + nimg * (size_t)Hi * Wi;
    float v00 = p[(size_t)y0 * Wi + x0], v01 = p[(size_t)y0 * Wi + x0 + xp];
    float v10 = p[(size_t)(y0 + yp) * Wi + x0], v11 = p[(size_t)(y0 + yp) * Wi + x0 + xp];
    out[i] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
}
__global__ void k_nearest(const float* __restrict__ in, float* __restrict__ out, int Hi, int Wi, int Ho, int Wo, float rh,
                          float rw, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int x = (int)(i % Wo);
    int y = (int)((i / Wo) % Ho);
    size_t nimg = i / ((size_t)Wo * Ho);
    int ys = min((int)floorf((float)y * rh), Hi - 1);
    int xs = min((int)floorf((float)x * rw), Wi - 1);
    out[i] = in[nimg * (size_t)Hi * Wi + (size_t)ys * Wi + xs];
  }
}
static int resize_common(const char* name, bool bil, const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo,
                         void* stream) {
  WF_CHECK_ARG(in && out, "%s: null pointer", name);
  WF_CHECK_ARG(N >= 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "%s: bad sizes", name);
  size_t n = (size_t)N * Ho * Wo;
  if (n == 0) return WF_OK;
  float rh = (float)Hi / (float)Ho, rw = (float)Wi / (float)Wo;
  if (bil)
    hipLaunchKernelGGL(k_bilinear, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, in, out, Hi, Wi, Ho, Wo,
                       rh, rw, n);
  else
    hipLaunchKernelGGL(k_nearest, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, in, out, Hi, Wi, Ho, Wo,
                       rh, rw, n);
  WF_LAUNCH_CHECK(name);
  return WF_OK;
}
extern "C" int wf_resize_bilinear2d(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, void* stream) {
  return resize_common("wf_resize_bilinear2d", true, in, out, N, Hi, Wi, Ho, Wo, stream);
}
extern "C" int wf_resize_nearest2d(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, void* stream) {
  return resize_common("wf_resize_nearest2d", false, in, out, N, Hi, Wi, Ho, Wo, stream);
}

// ---- soften_mask (infer_worldforge.py:105-150) --------------------------------------------------------------------------------------
// Inside the ones-region of each frame, pixels within `td` pixels of the zero-region get ramp(d / td), d = Euclidean distance to the
// nearest zero pixel of the frame (scipy.ndimage.distance_transform_edt in the reference: the exact EDT, float64).  Only d <= td
// matters, so the nearest zero is searched in the (2 td + 1)^2 window: d^2 is an integer, d = sqrt in double is the value scipy
// returns; the ramp is evaluated in double and rounded to float once, as numpy does.  Frames that are all ones / all zeros come out
// unchanged by construction.  One thread per pixel, the window is served from L1 / L2 (0.4 M pixels per frame).
__global__ void k_soften_mask(const float* __restrict__ mask, float* __restrict__ out, int H, int W, int td, int decay, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  const int y = (int)((i / W) % H);
  const float* fr = mask + (i - (size_t)y * W - x);
  const float v = fr[(size_t)y * W + x];
  if (v == 0.0f) {  // .astype(bool): any non-zero value belongs to the ones-region
    out[i] = v;
    return;
  }
  int best = 0x7fffffff;
  const int td2 = td * td;
  for (int dy = -td; dy <= td; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    const int rem = td2 - dy * dy;  // dx^2 <= rem
    const float* row = fr + (size_t)yy * W;
    for (int dx = 0; dx * dx <= rem; ++dx) {
      const bool zl = x - dx >= 0 && row[x - dx] == 0.0f;
      const bool zr = x + dx < W && row[x + dx] == 0.0f;
      if (zl || zr) {
        best = min(best, dx * dx + dy * dy);
        break;  // larger |dx| in this row is farther
      }
    }
  }
  if (best > td2) {  // distance_from_ones > transition_distance (or no zero in the frame): value kept
    out[i] = v;
    return;
  }
  const double t = fmin(fmax(sqrt((double)best) / (double)td, 0.0), 1.0);
  const double kHalfPi = 3.14159265358979323846 / 2;
  double r;
  switch (decay) {
    case 0: r = t; break;                          // linear
    case 1: r = 1.0 - exp(-3.0 * t); break;        // exponential
    case 2: r = sin(kHalfPi * t); break;           // sine
    default: r = 1.0 - cos(kHalfPi * t); break;    // cosine
  }
  out[i] = (float)r;
}
extern "C" int wf_soften_mask(const float* mask, float* out, int F, int H, int W, int transition_distance, int decay_type,
                              void* stream) {
  WF_CHECK_ARG(mask && out, "wf_soften_mask: null pointer");
  WF_CHECK_ARG(F >= 0 && H > 0 && W > 0, "wf_soften_mask: bad sizes");
  WF_CHECK_ARG(transition_distance >= 1 && transition_distance <= 64, "wf_soften_mask: transition_distance %d out of range 1..64",
               transition_distance);
  WF_CHECK_ARG(decay_type >= 0 && decay_type <= 3, "wf_soften_mask: decay_type %d (0 linear, 1 exponential, 2 sine, 3 cosine)",
               decay_type);
  const size_t n = (size_t)F * H * W;
  if (n == 0) return WF_OK;
  hipLaunchKernelGGL(k_soften_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, out, H, W,
                     transition_distance, decay_type, n);
  WF_LAUNCH_CHECK("wf_soften_mask");
  return WF_OK;
}
