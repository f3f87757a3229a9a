// Scheduler / injection element-wise kernels (HBM-bound).  Compiled with -ffp-contract=off so that every
// multiply and add rounds separately, exactly like the eager PyTorch statements they replace.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

using namespace wf;

// ------------------------------------------------------------------------------------------------
// error plumbing + library info
// ------------------------------------------------------------------------------------------------
namespace wf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return WF_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return WF_EHIP;
}
}  // namespace wf

extern "C" int wf_version(void) { return 100; }
extern "C" const char* wf_last_error(void) { return wf::g_err; }
extern "C" int wf_device_info(int dev, int* n_cu, int* clock_khz, char* name, int name_len) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return check_hip(e, "hipGetDeviceProperties");
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (clock_khz) *clock_khz = p.clockRate;
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return WF_OK;
}

#define EW_BLOCK 256
#define EW_LOOP(i, n) \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------
// PIPE:611   noise_pred + guidance_scale * (noise_pred - noise_uncond)
//   torch ops: d = cond - uncond ; s = g * d ; out = cond + s   (each rounds to bf16 when the tensors are bf16)
// ------------------------------------------------------------------------------------------------
template <bool RB>
__global__ void k_cfg(TView c, TView u, TView o, float g, size_t n) {
  EW_LOOP(i, n) {
    float a = tload(c, i), b = tload(u, i);
    float d = rnd<RB>(a - b);
    float s = rnd<RB>(g * d);
    tstore(o, i, rnd<RB>(a + s));
  }
}
extern "C" int wf_cfg_combine(const void* cond, const void* uncond, void* out, int dt, float g, size_t n, void* stream) {
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(cond && uncond && out, "wf_cfg_combine: null pointer");
  WF_CHECK_ARG(dt == WF_F32 || dt == WF_BF16, "wf_cfg_combine: bad dtype %d", dt);
  TView c{(void*)cond, dt}, u{(void*)uncond, dt}, o{out, dt};
  dim3 g3(grid_for(n, EW_BLOCK));
  if (dt == WF_BF16)
    hipLaunchKernelGGL(k_cfg<true>, g3, dim3(EW_BLOCK), 0, (hipStream_t)stream, c, u, o, g, n);
  else
    hipLaunchKernelGGL(k_cfg<false>, g3, dim3(EW_BLOCK), 0, (hipStream_t)stream, c, u, o, g, n);
  WF_LAUNCH_CHECK("wf_cfg_combine");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:958  x0_pred = sample - sigma_t * model_output
//   sigma_t is a 0-dim fp32 tensor: sigma_t*v keeps v's dtype; the subtraction promotes (sample, v).
// ------------------------------------------------------------------------------------------------
__global__ void k_x0(TView s, TView v, TView o, float sigma, bool rb_v, bool rb_o, size_t n) {
  EW_LOOP(i, n) {
    float t = sigma * tload(v, i);
    if (rb_v) t = rbf(t);
    float r = tload(s, i) - t;
    if (rb_o) r = rbf(r);
    tstore(o, i, r);
  }
}
extern "C" int wf_x0_from_v(const void* sample, int dt_s, const void* v, int dt_v, void* out, float sigma, size_t n,
                            void* stream) {
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(sample && v && out, "wf_x0_from_v: null pointer");
  int dt_o = (dt_s == WF_BF16 && dt_v == WF_BF16) ? WF_BF16 : WF_F32;
  TView s{(void*)sample, dt_s}, vv{(void*)v, dt_v}, o{out, dt_o};
  hipLaunchKernelGGL(k_x0, dim3(grid_for(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, s, vv, o, sigma,
                     dt_v == WF_BF16, dt_o == WF_BF16, n);
  WF_LAUNCH_CHECK("wf_x0_from_v");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:1083-1098 UniP (bh2, predict_x0):
//   x_t_ = sigma_t/sigma_s0 * x - alpha_t*h_phi_1 * m0
//   D1   = (m1 - m0) / rk ; pred_res = einsum('k,bkc...', [0.5], [D1]) ; x_t = x_t_ - alpha_t*B_h*pred_res ; .to(x.dtype)
// ------------------------------------------------------------------------------------------------
struct UniPFlags {
  bool ra, rb, rxt, rD, rP, rout;
  bool order2;
};
__global__ void k_unipc(TView x, TView m0, TView m1, TView o, float c1, float c2, float c3, float rk, UniPFlags f, size_t n) {
  EW_LOOP(i, n) {
    float xv = tload(x, i), m0v = tload(m0, i);
    float a = c1 * xv;
    if (f.ra) a = rbf(a);
    float b = c2 * m0v;
    if (f.rb) b = rbf(b);
    float xt = a - b;
    if (f.rxt) xt = rbf(xt);
    if (f.order2) {
      float d = tload(m1, i) - m0v;
      if (f.rD) d = rbf(d);
      d = d / rk;
      if (f.rD) d = rbf(d);
      float p = 0.5f * d;
      if (f.rP) p = rbf(p);
      float t = c3 * p;
      if (f.rP) t = rbf(t);
      xt = xt - t;
      if (f.rout) xt = rbf(xt);
    }
    tstore(o, i, xt);
  }
}
extern "C" int wf_unipc_update(const void* x, int dt_x, const void* m0, int dt_m0, const void* m1, int dt_m1, void* out,
                               float c1, float c2, float c3, float rk, size_t n, void* stream) {
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(x && m0 && out, "wf_unipc_update: null pointer");
  UniPFlags f;
  f.order2 = (m1 != nullptr);
  f.ra = dt_x == WF_BF16;
  f.rb = dt_m0 == WF_BF16;
  f.rxt = f.ra && f.rb;
  f.rD = f.order2 && dt_m1 == WF_BF16 && dt_m0 == WF_BF16;
  f.rP = f.rD && dt_x == WF_BF16;  // einsum(rhos_p[x.dtype], D1s)
  f.rout = f.rxt && f.rP;
  TView xv{(void*)x, dt_x}, m0v{(void*)m0, dt_m0}, m1v{(void*)m1, dt_m1}, o{out, dt_x};
  hipLaunchKernelGGL(k_unipc, dim3(grid_for(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, xv, m0v, m1v, o, c1, c2,
                     c3, rk, f, n);
  WF_LAUNCH_CHECK("wf_unipc_update");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:1584  noisy = (1 - sigma_a) * original_samples + sigma_a * noise
// ------------------------------------------------------------------------------------------------
__global__ void k_add_noise(TView x0, TView nz, TView o, float oms, float s, bool rb_a, bool rb_b, bool rb_o, size_t n) {
  EW_LOOP(i, n) {
    float a = oms * tload(x0, i);
    if (rb_a) a = rbf(a);
    float b = s * tload(nz, i);
    if (rb_b) b = rbf(b);
    float r = a + b;
    if (rb_o) r = rbf(r);
    tstore(o, i, r);
  }
}
extern "C" int wf_add_noise(const void* x0, int dt_x0, const void* noise, int dt_n, void* out, float one_minus_sigma,
                            float sigma, size_t n, void* stream) {
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(x0 && noise && out, "wf_add_noise: null pointer");
  bool bx = dt_x0 == WF_BF16, bn = dt_n == WF_BF16;
  int dt_o = (bx && bn) ? WF_BF16 : WF_F32;
  TView a{(void*)x0, dt_x0}, b{(void*)noise, dt_n}, o{out, dt_o};
  hipLaunchKernelGGL(k_add_noise, dim3(grid_for(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, a, b, o,
                     one_minus_sigma, sigma, bx, bx && bn, bx && bn, n);
  WF_LAUNCH_CHECK("wf_add_noise");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:1281 / 1385 per-channel affine
// ------------------------------------------------------------------------------------------------
struct ChanConst {
  float mean[64];
  float istd[64];
};
__global__ void k_affine(TView z, TView o, ChanConst cc, int dir, int C, size_t inner, bool rb, size_t n) {
  EW_LOOP(i, n) {
    int c = (int)((i / inner) % (size_t)C);
    float v = tload(z, i);
    float r;
    if (dir == 0) {
      r = v / cc.istd[c];
      if (rb) r = rbf(r);
      r = r + cc.mean[c];
      if (rb) r = rbf(r);
    } else {
      r = v - cc.mean[c];
      if (rb) r = rbf(r);
      r = r * cc.istd[c];
      if (rb) r = rbf(r);
    }
    tstore(o, i, r);
  }
}
extern "C" int wf_latent_affine(const void* z, int dt_in, void* out, int dt_out, const float* mean, const float* istd,
                                int dir, int B, int C, size_t inner, void* stream) {
  WF_CHECK_ARG(z && out && mean && istd, "wf_latent_affine: null pointer");
  WF_CHECK_ARG(C > 0 && C <= 64, "wf_latent_affine: C=%d out of range (1..64)", C);
  WF_CHECK_ARG(dir == 0 || dir == 1, "wf_latent_affine: dir must be 0 or 1");
  size_t n = (size_t)B * C * inner;
  if (n == 0) return WF_OK;
  ChanConst cc;
  for (int c = 0; c < C; ++c) {
    cc.mean[c] = mean[c];
    cc.istd[c] = istd[c];
  }
  TView zi{(void*)z, dt_in}, o{out, dt_out};
  hipLaunchKernelGGL(k_affine, dim3(grid_for(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, zi, o, cc, dir, C,
                     inner, dt_in == WF_BF16, n);
  WF_LAUNCH_CHECK("wf_latent_affine");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:1375-1380  video_latents = 2*ref - 1 ; fused = video_latents*mask + decoded*(1 - mask)
//   float4 path: 3 streams in, 1 out, mask broadcast over the 3 channels.
// ------------------------------------------------------------------------------------------------
__global__ void k_blend(const float* __restrict__ ref, const float* __restrict__ mask, const float* __restrict__ dec,
                        float* __restrict__ out, int C, size_t inner, size_t n) {
  EW_LOOP(i, n) {
    size_t b = i / ((size_t)C * inner);
    size_t p = i % inner;
    float m = mask[b * inner + p];
    float r = 2.0f * ref[i] - 1.0f;
    float a = r * m;
    float bb = dec[i] * (1.0f - m);
    out[i] = a + bb;
  }
}
__global__ void k_blend4(const float4* __restrict__ ref, const float4* __restrict__ mask, const float4* __restrict__ dec,
                         float4* __restrict__ out, int C, size_t inner4, size_t n4) {
  EW_LOOP(i, n4) {
    size_t b = i / ((size_t)C * inner4);
    size_t p = i % inner4;
    float4 m = mask[b * inner4 + p];
    float4 r = ref[i], d = dec[i], o;
    o.x = (2.0f * r.x - 1.0f) * m.x + d.x * (1.0f - m.x);
    o.y = (2.0f * r.y - 1.0f) * m.y + d.y * (1.0f - m.y);
    o.z = (2.0f * r.z - 1.0f) * m.z + d.z * (1.0f - m.z);
    o.w = (2.0f * r.w - 1.0f) * m.w + d.w * (1.0f - m.w);
    out[i] = o;
  }
}
// The video case (C = 3 colour planes, 16-byte aligned, inner % 4 == 0): one thread handles one float4 of pixels in ALL THREE planes -- the
// mask is read once (the algorithmic (3 + 1 + 3 + 3) x 4 bytes per pixel), no 64-bit division per element (k_blend4 spends two on
// its flat index), blockIdx.y = batch.  Same expression per element as k_blend4 / k_blend: same bits.
__global__ __launch_bounds__(256) void k_blend4_rgb(const float4* __restrict__ ref, const float4* __restrict__ mask, const float4* __restrict__ dec,
                                                    float4* __restrict__ out, size_t inner4) {
  const size_t b = blockIdx.y;
  const float4* mb = mask + b * inner4;
  const size_t base = b * 3 * inner4;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < inner4; p += (size_t)gridDim.x * blockDim.x) {
    const float4 m = mb[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = base + (size_t)c * inner4 + p;
      const float4 r = ref[i], d = dec[i];
      float4 o;
      o.x = (2.0f * r.x - 1.0f) * m.x + d.x * (1.0f - m.x);
      o.y = (2.0f * r.y - 1.0f) * m.y + d.y * (1.0f - m.y);
      o.z = (2.0f * r.z - 1.0f) * m.z + d.z * (1.0f - m.z);
      o.w = (2.0f * r.w - 1.0f) * m.w + d.w * (1.0f - m.w);
      out[i] = o;
    }
  }
}
extern "C" int wf_blend_pixels(const float* ref, const float* mask, const float* dec, float* out, int B, int C,
                               size_t inner, void* stream) {
  WF_CHECK_ARG(ref && mask && dec && out, "wf_blend_pixels: null pointer");
  size_t n = (size_t)B * C * inner;
  if (n == 0) return WF_OK;
  bool al = (((uintptr_t)ref | (uintptr_t)mask | (uintptr_t)dec | (uintptr_t)out) & 15) == 0 && (inner % 4 == 0);
  if (al && C == 3 && B <= 65535) {
    hipLaunchKernelGGL(k_blend4_rgb, dim3(grid_for(inner / 4, 256, 8192), B), dim3(256), 0, (hipStream_t)stream, (const float4*)ref,
                       (const float4*)mask, (const float4*)dec, (float4*)out, inner / 4);
  } else if (al) {
    size_t n4 = n / 4;
    hipLaunchKernelGGL(k_blend4, dim3(grid_for(n4, EW_BLOCK, 4096)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       (const float4*)ref, (const float4*)mask, (const float4*)dec, (float4*)out, C, inner / 4, n4);
  } else {
    hipLaunchKernelGGL(k_blend, dim3(grid_for(n, EW_BLOCK, 4096)), dim3(EW_BLOCK), 0, (hipStream_t)stream, ref, mask, dec,
                       out, C, inner, n);
  }
  WF_LAUNCH_CHECK("wf_blend_pixels");
  return WF_OK;
}

// Which pixel COLUMNS of the decoded video can reach the blend's result: fused = (2 ref - 1) m + dec (1 - m) is (2 ref - 1) + (+-0) wherever
// m == 1 and dec is finite, whatever dec is there.  out[0] = first column holding a pixel with mask != 1 (NaN counts), out[1] = last such
// column + 1, over n_rows rows of W pixels; {W, 0} when every pixel is 1.  Integer atomics only: deterministic.
__global__ void k_mask_cols_init(int* out, int W) {
  out[0] = W;
  out[1] = 0;
}
__global__ void k_mask_cols(const float* __restrict__ mask, size_t n, int W, int* __restrict__ out) {
  int lo = W, hi = 0;
  EW_LOOP(i, n) {
    if (!(mask[i] == 1.0f)) {
      const int x = (int)(i % (size_t)W);
      lo = min(lo, x);
      hi = max(hi, x + 1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o, 64));
    hi = max(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0 && hi > 0) {
    atomicMin(&out[0], lo);
    atomicMax(&out[1], hi);
  }
}
extern "C" int wf_mask_column_range(const float* mask, size_t n_rows, int W, int* out2, void* stream) {
  WF_CHECK_ARG(mask && out2 && W > 0, "wf_mask_column_range: null pointer / W = %d", W);
  hipLaunchKernelGGL(k_mask_cols_init, dim3(1), dim3(1), 0, (hipStream_t)stream, out2, W);
  const size_t n = n_rows * (size_t)W;
  if (n) hipLaunchKernelGGL(k_mask_cols, dim3(grid_for(n, EW_BLOCK, 4096)), dim3(EW_BLOCK), 0, (hipStream_t)stream, mask, n, W, out2);
  WF_LAUNCH_CHECK("wf_mask_column_range");
  return WF_OK;
}

// The same blend as eager PyTorch evaluates it when the VAE is a bf16 module (LongCat: run_longcat_worldforge_single.py:205): the decoded
// video is bf16, `video_latents.to(decoded_video.dtype)` / `mask.to(...)` round the fp32 reference and mask to bf16
// (scheduling_flow_match_euler_discrete.py:1152-1153) and every statement of :1156-1164 rounds its result to bf16:
//   v = bf16(2 * r) ; v = bf16(v - 1) ; a = bf16(v * m) ; om = bf16(1 - m) ; b = bf16(dec * om) ; fused = bf16(a + b)
__global__ void k_blend_bf16(const float* __restrict__ ref, const float* __restrict__ mask, const uint16_t* __restrict__ dec,
                             uint16_t* __restrict__ out, int C, size_t inner, size_t n) {
  EW_LOOP(i, n) {
    size_t b = i / ((size_t)C * inner);
    size_t p = i % inner;
    const float m = rbf(mask[b * inner + p]);
    float v = rbf(2.0f * rbf(ref[i]));
    v = rbf(v - 1.0f);
    const float a = rbf(v * m);
    const float om = rbf(1.0f - m);
    const float bb = rbf(bf16_to_f32(dec[i]) * om);
    out[i] = f32_to_bf16(a + bb);
  }
}
extern "C" int wf_blend_pixels_bf16(const float* ref, const float* mask, const void* dec, void* out, int B, int C, size_t inner,
                                    void* stream) {
  WF_CHECK_ARG(ref && mask && dec && out, "wf_blend_pixels_bf16: null pointer");
  size_t n = (size_t)B * C * inner;
  if (n == 0) return WF_OK;
  hipLaunchKernelGGL(k_blend_bf16, dim3(grid_for(n, EW_BLOCK, 8192)), dim3(EW_BLOCK), 0, (hipStream_t)stream, ref, mask,
                     (const uint16_t*)dec, (uint16_t*)out, C, inner, n);
  WF_LAUNCH_CHECK("wf_blend_pixels_bf16");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// PIPE:744  (video / 2 + 0.5).clamp(0, 1), [C,F,H,W] -> [F,H,W,C]
// ------------------------------------------------------------------------------------------------
__global__ void k_post(const float* __restrict__ x, float* __restrict__ out, int C, size_t fhw, size_t n) {
  EW_LOOP(i, n) {
    size_t p = i / C;
    int c = (int)(i % C);
    float v = x[(size_t)c * fhw + p] / 2.0f + 0.5f;
    out[i] = fminf(fmaxf(v, 0.0f), 1.0f);
  }
}
// a bf16 video (bf16 VAE module): diffusers' VideoProcessor.denormalize runs in the video's dtype -- bf16(x / 2), bf16(+ 0.5), clamp --
// and pt_to_numpy converts to float afterwards
__global__ void k_post_bf16(const uint16_t* __restrict__ x, float* __restrict__ out, int C, size_t fhw, size_t n) {
  EW_LOOP(i, n) {
    size_t p = i / C;
    int c = (int)(i % C);
    float v = rbf(rbf(bf16_to_f32(x[(size_t)c * fhw + p]) / 2.0f) + 0.5f);
    out[i] = fminf(fmaxf(v, 0.0f), 1.0f);
  }
}
extern "C" int wf_postprocess_video_bf16(const void* x, float* out, int C, int F, int H, int W, void* stream) {
  size_t fhw = (size_t)F * H * W, n = fhw * C;
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(x && out, "wf_postprocess_video_bf16: null pointer");
  hipLaunchKernelGGL(k_post_bf16, dim3(grid_for(n, EW_BLOCK, 4096)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const uint16_t*)x, out, C, fhw, n);
  WF_LAUNCH_CHECK("wf_postprocess_video_bf16");
  return WF_OK;
}
extern "C" int wf_postprocess_video(const float* x, float* out, int C, int F, int H, int W, void* stream) {
  size_t fhw = (size_t)F * H * W, n = fhw * C;
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(x && out, "wf_postprocess_video: null pointer");
  hipLaunchKernelGGL(k_post, dim3(grid_for(n, EW_BLOCK, 4096)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, out, C, fhw, n);
  WF_LAUNCH_CHECK("wf_postprocess_video");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void k_cast(TView a, TView o, size_t n) {
  EW_LOOP(i, n) tstore(o, i, tload(a, i));
}
extern "C" int wf_cast(const void* in, int dt_in, void* out, int dt_out, size_t n, void* stream) {
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(in && out, "wf_cast: null pointer");
  TView a{(void*)in, dt_in}, o{out, dt_out};
  hipLaunchKernelGGL(k_cast, dim3(grid_for(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, a, o, n);
  WF_LAUNCH_CHECK("wf_cast");
  return WF_OK;
}

// ------------------------------------------------------------------------------------------------
// SCHED:1410-1412  encoded_video[:, c] = pred_original_sample[:, c]
// ------------------------------------------------------------------------------------------------
struct IdxList {
  int idx[64];
  int n;
};
__global__ void k_chswap(TView enc, TView pred, IdxList il, int B, int C, size_t inner) {
  size_t per = (size_t)il.n * inner;
  size_t n = per * B;
  EW_LOOP(i, n) {
    size_t b = i / per;
    size_t r = i % per;
    int c = il.idx[r / inner];
    size_t off = (b * C + c) * inner + r % inner;
    tstore(enc, off, tload(pred, off));
  }
}
extern "C" int wf_channel_swap(void* enc, int dt_enc, const void* pred, int dt_pred, const int* idx, int n_idx, int B,
                               int C, size_t inner, void* stream) {
  WF_CHECK_ARG(enc && pred, "wf_channel_swap: null pointer");
  WF_CHECK_ARG(n_idx >= 0 && n_idx <= 64 && n_idx <= C, "wf_channel_swap: n_idx=%d out of range", n_idx);
  if (n_idx == 0 || B == 0 || inner == 0) return WF_OK;
  IdxList il;
  il.n = n_idx;
  for (int i = 0; i < n_idx; ++i) {
    WF_CHECK_ARG(idx[i] >= 0 && idx[i] < C, "wf_channel_swap: channel %d out of range", idx[i]);
    il.idx[i] = idx[i];
  }
  TView e{enc, dt_enc}, p{(void*)pred, dt_pred};
  hipLaunchKernelGGL(k_chswap, dim3(grid_for((size_t)B * n_idx * inner, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     e, p, il, B, C, inner);
  WF_LAUNCH_CHECK("wf_channel_swap");
  return WF_OK;
}
