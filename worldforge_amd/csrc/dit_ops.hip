// HBM-bound companions of the DiT GEMM / attention kernels (wan/modules/model.py):
//   wf_ln_modulate   : WanLayerNorm (+ AdaLN modulate or affine)         model.py:92-102, 303, 311, 346, 262-264, 356-358
//   wf_rmsnorm_heads : WanRMSNorm over the full channel dim + 3-axis RoPE, scattered to the attention layout
//                      model.py:73-89, 142-143, 43-70 (q, k of self-attention), 215-218 (cross-attention, no RoPE)
//   wf_v_transpose   : V -> blocked V^T layout consumed by wf_attn_fwd
//   wf_patchify / wf_unpatchify : model.py:534-537, 584-607
//   wf_act           : SiLU / GELU(erf) / add for the tiny embedding MLPs (model.py:458-464, 355-358)
// All are one-pass, 16-byte vectorised, one workgroup per token row (or per tile for the transpose).
#include "common.h"
#include "mfma.h"

using namespace wf;

namespace {

__device__ __forceinline__ float block_sum_1(float v, float* sm, int nwaves) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nwaves; ++i) t += sm[i];
  return t;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (eps, no affine) then y = ln * (plus_one + mul[c]) + add[c];  x fp32 [L, C] -> out bf16 or f32
// Two-pass statistics over register-resident data (mean first, then centred variance), as torch's LayerNorm.
// ------------------------------------------------------------------------------------------------
template <int VPT>  // float4 vectors per thread (C <= 256 * 4 * VPT)
__global__ __launch_bounds__(256) void k_ln_mod(const float* __restrict__ x, const float* __restrict__ mul,
                                                const float* __restrict__ add, void* __restrict__ out, int out_bf16, int C,
                                                float eps, float plus_one) {
  __shared__ float sm[8];
  const size_t row = blockIdx.x;
  const float4* xr = reinterpret_cast<const float4*>(x + row * C);
  const int nvec = C >> 2;
  float4 v[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int id = threadIdx.x + 256 * i;
    if (id < nvec) {
      v[i] = xr[id];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = block_sum_1(s, sm, 4) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int id = threadIdx.x + 256 * i;
    if (id < nvec) {
      float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = block_sum_1(q, sm, 4) / (float)C;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int id = threadIdx.x + 256 * i;
    if (id < nvec) {
      float4 m4 = mul ? reinterpret_cast<const float4*>(mul)[id] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 a4 = add ? reinterpret_cast<const float4*>(add)[id] : make_float4(0.f, 0.f, 0.f, 0.f);
      float y0 = (v[i].x - mean) * rstd * (plus_one + m4.x) + a4.x;
      float y1 = (v[i].y - mean) * rstd * (plus_one + m4.y) + a4.y;
      float y2 = (v[i].z - mean) * rstd * (plus_one + m4.z) + a4.z;
      float y3 = (v[i].w - mean) * rstd * (plus_one + m4.w) + a4.w;
      if (out_bf16) {
        u32x2 pk = {pack_bf16x2(y0, y1), pack_bf16x2(y2, y3)};
        reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(out) + row * C)[id] = pk;
      } else {
        reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row * C)[id] = make_float4(y0, y1, y2, y3);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm over C (weight w), optional RoPE (cos/sin [L, 64] per rotation pair), scatter to [H][Lout][128].
// in: bf16 [L, ld] (pointer already offset to the q / k column block).  128 threads, chunk = 8 bf16 = 4 rotation pairs.
// Rounding points follow the bf16 autocast flow: norm -> bf16 (type_as), * weight -> bf16, RoPE in fp32 -> bf16.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_rms_heads(const uint16_t* __restrict__ in, int ld, const float* __restrict__ w,
                                                   const float* __restrict__ cs, const float* __restrict__ sn,
                                                   uint16_t* __restrict__ out, int L, int Lout, int C, float eps, float out_scale,
                                                   float* __restrict__ norm2_rows) {
  __shared__ float sm[4];
  const int row = blockIdx.x;
  const uint16_t* xr = in + (size_t)row * ld;
  const int nch = C >> 3;
  constexpr int MAXC = 8;  // up to 8 chunks per thread: C <= 8192
  u32x4 v[MAXC];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    int id = threadIdx.x + 128 * i;
    if (id < nch) {
      v[i] = reinterpret_cast<const u32x4*>(xr)[id];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = __uint_as_float(v[i][k] << 16), b = __uint_as_float(v[i][k] & 0xffff0000u);
        ss += a * a + b * b;
      }
    }
  }
  const float tot = block_sum_1(ss, sm, 2);
  const float rinv = rsqrtf(tot / (float)C + eps);
  const int H = C >> 7;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    int id = threadIdx.x + 128 * i;
    if (id < nch) {
      const int head = id >> 4, within = id & 15;
      const float4 w0 = reinterpret_cast<const float4*>(w)[2 * id], w1 = reinterpret_cast<const float4*>(w)[2 * id + 1];
      const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      float y[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = __uint_as_float(v[i][k] << 16), b = __uint_as_float(v[i][k] & 0xffff0000u);
        y[2 * k] = rbf(rbf(a * rinv) * ww[2 * k]);
        y[2 * k + 1] = rbf(rbf(b * rinv) * ww[2 * k + 1]);
      }
      if (cs) {
        const float4 c4 = reinterpret_cast<const float4*>(cs + (size_t)row * 64)[within];
        const float4 s4 = reinterpret_cast<const float4*>(sn + (size_t)row * 64)[within];
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float re = y[2 * k] * cc[k] - y[2 * k + 1] * sv[k];
          float im = y[2 * k] * sv[k] + y[2 * k + 1] * cc[k];
          y[2 * k] = re;
          y[2 * k + 1] = im;
        }
      }
      if (out_scale != 1.0f) {  // softmax_scale * log2(e) folded into Q in front of the one bf16 rounding (wf_attn_fwd, softmax_scale = 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] *= out_scale;
      }
      u32x4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
      *reinterpret_cast<u32x4*>(out + ((size_t)head * Lout + row) * 128 + within * 8) = pk;
      if (norm2_rows) {
        // |y|^2 of this (row, head) over the values AS STORED (after the bf16 rounding): the 16 lanes of a head are consecutive lanes of one
        // wave (id = tid + 128 i, within = tid & 15).  Feeds the per-head norm bound of the attention kernel (wf_head_max_norm2's result)
        // without a second pass over the tensor.
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = __uint_as_float(pk[k] << 16), b = __uint_as_float(pk[k] & 0xffff0000u);
          q += a * a + b * b;
        }
        q += __shfl_xor(q, 8, 64);
        q += __shfl_xor(q, 4, 64);
        q += __shfl_xor(q, 2, 64);
        q += __shfl_xor(q, 1, 64);
        if (within == 0) norm2_rows[(size_t)row * H + head] = q;
      }
    }
  }
}

// max over the rows of norm2_rows [L][H] -> part [nblk][H] (RB rows per block), then part -> out [H].  Rows holding a NaN / inf report +inf
// (fmaxf would drop the NaN): the attention kernel must then take its max-tracking body (see k_head_max_norm2 in attention.hip).
constexpr int NB_RB = 128;
__device__ __forceinline__ float nb_max(float best, float v) { return !(v <= 3.0e38f) ? INFINITY : fmaxf(best, v); }
// thread t -> (head h = t % H, slice s = t / H of 256 / H slices); items s, s + ns, ... of [n][H]; the loads of 8 items are issued together
// (a dependent fmaxf chain would otherwise expose one memory latency per item: the pass is latency-, not bandwidth-bound)
__device__ __forceinline__ float nb_column_max(const float* __restrict__ src, int n, int H, int h, int s, int ns) {
  float best = 0.f;
  int r = s;
  for (; r + 7 * ns < n; r += 8 * ns) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(r + u * ns) * H + h];
#pragma unroll
    for (int u = 0; u < 8; ++u) best = nb_max(best, v[u]);
  }
  for (; r < n; r += ns) best = nb_max(best, src[(size_t)r * H + h]);
  return best;
}
__global__ __launch_bounds__(256) void k_norm2_part(const float* __restrict__ rows, int L, int H, float* __restrict__ part) {
  __shared__ float red[256];
  const int t = threadIdx.x, h = t % H, s = t / H, ns = 256 / H;
  const int r0 = blockIdx.x * NB_RB, n = min(L - r0, NB_RB);
  float best = s < ns ? nb_column_max(rows + (size_t)r0 * H, n, H, h, s, ns) : 0.f;
  red[t] = best;
  __syncthreads();
  if (t < H) {
    for (int k = 1; k < ns; ++k) best = fmaxf(best, red[k * H + t]);
    part[(size_t)blockIdx.x * H + t] = best;
  }
}
__global__ __launch_bounds__(256) void k_norm2_final(const float* __restrict__ part, int nblk, int H, float* __restrict__ out) {
  __shared__ float red[256];
  const int t = threadIdx.x, h = t % H, s = t / H, ns = 256 / H;
  float best = s < ns ? nb_column_max(part, nblk, H, h, s, ns) : 0.f;
  red[t] = best;
  __syncthreads();
  if (t < H) {
    for (int k = 1; k < ns; ++k) best = fmaxf(best, red[k * H + t]);
    out[t] = best;
  }
}

// ------------------------------------------------------------------------------------------------
// V [L, ld] bf16 (columns head*128 + d) -> Vt [H][Lp/64][128][64]; keys >= L are written as zeros.
// grid (Lp/64, H), 256 threads; 64x128 tile through LDS (row stride 130 halfwords to spread banks).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vt(const uint16_t* __restrict__ V, int ld, uint16_t* __restrict__ Vt, int L, int hstride) {
  __shared__ uint16_t tile[64][136];
  const int kt = blockIdx.x, head = blockIdx.y;
  const int tid = threadIdx.x;
  // load: 64 rows x 16 chunks of 8 -> 1024 chunks, 4 per thread
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int id = tid + 256 * i;
    int r = id >> 4, c = id & 15;
    int key = kt * 64 + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (key < L) v = *reinterpret_cast<const u32x4*>(V + (size_t)key * ld + head * 128 + c * 8);
    *reinterpret_cast<u32x4*>(&tile[r][c * 8]) = v;
  }
  __syncthreads();
  // store: 128 d-rows x 8 chunks of 8 keys -> 1024 chunks, 4 per thread
  uint16_t* dst = Vt + ((size_t)head * hstride + kt) * (128 * 64);  // hstride = 64-key tiles per head in the destination
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int id = tid + 256 * i;
    int d = id >> 3, c = id & 7;
    uint32_t p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = (uint32_t)tile[c * 8 + 2 * k][d] | ((uint32_t)tile[c * 8 + 2 * k + 1][d] << 16);
    u32x4 pk = {p[0], p[1], p[2], p[3]};
    *reinterpret_cast<u32x4*>(dst + d * 64 + c * 8) = pk;
  }
}

// ------------------------------------------------------------------------------------------------
// patchify: x [Cin, T, Hh, Ww] (bf16) -> tokens [L = T*(Hh/2)*(Ww/2), Cin*4] bf16, k = c*4 + ph*2 + pw  (model.py:534-537)
// ------------------------------------------------------------------------------------------------
__global__ void k_patchify(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int Cin, int T, int Hh, int Ww, size_t n) {
  const int h2 = Hh >> 1, w2 = Ww >> 1, K = Cin * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(i % K);
    size_t tok = i / K;
    int c = k >> 2, ph = (k >> 1) & 1, pw = k & 1;
    int wx = (int)(tok % w2), hy = (int)((tok / w2) % h2), f = (int)(tok / ((size_t)w2 * h2));
    out[i] = x[(((size_t)c * T + f) * Hh + (2 * hy + ph)) * Ww + (2 * wx + pw)];
  }
}
// unpatchify: y [L, 4*Cout] f32 (k = (ph*2+pw)*Cout + c) -> [Cout, T, Hh, Ww] f32   (model.py:584-607)
__global__ void k_unpatchify(const float* __restrict__ y, float* __restrict__ out, int Cout, int T, int Hh, int Ww, size_t n) {
  const int h2 = Hh >> 1, w2 = Ww >> 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int xw = (int)(i % Ww), yh = (int)((i / Ww) % Hh), f = (int)((i / ((size_t)Ww * Hh)) % T), c = (int)(i / ((size_t)Ww * Hh * T));
    size_t tok = ((size_t)f * h2 + (yh >> 1)) * w2 + (xw >> 1);
    int k = (((yh & 1) << 1) | (xw & 1)) * Cout + c;
    out[i] = y[tok * (4 * Cout) + k];
  }
}

// ------------------------------------------------------------------------------------------------
// tiny activations: mode 0 silu, 1 gelu(erf), 2 identity;  optional second input added first (a + b)
// ------------------------------------------------------------------------------------------------
__global__ void k_act(TView a, TView b, TView o, int mode, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = tload(a, i);
    if (b.p) v += tload(b, i);
    if (mode == 0)
      v = v / (1.0f + __expf(-v));
    else if (mode == 1)
      v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    tstore(o, i, v);
  }
}

}  // namespace

extern "C" int wf_ln_modulate(const float* x, const float* mul, const float* add, void* out, int out_dtype, int L, int C,
                              float eps, int plus_one, void* stream) {
  WF_CHECK_ARG(x && out, "wf_ln_modulate: null pointer");
  WF_CHECK_ARG(C % 4 == 0 && C <= 8192 && C > 0, "wf_ln_modulate: C=%d must be a multiple of 4 and <= 8192", C);
  if (L == 0) return WF_OK;
  hipStream_t s = (hipStream_t)stream;
  const float p1 = plus_one ? 1.0f : 0.0f;
  const int ob = out_dtype == WF_BF16;
  const int nvec = C / 4;
  if (nvec <= 512)
    hipLaunchKernelGGL(k_ln_mod<2>, dim3(L), dim3(256), 0, s, x, mul, add, out, ob, C, eps, p1);
  else if (nvec <= 1280)
    hipLaunchKernelGGL(k_ln_mod<5>, dim3(L), dim3(256), 0, s, x, mul, add, out, ob, C, eps, p1);
  else
    hipLaunchKernelGGL(k_ln_mod<8>, dim3(L), dim3(256), 0, s, x, mul, add, out, ob, C, eps, p1);
  WF_LAUNCH_CHECK("wf_ln_modulate");
  return WF_OK;
}

extern "C" int wf_rmsnorm_heads(const void* in, int ld, const float* weight, const float* cos_tab, const float* sin_tab,
                                void* out, int L, int Lout, int C, float eps, float out_scale, void* stream) {
  WF_CHECK_ARG(in && weight && out, "wf_rmsnorm_heads: null pointer");
  WF_CHECK_ARG(out_scale > 0.0f, "wf_rmsnorm_heads: out_scale must be positive (1 = none)");
  WF_CHECK_ARG(C % 128 == 0 && C <= 8192 && ld % 8 == 0, "wf_rmsnorm_heads: C=%d must be a multiple of 128 (<= 8192), ld %% 8", C);
  WF_CHECK_ARG(Lout >= L, "wf_rmsnorm_heads: Lout < L");
  WF_CHECK_ARG((cos_tab == nullptr) == (sin_tab == nullptr), "wf_rmsnorm_heads: cos/sin must both be given or both null");
  if (L == 0) return WF_OK;
  hipLaunchKernelGGL(k_rms_heads, dim3(L), dim3(128), 0, (hipStream_t)stream, (const uint16_t*)in, ld, weight, cos_tab,
                     sin_tab, (uint16_t*)out, L, Lout, C, eps, out_scale, (float*)nullptr);
  WF_LAUNCH_CHECK("wf_rmsnorm_heads");
  return WF_OK;
}

extern "C" size_t wf_rmsnorm_heads_bound_ws_floats(int L, int C) {
  if (L <= 0 || C <= 0) return 0;
  const size_t H = (size_t)C / 128;
  return (size_t)L * H + (size_t)((L + NB_RB - 1) / NB_RB) * H;
}

extern "C" int wf_rmsnorm_heads_bound(const void* in, int ld, const float* weight, const float* cos_tab, const float* sin_tab, void* out,
                                      int L, int Lout, int C, float eps, float out_scale, float* ws, float* max_norm2, void* stream) {
  WF_CHECK_ARG(in && weight && out && ws && max_norm2, "wf_rmsnorm_heads_bound: null pointer");
  WF_CHECK_ARG(out_scale > 0.0f, "wf_rmsnorm_heads_bound: out_scale must be positive (1 = none)");
  // C >= 128: H = C / 128 heads is a divisor in k_norm2_part / k_norm2_final (C = 0 passed the old check)
  WF_CHECK_ARG(C >= 128 && C % 128 == 0 && C <= 8192 && ld % 8 == 0, "wf_rmsnorm_heads_bound: C=%d must be a multiple of 128 in 128..8192, ld %% 8", C);
  WF_CHECK_ARG(Lout >= L && L > 0, "wf_rmsnorm_heads_bound: need 0 < L <= Lout");
  WF_CHECK_ARG((cos_tab == nullptr) == (sin_tab == nullptr), "wf_rmsnorm_heads_bound: cos/sin must both be given or both null");
  const int H = C / 128, nblk = (L + NB_RB - 1) / NB_RB;
  float* rows = ws;
  float* part = ws + (size_t)L * H;
  hipLaunchKernelGGL(k_rms_heads, dim3(L), dim3(128), 0, (hipStream_t)stream, (const uint16_t*)in, ld, weight, cos_tab,
                     sin_tab, (uint16_t*)out, L, Lout, C, eps, out_scale, rows);
  hipLaunchKernelGGL(k_norm2_part, dim3(nblk), dim3(256), 0, (hipStream_t)stream, rows, L, H, part);
  hipLaunchKernelGGL(k_norm2_final, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nblk, H, max_norm2);
  WF_LAUNCH_CHECK("wf_rmsnorm_heads_bound");
  return WF_OK;
}

extern "C" int wf_v_transpose(const void* V, int ld, void* Vt, int L, int Lp, int H, void* stream) {
  WF_CHECK_ARG(V && Vt, "wf_v_transpose: null pointer");
  WF_CHECK_ARG(Lp % 64 == 0 && Lp >= L && ld % 8 == 0, "wf_v_transpose: Lp must be a multiple of 64 and >= L");
  if (Lp == 0) return WF_OK;
  hipLaunchKernelGGL(k_vt, dim3(Lp / 64, H), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)V, ld, (uint16_t*)Vt, L, Lp / 64);
  WF_LAUNCH_CHECK("wf_v_transpose");
  return WF_OK;
}

extern "C" int wf_v_transpose_seg(const void* V, int ld, void* Vt, int L, int Lp, int H, int head_stride_tiles, void* stream) {
  WF_CHECK_ARG(V && Vt, "wf_v_transpose_seg: null pointer");
  WF_CHECK_ARG(Lp % 64 == 0 && Lp >= L && ld % 8 == 0, "wf_v_transpose_seg: Lp must be a multiple of 64 and >= L");
  WF_CHECK_ARG(head_stride_tiles >= Lp / 64, "wf_v_transpose_seg: head stride (%d tiles) shorter than the segment (%d)", head_stride_tiles, Lp / 64);
  if (Lp == 0) return WF_OK;
  hipLaunchKernelGGL(k_vt, dim3(Lp / 64, H), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)V, ld, (uint16_t*)Vt, L, head_stride_tiles);
  WF_LAUNCH_CHECK("wf_v_transpose_seg");
  return WF_OK;
}

extern "C" int wf_patchify(const void* x, void* tokens, int Cin, int T, int Hh, int Ww, void* stream) {
  WF_CHECK_ARG(x && tokens, "wf_patchify: null pointer");
  WF_CHECK_ARG(Hh % 2 == 0 && Ww % 2 == 0, "wf_patchify: latent H, W must be even");
  size_t n = (size_t)T * (Hh / 2) * (Ww / 2) * Cin * 4;
  if (n == 0) return WF_OK;
  hipLaunchKernelGGL(k_patchify, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (uint16_t*)tokens, Cin, T, Hh, Ww, n);
  WF_LAUNCH_CHECK("wf_patchify");
  return WF_OK;
}

extern "C" int wf_unpatchify(const float* y, float* out, int Cout, int T, int Hh, int Ww, void* stream) {
  WF_CHECK_ARG(y && out, "wf_unpatchify: null pointer");
  WF_CHECK_ARG(Hh % 2 == 0 && Ww % 2 == 0, "wf_unpatchify: latent H, W must be even");
  size_t n = (size_t)Cout * T * Hh * Ww;
  if (n == 0) return WF_OK;
  hipLaunchKernelGGL(k_unpatchify, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, y, out, Cout, T, Hh, Ww, n);
  WF_LAUNCH_CHECK("wf_unpatchify");
  return WF_OK;
}

extern "C" int wf_act(const void* a, int dt_a, const void* b, int dt_b, void* out, int dt_out, int mode, size_t n, void* stream) {
  if (n == 0) return WF_OK;
  WF_CHECK_ARG(a && out, "wf_act: null pointer");
  WF_CHECK_ARG(mode >= 0 && mode <= 2, "wf_act: mode must be 0 (silu), 1 (gelu erf) or 2 (identity)");
  TView av{(void*)a, dt_a}, bv{(void*)b, dt_b}, ov{out, dt_out};
  hipLaunchKernelGGL(k_act, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, av, bv, ov, mode, n);
  WF_LAUNCH_CHECK("wf_act");
  return WF_OK;
}
