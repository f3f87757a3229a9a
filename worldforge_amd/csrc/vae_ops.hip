// HBM-bound companions of the VAE convolutions (wan/modules/vae.py), channels-last activations:
//   wf_rms_silu_cl   : RMS_norm (vae.py:39-54: F.normalize over channels * sqrt(C) * gamma) + optional SiLU (:195,197)
//   wf_softmax_rows  : softmax of the single-head mid-block attention scores (vae.py:252-256)
//   wf_transpose_bf16: V -> V^T for the P.V GEMM of that attention
//   wf_ncthw_to_cl / wf_cl_to_ncthw : layout change at the VAE boundary ([C,T,H,W] <-> [T,H,W,C])
#include "common.h"
#include "mfma.h"

using namespace wf;

namespace {

// fp16 operand formats (round 4): the producers below write fp16 instead of bf16 when `f16` is set (wave-uniform).  fp16 has 11
// significand bits but only 5 exponent bits: a value beyond +-65504 cannot be represented -- the producers raise a sticky device flag
// (wf_f16_overflow_flag) that the host checks after every VAE call and turns into an error (never a silent inf / NaN).
__device__ unsigned int g_f16_overflow;
__device__ __forceinline__ float r16(int f16, float x) {  // x rounded through the 16-bit operand type
  if (f16) return (float)(_Float16)x;
  return rbf(x);
}
__device__ __forceinline__ uint16_t f32_to_f16(float x) {
  union {
    _Float16 h;
    uint16_t u;
  } c;
  c.h = (_Float16)x;
  return c.u;
}
__device__ __forceinline__ void note_range(int f16, const float (&y)[4]) {
  if (f16 && !(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))) <= 65504.0f)) g_f16_overflow = 1u;  // NaN too
}

// RMS_norm (+ SiLU) over the C f32 channels of each pixel (C % 4 == 0, C <= 1024).  A pixel is handled by a group of G lanes
// (G = 8 / 16 / 32 / 64 for C <= 128 / 256 / 512 / 1024: three or four float4 per lane), i.e. 64 / G pixels per wave: with one wave per
// pixel only 24 of 64 lanes worked at C = 96 and the kernel ran at 2.4 TB/s.  The sum of squares is reduced inside the group by
// xor-shuffles (fixed order: bit-identical wherever the pixel sits).
// SPLIT: the bf16 output is the three-term fp32-class operand [hi | lo | hi] (3C channels per pixel), see k_split3.
template <int G, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_rms_silu(const float* __restrict__ x, const float* __restrict__ gamma,
                                                  uint16_t* __restrict__ out_bf16, float* __restrict__ out_f32, int C,
                                                  float scale, int silu, size_t npix, int Wrow = 0, int halo_rows = 0, int f16 = 0) {
  constexpr int PPW = 64 / G;  // pixels per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane & (G - 1), q = lane / G;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = C >> 2;
  float4 g4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = sub + G * i;
    g4[i] = id < nvec ? reinterpret_cast<const float4*>(gamma)[id] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (size_t p0 = wave * PPW; p0 < npix; p0 += nwaves * PPW) {
    const size_t p = p0 + q;
    const bool live = p < npix;
    const float4* xr = reinterpret_cast<const float4*>(x + (live ? p : 0) * C);
    float4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = sub + G * i;
      if (id < nvec) {
        v[i] = xr[id];
        ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = scale / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps = 1e-12
    if (!live) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = sub + G * i;
      if (id < nvec) {
        const float4 g = g4[i];
        float y[4] = {v[i].x * inv * g.x, v[i].y * inv * g.y, v[i].z * inv * g.z, v[i].w * inv * g.w};
        if (silu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) y[k] = y[k] / (1.0f + (SPLIT ? expf(-y[k]) : __expf(-y[k])));
        }
        note_range(f16, y);
        if (Wrow > 0) {
          // slice-major operand for wf_conv3d_333 (layout 1): [row = p / W][stored slice][x][16]; SPLIT stores [hi | lo] slices
          size_t row = p / (size_t)Wrow;
          const int xx = (int)(p - row * (size_t)Wrow);
          // halo_rows = rows per frame Hs of a row slab: the output is the halo-padded operand [T][Hs + 2][...], row (t, y) -> (t, y + 1)
          if (halo_rows > 0) row += 2 * (row / (size_t)halo_rows) + 1;
          const int S = C >> 4, stot = SPLIT ? 2 * S : S;
          uint16_t* o = out_bf16 + ((row * stot + (id >> 2)) * (size_t)Wrow + xx) * 16 + (id & 3) * 4;
          const u32x2 hi = {pack16x2(f16, y[0], y[1]), pack16x2(f16, y[2], y[3])};
          *reinterpret_cast<u32x2*>(o) = hi;
          if constexpr (SPLIT) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = y[k] - r16(f16, y[k]);
            const u32x2 lo = {pack16x2(f16, r[0], r[1]), pack16x2(f16, r[2], r[3])};
            *reinterpret_cast<u32x2*>(o + (size_t)S * Wrow * 16) = lo;
          }
        } else if constexpr (SPLIT) {
          float r[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) r[k] = y[k] - r16(f16, y[k]);
          const u32x2 hi = {pack16x2(f16, y[0], y[1]), pack16x2(f16, y[2], y[3])};
          const u32x2 lo = {pack16x2(f16, r[0], r[1]), pack16x2(f16, r[2], r[3])};
          uint16_t* o = out_bf16 + p * (size_t)(3 * C);
          reinterpret_cast<u32x2*>(o)[id] = hi;
          reinterpret_cast<u32x2*>(o + C)[id] = lo;
          reinterpret_cast<u32x2*>(o + 2 * C)[id] = hi;
        } else if (out_bf16) {
          u32x2 pk = {pack16x2(f16, y[0], y[1]), pack16x2(f16, y[2], y[3])};
          reinterpret_cast<u32x2*>(out_bf16 + p * C)[id] = pk;
        }
        if (out_f32) reinterpret_cast<float4*>(out_f32 + p * C)[id] = make_float4(y[0], y[1], y[2], y[3]);
      }
    }
  }
}

// softmax over each row of S [M, N] f32 (row stride lds) * scale -> P bf16 [M, ldp], columns N..ldp zero-filled
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ S, int lds, uint16_t* __restrict__ P, int ldp,
                                                      int N, float scale, int f16) {
  __shared__ float sm[8];
  const size_t row = blockIdx.x;
  const float* s = S + row * lds;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, s[i]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float sum = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) sum += __expf((s[i] - mx) * scale);
  sum = wave_sum(sum);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  sum = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  const float inv = 1.0f / sum;
  uint16_t* p = P + row * ldp;
  for (int i = threadIdx.x; i < ldp; i += 256) {
    const float v = i < N ? __expf((s[i] - mx) * scale) * inv : 0.f;   // in [0, 1]: inside the fp16 range
    p[i] = f16 ? f32_to_f16(v) : f32_to_bf16(v);
  }
}

// fp32-class operands for the bf16 matrix cores: x = hi + lo, hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in fp32; the
// dropped tail is <= 2^-17 |x|).  src f32 [rows, C] -> dst bf16 [rows, 3C]: side 0 (activation) [hi | lo | hi], side 1 (weight)
// [hi | hi | lo], so that one K-concatenated bf16 MFMA contraction of a side-0 row with a side-1 row is hi.hi + lo.hi + hi.lo.
__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ src, long ld_src, uint16_t* __restrict__ dst, long ld_dst,
                                                size_t rows, int C, int side, int f16) {
  const int nvec = C >> 2;
  const size_t n = rows * (size_t)nvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / nvec;
    const int id = (int)(i % nvec);
    const float4 v = reinterpret_cast<const float4*>(src + r * ld_src)[id];
    const float y[4] = {v.x, v.y, v.z, v.w};
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = y[k] - r16(f16, y[k]);
    note_range(f16, y);
    const u32x2 hi = {pack16x2(f16, y[0], y[1]), pack16x2(f16, y[2], y[3])};
    const u32x2 lo = {pack16x2(f16, t[0], t[1]), pack16x2(f16, t[2], t[3])};
    uint16_t* o = dst + r * ld_dst;
    reinterpret_cast<u32x2*>(o)[id] = hi;
    reinterpret_cast<u32x2*>(o + C)[id] = side ? hi : lo;
    reinterpret_cast<u32x2*>(o + 2 * C)[id] = side ? lo : hi;
  }
}

// One-term fp16 operand (the VAE's "fp16" mode: 11 significand bits, what a TF32 convolution keeps of an fp32 operand): src f32
// [rows, C] -> dst fp16 [rows, C], with the range flag of the three-term producers.
__global__ __launch_bounds__(256) void k_cast_f16(const float* __restrict__ src, long ld_src, uint16_t* __restrict__ dst, long ld_dst,
                                                  size_t rows, int C) {
  const int nvec = C >> 2;
  const size_t n = rows * (size_t)nvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / nvec;
    const int id = (int)(i % nvec);
    const float4 v = reinterpret_cast<const float4*>(src + r * ld_src)[id];
    const float y[4] = {v.x, v.y, v.z, v.w};
    note_range(1, y);
    const u32x2 hi = {pack16x2(1, y[0], y[1]), pack16x2(1, y[2], y[3])};
    reinterpret_cast<u32x2*>(dst + r * ld_dst)[id] = hi;
  }
}

// The operand producers above writing straight into a HALO-PADDED destination (row slabs of the sharded VAE: [T][Hs + 2][W][..], the
// slab's own rows at 1 .. Hs of every frame): source row r lands at destination row lead + r + (r / group) * gap.  fmt 0 / 1: three-term
// bf16 / fp16 split (side as k_split3), fmt 2: one-term fp16, fmt 3: one-term bf16.  Same conversions as k_split3 / k_cast_f16.
__global__ __launch_bounds__(256) void k_operand_rows(const float* __restrict__ src, long ld_src, uint16_t* __restrict__ dst, long ld_dst,
                                                      size_t rows, int C, int fmt, int side, size_t group, size_t gap, size_t lead) {
  const int nvec = C >> 2;
  const size_t n = rows * (size_t)nvec;
  const int f16 = fmt == 1 || fmt == 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / nvec;
    const int id = (int)(i % nvec);
    const float4 v = reinterpret_cast<const float4*>(src + r * ld_src)[id];
    const float y[4] = {v.x, v.y, v.z, v.w};
    note_range(f16, y);
    const u32x2 hi = {pack16x2(f16, y[0], y[1]), pack16x2(f16, y[2], y[3])};
    uint16_t* o = dst + (lead + r + (group ? (r / group) * gap : 0)) * ld_dst;
    reinterpret_cast<u32x2*>(o)[id] = hi;
    if (fmt < 2) {
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = y[k] - r16(f16, y[k]);
      const u32x2 lo = {pack16x2(f16, t[0], t[1]), pack16x2(f16, t[2], t[3])};
      reinterpret_cast<u32x2*>(o + C)[id] = side ? hi : lo;
      reinterpret_cast<u32x2*>(o + 2 * C)[id] = side ? lo : hi;
    }
  }
}

// softmax over each row of S [M, N] f32 * scale -> P f32 [M, ldp] (columns N..ldp zero): the fp32-class VAE keeps P in f32 and splits it
__global__ __launch_bounds__(256) void k_softmax_rows_f32(const float* __restrict__ S, int lds, float* __restrict__ P, int ldp, int N,
                                                          float scale) {
  __shared__ float sm[8];
  const size_t row = blockIdx.x;
  const float* s = S + row * lds;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, s[i]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float sum = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) sum += expf((s[i] - mx) * scale);
  sum = wave_sum(sum);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  sum = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  float* p = P + row * ldp;
  for (int i = threadIdx.x; i < ldp; i += 256) p[i] = i < N ? expf((s[i] - mx) * scale) / sum : 0.f;
}

// in [R, ld_in] (first C columns) -> out [C, ld_out], columns R..ld_out zero-filled.  32x32 tiles through LDS.
template <typename E>
__global__ void k_transpose(const E* __restrict__ in, int ld_in, E* __restrict__ out, int ld_out, int R, int C) {
  __shared__ E tile[32][33 + (sizeof(E) == 2)];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int k = ty; k < 32; k += 8) {
    int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : (E)0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int c = c0 + k, r = r0 + tx;
    if (c < C && r < ld_out) out[(size_t)c * ld_out + r] = tile[tx][k];
  }
}

// [C, N] -> [N, Cpad] (N = T*H*W; channels C..Cpad zero), in f32 -> out f32 or bf16
__global__ void k_to_cl(const float* __restrict__ in, float* __restrict__ of, uint16_t* __restrict__ ob, int C, int Cpad, size_t N, int f16) {
  const size_t n = N * Cpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const size_t p = i / Cpad;
    const float v = c < C ? in[(size_t)c * N + p] : 0.f;
    if (of) of[i] = v;
    if (ob) {
      if (f16 && !(fabsf(v) <= 65504.0f)) g_f16_overflow = 1u;
      ob[i] = f16 ? f32_to_f16(v) : f32_to_bf16(v);
    }
  }
}
// [N, ld] f32 (first C channels) -> [C, N] f32, optional clamp
__global__ void k_from_cl(const float* __restrict__ in, float* __restrict__ out, int C, int ld, size_t N, float clampv) {
  const size_t n = N * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % N;
    const int c = (int)(i / N);
    float v = in[p * ld + c];
    if (clampv > 0.f) v = fminf(fmaxf(v, -clampv), clampv);
    out[i] = v;
  }
}

}  // namespace

static int rms_silu_impl(const float* x, const float* gamma, void* out_bf16, float* out_f32, size_t npix, int C, int silu, void* stream,
                         int f16) {
  WF_CHECK_ARG(x && gamma && (out_bf16 || out_f32), "wf_rms_silu_cl: null pointer");
  WF_CHECK_ARG(C % 4 == 0 && C > 0 && C <= 1024, "wf_rms_silu_cl: C=%d must be a multiple of 4 and <= 1024", C);
  if (npix == 0) return WF_OK;
  const int G = C <= 128 ? 8 : (C <= 256 ? 16 : (C <= 512 ? 32 : 64));
  size_t blocks = (npix + (size_t)(4 * (64 / G)) - 1) / (size_t)(4 * (64 / G));
  if (blocks > 16384) blocks = 16384;
  const float sc = sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
  if (G == 8)
    hipLaunchKernelGGL(k_rms_silu<8>, dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, (uint16_t*)out_bf16, out_f32, C, sc, silu, npix, 0, 0, f16);
  else if (G == 16)
    hipLaunchKernelGGL(k_rms_silu<16>, dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, (uint16_t*)out_bf16, out_f32, C, sc, silu, npix, 0, 0, f16);
  else if (G == 32)
    hipLaunchKernelGGL(k_rms_silu<32>, dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, (uint16_t*)out_bf16, out_f32, C, sc, silu, npix, 0, 0, f16);
  else
    hipLaunchKernelGGL(k_rms_silu<64>, dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, (uint16_t*)out_bf16, out_f32, C, sc, silu, npix, 0, 0, f16);
  WF_LAUNCH_CHECK("wf_rms_silu_cl");
  return WF_OK;
}
extern "C" int wf_rms_silu_cl(const float* x, const float* gamma, void* out_bf16, float* out_f32, size_t npix, int C, int silu,
                              void* stream) {
  return rms_silu_impl(x, gamma, out_bf16, out_f32, npix, C, silu, stream, 0);
}
extern "C" int wf_rms_silu_cl_f16(const float* x, const float* gamma, void* out_f16, float* out_f32, size_t npix, int C, int silu,
                                  void* stream) {
  return rms_silu_impl(x, gamma, out_f16, out_f32, npix, C, silu, stream, 1);
}

extern "C" int wf_softmax_rows(const float* S, int lds, void* P, int ldp, int M, int N, float scale, void* stream) {
  WF_CHECK_ARG(S && P, "wf_softmax_rows: null pointer");
  WF_CHECK_ARG(N > 0 && ldp >= N && lds >= N, "wf_softmax_rows: bad sizes");
  if (M == 0) return WF_OK;
  hipLaunchKernelGGL(k_softmax_rows, dim3(M), dim3(256), 0, (hipStream_t)stream, S, lds, (uint16_t*)P, ldp, N, scale, 0);
  WF_LAUNCH_CHECK("wf_softmax_rows");
  return WF_OK;
}
extern "C" int wf_softmax_rows_f16(const float* S, int lds, void* P, int ldp, int M, int N, float scale, void* stream) {
  WF_CHECK_ARG(S && P, "wf_softmax_rows_f16: null pointer");
  WF_CHECK_ARG(N > 0 && ldp >= N && lds >= N, "wf_softmax_rows_f16: bad sizes");
  if (M == 0) return WF_OK;
  hipLaunchKernelGGL(k_softmax_rows, dim3(M), dim3(256), 0, (hipStream_t)stream, S, lds, (uint16_t*)P, ldp, N, scale, 1);
  WF_LAUNCH_CHECK("wf_softmax_rows_f16");
  return WF_OK;
}

extern "C" int wf_transpose_bf16(const void* in, int ld_in, void* out, int ld_out, int R, int C, void* stream) {
  WF_CHECK_ARG(in && out, "wf_transpose_bf16: null pointer");
  WF_CHECK_ARG(ld_out >= R && ld_in >= C, "wf_transpose_bf16: bad leading dimensions");
  if (R == 0 || C == 0) return WF_OK;
  hipLaunchKernelGGL(k_transpose<uint16_t>, dim3((C + 31) / 32, (ld_out + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)in, ld_in, (uint16_t*)out, ld_out, R, C);
  WF_LAUNCH_CHECK("wf_transpose_bf16");
  return WF_OK;
}

extern "C" int wf_transpose_f32(const float* in, int ld_in, float* out, int ld_out, int R, int C, void* stream) {
  WF_CHECK_ARG(in && out, "wf_transpose_f32: null pointer");
  WF_CHECK_ARG(ld_out >= R && ld_in >= C, "wf_transpose_f32: bad leading dimensions");
  if (R == 0 || C == 0) return WF_OK;
  hipLaunchKernelGGL(k_transpose<float>, dim3((C + 31) / 32, (ld_out + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, ld_in, out,
                     ld_out, R, C);
  WF_LAUNCH_CHECK("wf_transpose_f32");
  return WF_OK;
}

extern "C" int wf_softmax_rows_f32(const float* S, int lds, float* P, int ldp, int M, int N, float scale, void* stream) {
  WF_CHECK_ARG(S && P, "wf_softmax_rows_f32: null pointer");
  WF_CHECK_ARG(N > 0 && ldp >= N && lds >= N, "wf_softmax_rows_f32: bad sizes");
  if (M == 0) return WF_OK;
  hipLaunchKernelGGL(k_softmax_rows_f32, dim3(M), dim3(256), 0, (hipStream_t)stream, S, lds, P, ldp, N, scale);
  WF_LAUNCH_CHECK("wf_softmax_rows_f32");
  return WF_OK;
}

static int split_x3_impl(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, int side, void* stream, int f16) {
  WF_CHECK_ARG(src && dst, "wf_split_bf16x3: null pointer");
  WF_CHECK_ARG(C > 0 && C % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= C && ld_dst >= 3L * C,
               "wf_split_bf16x3: C=%d ld_src=%ld ld_dst=%ld (C, strides multiples of 4; ld_dst >= 3C)", C, (long)ld_src, (long)ld_dst);
  WF_CHECK_ARG(side == 0 || side == 1, "wf_split_bf16x3: side must be 0 (activation) or 1 (weight)");
  if (rows == 0) return WF_OK;
  hipLaunchKernelGGL(k_split3, dim3(grid_for(rows * (size_t)(C / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     (uint16_t*)dst, ld_dst, rows, C, side, f16);
  WF_LAUNCH_CHECK("wf_split_bf16x3");
  return WF_OK;
}
extern "C" int wf_split_bf16x3(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, int side, void* stream) {
  return split_x3_impl(src, ld_src, dst, ld_dst, rows, C, side, stream, 0);
}
extern "C" int wf_split_f16x3(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, int side, void* stream) {
  return split_x3_impl(src, ld_src, dst, ld_dst, rows, C, side, stream, 1);
}
extern "C" int wf_cast_f16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, void* stream) {
  WF_CHECK_ARG(src && dst, "wf_cast_f16: null pointer");
  WF_CHECK_ARG(C > 0 && C % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= C && ld_dst >= C,
               "wf_cast_f16: C=%d ld_src=%ld ld_dst=%ld (C, strides multiples of 4)", C, (long)ld_src, (long)ld_dst);
  if (rows == 0) return WF_OK;
  hipLaunchKernelGGL(k_cast_f16, dim3(grid_for(rows * (size_t)(C / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     (uint16_t*)dst, ld_dst, rows, C);
  WF_LAUNCH_CHECK("wf_cast_f16");
  return WF_OK;
}
extern "C" int wf_operand_rows(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, int fmt, int side,
                               size_t group_rows, size_t gap_rows, size_t lead_rows, void* stream) {
  WF_CHECK_ARG(src && dst, "wf_operand_rows: null pointer");
  WF_CHECK_ARG(fmt >= 0 && fmt <= 3 && (side == 0 || side == 1), "wf_operand_rows: fmt %d (0 bf16x3, 1 f16x3, 2 f16, 3 bf16) / side %d", fmt, side);
  WF_CHECK_ARG(C > 0 && C % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= C && ld_dst >= (fmt < 2 ? 3L : 1L) * C,
               "wf_operand_rows: C=%d ld_src=%ld ld_dst=%ld (C, strides multiples of 4; ld_dst >= the operand's width)", C, (long)ld_src,
               (long)ld_dst);
  if (rows == 0) return WF_OK;
  hipLaunchKernelGGL(k_operand_rows, dim3(grid_for(rows * (size_t)(C / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     (uint16_t*)dst, ld_dst, rows, C, fmt, side, group_rows, gap_rows, lead_rows);
  WF_LAUNCH_CHECK("wf_operand_rows");
  return WF_OK;
}
// Sticky range flag of the fp16 producers: *out = 1 if any value beyond the fp16 range (or a NaN) was converted since the last reset.
// Synchronous (a 4-byte device -> host copy on the NULL stream after `stream` has drained).
extern "C" int wf_f16_overflow_flag(int* out, int reset, void* stream) {
  WF_CHECK_ARG(out, "wf_f16_overflow_flag: null pointer");
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return WF_EHIP;
  unsigned int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16_overflow), sizeof(v)) != hipSuccess) return WF_EHIP;
  *out = (int)v;
  if (reset && v) {
    const unsigned int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_f16_overflow), &z, sizeof(z)) != hipSuccess) return WF_EHIP;
  }
  return WF_OK;
}

// The same flag, asynchronously: a 4-byte copy into `pinned_host_out` (page-locked host memory owned by the caller) queued on `stream`
// behind the producers launched so far.  No synchronisation: the caller reads the word after an event it records behind this call.
extern "C" int wf_f16_overflow_flag_async(int* pinned_host_out, void* stream) {
  WF_CHECK_ARG(pinned_host_out, "wf_f16_overflow_flag_async: null pointer");
  if (hipMemcpyFromSymbolAsync(pinned_host_out, HIP_SYMBOL(g_f16_overflow), sizeof(unsigned int), 0, hipMemcpyDeviceToHost,
                               (hipStream_t)stream) != hipSuccess)
    return WF_EHIP;
  return WF_OK;
}

static int rms_silu_x3_impl(const float* x, const float* gamma, void* out_x3, size_t npix, int C, int silu, void* stream, int f16) {
  WF_CHECK_ARG(x && gamma && out_x3, "wf_rms_silu_cl_x3: null pointer");
  WF_CHECK_ARG(C % 4 == 0 && C > 0 && C <= 1024, "wf_rms_silu_cl_x3: C=%d must be a multiple of 4 and <= 1024", C);
  if (npix == 0) return WF_OK;
  const int G = C <= 128 ? 8 : (C <= 256 ? 16 : (C <= 512 ? 32 : 64));
  size_t blocks = (npix + (size_t)(4 * (64 / G)) - 1) / (size_t)(4 * (64 / G));
  if (blocks > 16384) blocks = 16384;
  const float sc = sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
  uint16_t* o = (uint16_t*)out_x3;
  if (G == 8)
    hipLaunchKernelGGL((k_rms_silu<8, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, o, (float*)nullptr, C, sc, silu, npix, 0, 0, f16);
  else if (G == 16)
    hipLaunchKernelGGL((k_rms_silu<16, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, o, (float*)nullptr, C, sc, silu, npix, 0, 0, f16);
  else if (G == 32)
    hipLaunchKernelGGL((k_rms_silu<32, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, o, (float*)nullptr, C, sc, silu, npix, 0, 0, f16);
  else
    hipLaunchKernelGGL((k_rms_silu<64, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, o, (float*)nullptr, C, sc, silu, npix, 0, 0, f16);
  WF_LAUNCH_CHECK("wf_rms_silu_cl_x3");
  return WF_OK;
}
extern "C" int wf_rms_silu_cl_x3(const float* x, const float* gamma, void* out_x3, size_t npix, int C, int silu, void* stream) {
  return rms_silu_x3_impl(x, gamma, out_x3, npix, C, silu, stream, 0);
}
extern "C" int wf_rms_silu_cl_x3_f16(const float* x, const float* gamma, void* out_x3, size_t npix, int C, int silu, void* stream) {
  return rms_silu_x3_impl(x, gamma, out_x3, npix, C, silu, stream, 1);
}

static int ncthw_to_cl_impl(const float* in, float* out_f32, void* out_16, int C, int Cpad, size_t N, void* stream, int f16) {
  WF_CHECK_ARG(in && (out_f32 || out_16), "wf_ncthw_to_cl: null pointer");
  WF_CHECK_ARG(Cpad >= C && C > 0, "wf_ncthw_to_cl: Cpad (%d) must be >= C (%d)", Cpad, C);
  if (N == 0) return WF_OK;
  hipLaunchKernelGGL(k_to_cl, dim3(grid_for(N * Cpad, 256, 8192)), dim3(256), 0, (hipStream_t)stream, in, out_f32,
                     (uint16_t*)out_16, C, Cpad, N, f16);
  WF_LAUNCH_CHECK("wf_ncthw_to_cl");
  return WF_OK;
}
extern "C" int wf_ncthw_to_cl(const float* in, float* out_f32, void* out_bf16, int C, int Cpad, size_t N, void* stream) {
  return ncthw_to_cl_impl(in, out_f32, out_bf16, C, Cpad, N, stream, 0);
}
extern "C" int wf_ncthw_to_cl_f16(const float* in, float* out_f32, void* out_f16, int C, int Cpad, size_t N, void* stream) {
  return ncthw_to_cl_impl(in, out_f32, out_f16, C, Cpad, N, stream, 1);
}

extern "C" int wf_cl_to_ncthw(const float* in, float* out, int C, int ld, size_t N, float clamp, void* stream) {
  WF_CHECK_ARG(in && out, "wf_cl_to_ncthw: null pointer");
  WF_CHECK_ARG(ld >= C && C > 0, "wf_cl_to_ncthw: ld (%d) must be >= C (%d)", ld, C);
  if (N == 0) return WF_OK;
  hipLaunchKernelGGL(k_from_cl, dim3(grid_for(N * C, 256, 8192)), dim3(256), 0, (hipStream_t)stream, in, out, C, ld, N, clamp);
  WF_LAUNCH_CHECK("wf_cl_to_ncthw");
  return WF_OK;
}

static int rms_silu_blocked_impl(const float* x, const float* gamma, void* out, size_t npix, int C, int silu, int W, int split,
                                 int halo_rows, void* stream, int f16) {
  WF_CHECK_ARG(halo_rows >= 0 && (halo_rows == 0 || (npix / (size_t)(W > 0 ? W : 1)) % (size_t)halo_rows == 0),
               "wf_rms_silu_cl_blocked: halo_rows=%d must divide the number of rows", halo_rows);
  WF_CHECK_ARG(x && gamma && out, "wf_rms_silu_cl_blocked: null pointer");
  WF_CHECK_ARG(C % 16 == 0 && C > 0 && C <= 1024, "wf_rms_silu_cl_blocked: C=%d must be a multiple of 16 and <= 1024", C);
  WF_CHECK_ARG(W > 0 && npix % (size_t)W == 0, "wf_rms_silu_cl_blocked: npix must be whole rows of W=%d pixels", W);
  if (npix == 0) return WF_OK;
  const int G = C <= 128 ? 8 : (C <= 256 ? 16 : (C <= 512 ? 32 : 64));
  size_t blocks = (npix + (size_t)(4 * (64 / G)) - 1) / (size_t)(4 * (64 / G));
  if (blocks > 16384) blocks = 16384;
  const float sc = sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
  uint16_t* o = (uint16_t*)out;
  float* nof = nullptr;
#define WF_RMS_BLOCKED(GG)                                                                                                          \
  do {                                                                                                                              \
    if (split)                                                                                                                      \
      hipLaunchKernelGGL((k_rms_silu<GG, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, o, nof, C, sc, silu, npix, W, halo_rows, f16);  \
    else                                                                                                                            \
      hipLaunchKernelGGL((k_rms_silu<GG, false>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, o, nof, C, sc, silu, npix, W, halo_rows, f16); \
  } while (0)
  if (G == 8) WF_RMS_BLOCKED(8);
  else if (G == 16) WF_RMS_BLOCKED(16);
  else if (G == 32) WF_RMS_BLOCKED(32);
  else WF_RMS_BLOCKED(64);
#undef WF_RMS_BLOCKED
  WF_LAUNCH_CHECK("wf_rms_silu_cl_blocked");
  return WF_OK;
}
extern "C" int wf_rms_silu_cl_blocked(const float* x, const float* gamma, void* out, size_t npix, int C, int silu, int W, int split,
                                      int halo_rows, void* stream) {
  return rms_silu_blocked_impl(x, gamma, out, npix, C, silu, W, split, halo_rows, stream, 0);
}
extern "C" int wf_rms_silu_cl_blocked_f16(const float* x, const float* gamma, void* out, size_t npix, int C, int silu, int W, int split,
                                          int halo_rows, void* stream) {
  return rms_silu_blocked_impl(x, gamma, out, npix, C, silu, W, split, halo_rows, stream, 1);
}
