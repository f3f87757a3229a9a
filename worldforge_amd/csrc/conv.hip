// Implicit-GEMM 3D convolution on MFMA for the Wan 3D causal VAE (wan/modules/vae.py), channels-last activations.
//
// Replaces CausalConv3d (vae.py:17-36) in ResidualBlock (:186-220) / Resample.time_conv (:84-96), the Conv2d of Resample
// (:76-94, optionally fused with the nearest-exact 2x Upsample :57-63 or the ZeroPad2d((0,1,0,1)) + stride 2), and keeps the
// feat_cache semantics (:202-217) by processing the WHOLE frame sequence at once with causal zero padding in time -- which
// is what the chunked cache computes, one latent frame at a time, in the reference (derivation in DESIGN.md "VAE").
// MFMA-bound: 2 * pixels * Cout * taps * Cin flop.
//
//   out[t,y,x,co] = bias[co] + sum_{dt,dy,dx,ci} in[t*st + dt - pt, y*ss + dy - ph, x*ss + dx - pw, ci] * w[co][dt][dy][dx][ci]
//   (+ residual[t,y,x,co]);  out-of-range input coordinates read as zero;  up2: the input is read through a nearest 2x
//   spatial upsample (source pixel = coordinate >> 1).
//
// GEMM view: M = To*Ho*Wo output pixels (MFMA B operand, rows of 32 input channels gathered per tap), N = Cout (MFMA A operand,
// weight rows), K = taps * Cin walked in BK = 32 slices (all VAE widths 96/192/384/768 are multiples of 96 = 3 * 32).
// Workgroup tile 256 pixels x 96 output channels, 4 waves, wave = 64 pixels x 96 channels = 2 x 3 MFMA 32x32x16 tiles;
// LDS double buffer, 64-byte rows, chunk c of row r stored at c ^ ((r >> 2) & 3) (conflict-free b128 reads and writes);
// global -> VGPR -> LDS staging with the next slice's gathers issued before the current slice's MFMAs.
#include "common.h"
#include "mfma.h"

using namespace wf;

namespace {

constexpr int CBM = 256;  // pixels per tile
constexpr int CBN = 96;   // output channels per tile
constexpr int CBK = 32;
constexpr int CNT = 256;
constexpr int XT_BYTES = CBM * CBK * 2;  // 16 KiB
constexpr int WT_BYTES = CBN * CBK * 2;  // 6 KiB
constexpr int CBUF = XT_BYTES + WT_BYTES;

struct ConvArgs {
  const uint16_t* in;   // [Ti, Hi, Wi, Cin] bf16
  const uint16_t* w;    // [Cout, taps, Cin] bf16
  const float* bias;    // [Cout] or null
  const float* resid;   // [To, Ho, Wo, Cout] f32 or null
  float* out_f32;       // may be null
  uint16_t* out_bf16;   // may be null
  int Ti, Hi, Wi, Cin;  // source tensor dims (before the optional 2x upsample)
  int To, Ho, Wo, Cout;
  int kt, kh, kw;
  int st, ss;      // temporal / spatial stride
  int pt, ph, pw;  // temporal (front) / top / left zero padding (ph may be negative: input slab carries halo rows)
  int up2;         // read input through nearest 2x spatial upsample
  int tsplit;      // Resample 'upsample3d' (vae.py:134-137): channel half h of output frame t goes to frame 1 + 2*t + h
  int silu_out;    // unused (reserved)
};

__global__ __launch_bounds__(CNT, 2) void k_conv(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long M = (long)a.To * a.Ho * a.Wo;
  const long m0 = (long)blockIdx.x * CBM;
  const int n0 = blockIdx.y * CBN;
  const int Hs = a.up2 ? a.Hi * 2 : a.Hi, Ws = a.up2 ? a.Wi * 2 : a.Wi;  // logical (upsampled) input extent
  const int taps = a.kt * a.kh * a.kw;
  const int kc_per_tap = a.Cin / CBK;
  const int nk = taps * kc_per_tap;

  // ---- staging geometry ---------------------------------------------------------------------------------------------
  // X tile: 256 rows x 4 chunks = 1024 chunks -> 4 per thread: rows (tid >> 2) + 64*i, chunk tid & 3
  // W tile:  96 rows x 4 chunks =  384 chunks -> ids tid, tid + 256 (second only for tid < 128)
  const int xc = tid & 3;
  int pt_[4], py_[4], px_[4];
  bool pvalid[4];
  int xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (tid >> 2) + 64 * i;
    long m = m0 + row;
    pvalid[i] = m < M;
    long mm = pvalid[i] ? m : 0;
    px_[i] = (int)(mm % a.Wo);
    py_[i] = (int)((mm / a.Wo) % a.Ho);
    pt_[i] = (int)(mm / ((long)a.Wo * a.Ho));
    xoff[i] = row * 64 + ((xc ^ ((row >> 2) & 3)) << 4);
  }
  const uint16_t* wp[2];
  int woff[2];
  bool wvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int id = tid + 256 * i;
    int row = id >> 2, c = id & 3;
    wvalid[i] = id < 384;
    int co = min(n0 + row, a.Cout - 1);
    wp[i] = a.w + (size_t)co * taps * a.Cin + c * 8;
    woff[i] = XT_BYTES + row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
  }
  u32x4 rX[4], rW[2];
  auto gload = [&](int kt_) {
    const int tap = kt_ / kc_per_tap;
    const int cin0 = (kt_ - tap * kc_per_tap) * CBK;
    const int dx = tap % a.kw, dy = (tap / a.kw) % a.kh, dt = tap / (a.kw * a.kh);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ti = pt_[i] * a.st + dt - a.pt;
      int yi = py_[i] * a.ss + dy - a.ph;
      int xi = px_[i] * a.ss + dx - a.pw;
      bool ok = pvalid[i] && ti >= 0 && ti < a.Ti && yi >= 0 && yi < Hs && xi >= 0 && xi < Ws;
      if (a.up2) {
        yi >>= 1;
        xi >>= 1;
      }
      u32x4 z = {0u, 0u, 0u, 0u};
      const uint16_t* p = a.in + (((size_t)ti * a.Hi + yi) * a.Wi + xi) * a.Cin + cin0 + xc * 8;
      rX[i] = ok ? *reinterpret_cast<const u32x4*>(p) : z;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 z = {0u, 0u, 0u, 0u};
      rW[i] = wvalid[i] ? *reinterpret_cast<const u32x4*>(wp[i] + (size_t)tap * a.Cin + cin0) : z;
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = smem + buf * CBUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + xoff[i]) = rX[i];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (wvalid[i]) *reinterpret_cast<u32x4*>(base + woff[i]) = rW[i];
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int fxoff[2], fxsw[2], fwoff[3], fwsw[3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = wid * 64 + j * 32 + l31;
    fxoff[j] = row * 64;
    fxsw[j] = (row >> 2) & 3;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int row = i * 32 + l31;
    fwoff[i] = XT_BYTES + row * 64;
    fwsw[i] = (row >> 2) & 3;
  }

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt_ = 0; kt_ < nk; ++kt_) {
    const int buf = kt_ & 1;
    const unsigned char* base = smem + buf * CBUF;
    if (kt_ + 1 < nk) gload(kt_ + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = 2 * s + hi;
      bf16x8 fw[3], fx[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) fw[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(base + fwoff[i] + ((c ^ fwsw[i]) << 4)));
#pragma unroll
      for (int j = 0; j < 2; ++j) fx[j] = as_bf16x8(*reinterpret_cast<const u32x4*>(base + fxoff[j] + ((c ^ fxsw[j]) << 4)));
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]);
    }
    if (kt_ + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------------
  const int chalf = a.Cout >> 1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long m = m0 + wid * 64 + j * 32 + l31;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = n0 + i * 32 + 8 * g + 4 * hi;
        if (co >= a.Cout) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q];
        if (a.bias) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += bb[q];
        }
        size_t o;
        if (a.tsplit) {
          const long hw = (long)a.Ho * a.Wo;
          const long t = m / hw, pix = m - t * hw;
          const int half = co >= chalf ? 1 : 0;
          o = ((size_t)(1 + 2 * t + half) * hw + pix) * chalf + (co - half * chalf);
        } else {
          o = (size_t)m * a.Cout + co;
        }
        if (a.resid) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(a.resid + o);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rr[q];
        }
        if (a.out_f32) {
          f32x4 ov = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(a.out_f32 + o) = ov;
        }
        if (a.out_bf16) {
          u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(a.out_bf16 + o) = pk;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Direct (VALU) convolution for the few layers whose channel counts do not fill an MFMA tile: encoder conv1 (3 -> 96),
// decoder conv1 (16 -> 384), decoder head (96 -> 3), encoder head (384 -> 32), quant convs (1x1x1).  vae.py:288, 392, 421, 316.
// in: channels-last f32 or bf16; w: [taps][Cin][Cout] f32 (Cout fastest, so consecutive lanes read consecutive weights).
// One thread per (pixel, co).
// ------------------------------------------------------------------------------------------------------------------
struct SmallConvArgs {
  const void* in;
  int in_bf16;
  const float* w;
  const float* bias;
  float* out_f32;
  uint16_t* out_bf16;
  int Ti, Hi, Wi, Cin, To, Ho, Wo, Cout, kt, kh, kw, st, ss, pt, ps;
  float clamp;  // > 0: clamp output to [-clamp, clamp]  (decoder output, autoencoder_kl_wan.py:1222)
};
__global__ void k_conv_small(SmallConvArgs a) {
  const size_t n = (size_t)a.To * a.Ho * a.Wo * a.Cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % a.Cout);
    const size_t m = i / a.Cout;
    const int x = (int)(m % a.Wo), y = (int)((m / a.Wo) % a.Ho), t = (int)(m / ((size_t)a.Wo * a.Ho));
    float acc = a.bias ? a.bias[co] : 0.f;
    for (int dt = 0; dt < a.kt; ++dt) {
      const int ti = t * a.st + dt - a.pt;
      if (ti < 0 || ti >= a.Ti) continue;
      for (int dy = 0; dy < a.kh; ++dy) {
        const int yi = y * a.ss + dy - a.ps;
        if (yi < 0 || yi >= a.Hi) continue;
        for (int dx = 0; dx < a.kw; ++dx) {
          const int xi = x * a.ss + dx - a.ps;
          if (xi < 0 || xi >= a.Wi) continue;
          const size_t src = (((size_t)ti * a.Hi + yi) * a.Wi + xi) * a.Cin;
          const float* wt = a.w + (size_t)((dt * a.kh + dy) * a.kw + dx) * a.Cin * a.Cout + co;
          if (a.in_bf16) {
            const uint16_t* p = (const uint16_t*)a.in + src;
            for (int ci = 0; ci < a.Cin; ++ci) acc = fmaf(bf16_to_f32(p[ci]), wt[(size_t)ci * a.Cout], acc);
          } else {
            const float* p = (const float*)a.in + src;
            for (int ci = 0; ci < a.Cin; ++ci) acc = fmaf(p[ci], wt[(size_t)ci * a.Cout], acc);
          }
        }
      }
    }
    if (a.clamp > 0.f) acc = fminf(fmaxf(acc, -a.clamp), a.clamp);
    if (a.out_f32) a.out_f32[i] = acc;
    if (a.out_bf16) a.out_bf16[i] = f32_to_bf16(acc);
  }
}

}  // namespace

extern "C" int wf_conv3d_cl(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_bf16,
                            int Ti, int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st,
                            int ss, int pt, int ph, int pw, int up2, int tsplit, void* stream) {
  WF_CHECK_ARG(in && w && (out_f32 || out_bf16), "wf_conv3d_cl: null pointer");
  WF_CHECK_ARG(Cin % CBK == 0, "wf_conv3d_cl: Cin (%d) must be a multiple of 32 (use wf_conv3d_small otherwise)", Cin);
  WF_CHECK_ARG(Cout % 4 == 0, "wf_conv3d_cl: Cout (%d) must be a multiple of 4", Cout);
  WF_CHECK_ARG(!tsplit || (Cout % 8 == 0 && !resid), "wf_conv3d_cl: tsplit needs Cout %% 8 == 0 and no residual");
  WF_CHECK_ARG(kt >= 1 && kh >= 1 && kw >= 1 && st >= 1 && ss >= 1, "wf_conv3d_cl: bad kernel / stride");
  const long M = (long)To * Ho * Wo;
  if (M == 0) return WF_OK;
  ConvArgs a;
  a.in = (const uint16_t*)in;
  a.w = (const uint16_t*)w;
  a.bias = bias;
  a.resid = resid;
  a.out_f32 = out_f32;
  a.out_bf16 = (uint16_t*)out_bf16;
  a.Ti = Ti; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin;
  a.To = To; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.kt = kt; a.kh = kh; a.kw = kw;
  a.st = st; a.ss = ss; a.pt = pt; a.ph = ph; a.pw = pw;
  a.up2 = up2; a.tsplit = tsplit; a.silu_out = 0;
  dim3 grid((unsigned)((M + CBM - 1) / CBM), (unsigned)((Cout + CBN - 1) / CBN));
  hipLaunchKernelGGL(k_conv, grid, dim3(CNT), 2 * CBUF, (hipStream_t)stream, a);
  WF_LAUNCH_CHECK("wf_conv3d_cl");
  return WF_OK;
}

extern "C" int wf_conv3d_small(const void* in, int in_dtype, const float* w, const float* bias, float* out_f32, void* out_bf16,
                               int Ti, int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st,
                               int ss, int pt, int ps, float clamp, void* stream) {
  WF_CHECK_ARG(in && w && (out_f32 || out_bf16), "wf_conv3d_small: null pointer");
  SmallConvArgs a;
  a.in = in; a.in_bf16 = in_dtype == WF_BF16; a.w = w; a.bias = bias; a.out_f32 = out_f32; a.out_bf16 = (uint16_t*)out_bf16;
  a.Ti = Ti; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.To = To; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.kt = kt; a.kh = kh; a.kw = kw; a.st = st; a.ss = ss; a.pt = pt; a.ps = ps; a.clamp = clamp;
  size_t n = (size_t)To * Ho * Wo * Cout;
  if (n == 0) return WF_OK;
  hipLaunchKernelGGL(k_conv_small, dim3(grid_for(n, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, a);
  WF_LAUNCH_CHECK("wf_conv3d_small");
  return WF_OK;
}
