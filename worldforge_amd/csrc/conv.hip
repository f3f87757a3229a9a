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
#include <stdlib.h>

#include <algorithm>

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
  float acc_scale;  // out = acc * acc_scale + bias (+ residual): 2^-k when the fp16 weight operand was stored scaled by 2^k (see wf_conv3d_cl_f16); 1 otherwise
  int f16;         // operands (in, w) and the 16-bit output copy are fp16 instead of bf16 (wf_*_f16 entry points)
  // output scatter (osy == 0: none): output pixel (t, y, x) of the [To, Ho, Wo] grid is written to pixel (t, osy * y + ooy, osx * x + oox)
  // of a [To, oH, oW] tensor -- the four phases of a nearest-2x-upsample + 3 x 3 convolution are 2 x 2 convolutions on the source grid
  int osy, ooy, osx, oox, oH, oW;
};
__device__ __forceinline__ size_t out_pixel(const ConvArgs& a, long m) {
  if (a.osy == 0) return (size_t)m;
  const long hw = (long)a.Ho * a.Wo;
  const long t = m / hw, pix = m - t * hw;
  const int y = (int)(pix / a.Wo), x = (int)(pix - (long)y * a.Wo);
  return ((size_t)t * a.oH + (size_t)(a.osy * y + a.ooy)) * a.oW + (size_t)(a.osx * x + a.oox);
}

// Epilogue of the implicit-GEMM kernels (k_conv, k_conv_pp): wave tile 3 x 2 MFMA tiles, lane = pixel l31 of each 32-pixel block, 4
// consecutive output channels per (i, g).  The twelve bias quads of the lane are loaded in ONE batch (tested and loaded per store group
// they were 24 L2 round trips waited one at a time -- on the small-K layers these kernels serve that was a large part of the tile), and
// the optional operands (residual, fp32 / bf16 output, the time-split destination) are tile-uniform, so the store loop is instantiated per
// combination and dispatched once (run-time tests inside the loop are ~4 scalar branches per store group: tools/isa_audit.py).
template <int R, int F, int H, int TS, class OutPixel>
__device__ __forceinline__ void conv_store_wave_tile(const ConvArgs& a, const f32x16 (&acc)[3][2], long m_first, long M, int n0, int l31, int hi,
                                                     OutPixel out_pixel_of) {
  // R / F / H / TS: 0 absent, 1 present, 2 decide at run time
  const bool has_resid = R == 2 ? a.resid != nullptr : R == 1;
  const bool has_f32 = F == 2 ? a.out_f32 != nullptr : F == 1;
  const bool has_bf16 = H == 2 ? a.out_bf16 != nullptr : H == 1;
  const bool tsplit = TS == 2 ? a.tsplit != 0 : TS == 1;
  f32x4 bq[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) bq[i][g] = *reinterpret_cast<const f32x4*>(a.bias + min(n0 + i * 32 + 8 * g + 4 * hi, a.Cout - 4));
  }
  const int chalf = a.Cout >> 1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long m = m_first + j * 32 + l31;
    if (m >= M) continue;
    const long hw = (long)a.Ho * a.Wo;
    const long tt = tsplit ? m / hw : 0, pix = tsplit ? m - tt * hw : 0;
    const size_t opix = tsplit ? 0 : out_pixel_of(m);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = n0 + i * 32 + 8 * g + 4 * hi;
        if (__builtin_expect(co >= a.Cout, 0)) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] * a.acc_scale + bq[i][g][q];  // (a power of two: exact, fused or not)
        size_t o;
        if (tsplit) {
          const int half = co >= chalf ? 1 : 0;
          o = ((size_t)(1 + 2 * tt + half) * hw + pix) * chalf + (co - half * chalf);
        } else {
          o = opix * a.Cout + co;
        }
        if (has_resid) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(a.resid + o);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rr[q];
        }
        if (has_f32) {
          f32x4 ov = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(a.out_f32 + o) = ov;
        }
        if (has_bf16) {
          u32x2 pk = {pack16x2(a.f16, v[0], v[1]), pack16x2(a.f16, v[2], v[3])};
          *reinterpret_cast<u32x2*>(a.out_bf16 + o) = pk;
        }
      }
    }
  }
}
template <class OutPixel>
__device__ __forceinline__ void conv_store_dispatch(const ConvArgs& a, const f32x16 (&acc)[3][2], long m_first, long M, int n0, int l31, int hi,
                                                    OutPixel out_pixel_of) {
  if (a.out_f32 && !a.out_bf16 && !a.tsplit) {
    if (a.resid)
      conv_store_wave_tile<1, 1, 0, 0>(a, acc, m_first, M, n0, l31, hi, out_pixel_of);
    else
      conv_store_wave_tile<0, 1, 0, 0>(a, acc, m_first, M, n0, l31, hi, out_pixel_of);
  } else {
    conv_store_wave_tile<2, 2, 2, 2>(a, acc, m_first, M, n0, l31, hi, out_pixel_of);
  }
}

template <bool F16>
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
      u32x4 fw[3], fx[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) fw[i] = *reinterpret_cast<const u32x4*>(base + fwoff[i] + ((c ^ fwsw[i]) << 4));
#pragma unroll
      for (int j = 0; j < 2; ++j) fx[j] = *reinterpret_cast<const u32x4*>(base + fxoff[j] + ((c ^ fxsw[j]) << 4));
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32t<F16>(fw[i], fx[j], acc[i][j]);
    }
    if (kt_ + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------------
  conv_store_dispatch(a, acc, m0 + wid * 64, M, n0, l31, hi, [&](long m) { return out_pixel(a, m); });
}

// ------------------------------------------------------------------------------------------------------------------
// Stride-1 variant for the FLOP-heavy 3x3x3 residual-block convolutions: 512 pixels x 96 output channels per workgroup,
// 8 waves (wave = 64 pixels x 96 channels), LDS-DMA staging into a double buffer, two-group ping-pong schedule (see gemm.hip
// k_gemm_pp): per 64-channel K super-tile each wave runs R0 | M0 | R1 | M1 (10 LDS fragment reads, 12 register-only MFMAs,
// twice); waves 0-3 and 4-7 run one phase apart so that one wave's MFMA phase always covers the other's LDS phase.
//   * a K super-tile is two independent 32-channel sub-tiles (tap, channel offset) stored side by side in 128-byte LDS rows
//     (chunk c of row r at c ^ ((r >> 1) & 7));
//   * the im2col gather is done by the DMA itself: every lane's source address is its pixel + tap offset, or a 16-byte zero
//     page when the tap falls outside the input (separable validity bits per pixel, prepared once per workgroup);
//   * all LDS-DMA pieces of super-tile t+1 are issued in the gaps of an MFMA phase of super-tile t.
// ------------------------------------------------------------------------------------------------------------------
constexpr int QM = 512, QN = 96, QT = 512;
constexpr int QX_BYTES = QM * 128;  // 64 KiB
constexpr int QW_BYTES = QN * 128;  // 12 KiB
constexpr int QBUF = QX_BYTES + QW_BYTES;
// k_conv_pp: a double buffer of [512 pixels x 64 channels | 96 output channels x 64 channels]; no region is overlaid (direct epilogue stores)
static_assert(2 * QBUF <= 160 * 1024, "k_conv_pp's double buffer must fit the LDS of a CU");
static_assert(QX_BYTES == (QT / 64) * 8192, "8 waves x 8 KiB of pixel rows fill the activation half of a buffer exactly (xb = ... + wid * 8192)");
static_assert(QW_BYTES % 1024 == 0, "weight rows are staged in 1 KiB LDS-DMA pieces");

// select between two global pointers with two 32-bit v_cndmask (a ?: on pointers is lowered to exec-mask branches by hipcc
// when one arm carries address arithmetic)
__device__ __forceinline__ const void* sel_ptr(int ok, const void* p, const void* z) {
  const uint64_t up = (uint64_t)p, uz = (uint64_t)z;
  const uint32_t lo = ok ? (uint32_t)up : (uint32_t)uz;
  const uint32_t hi = ok ? (uint32_t)(up >> 32) : (uint32_t)(uz >> 32);
  return (const void*)(((uint64_t)hi << 32) | lo);
}

struct ConvPPArgs {
  ConvArgs c;
  const uint16_t* zeros;  // >= 16 bytes of zeros in device memory
};

template <bool F16>
__global__ __launch_bounds__(QT, 2) void k_conv_pp(ConvPPArgs pa) {
  const ConvArgs& a = pa.c;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long M = (long)a.To * a.Ho * a.Wo;
  const long m0 = (long)blockIdx.x * QM;
  const int n0 = blockIdx.y * QN;
  const int taps = a.kt * a.kh * a.kw;
  const int kc_per_tap = a.Cin / CBK;
  const int nk = taps * kc_per_tap;       // 32-channel sub-tiles
  const int nk2 = (nk + 1) >> 1;          // 64-channel super-tiles
  const bool groupB = wid >= 4;

  // ---- per-lane DMA geometry ----------------------------------------------------------------------------------------------
  // X: wave w stages its own 64 pixel rows as 8 pieces (8 rows x 128 B).  lane -> (row = 64w + 8i + lane/8, slot = lane%8);
  // source chunk ch = slot ^ ((row >> 1) & 7); sub-tile = ch >> 2; chunk within the sub-tile = ch & 3.
  const int cin8 = a.Cin >> 3;
  int pixchunk[8];   // pixel base offset in 16-byte chunks (pixel index * Cin / 8), valid pixels only
  int vbits[8];      // separable validity: bit dt | bit (3+dy) | bit (6+dx) set when that tap offset stays inside the input
  const int cc = (lane & 3) ^ ((lane >> 4) & 3);
  const int sub_even = (lane & 7) >> 2;  // sub-tile handled by this lane for even pieces; odd pieces: the other one
  {
    // pixel coordinates of the lane's first row by one 32-bit division, the other 7 rows (8 pixels further each) incrementally
    const int Mi = (int)M;
    int m = (int)m0 + wid * 64 + (lane >> 3);
    const int mc = m < Mi ? m : 0;
    const int hw = a.Ho * a.Wo;
    int t = mc / hw;
    int rem = mc - t * hw;
    int y = rem / a.Wo;
    int x = rem - y * a.Wo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int bits = 0;
      if (m < Mi) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (d < a.kt && (unsigned)(t + d - a.pt) < (unsigned)a.Ti) bits |= 1 << d;
          if (d < a.kh && (unsigned)(y + d - a.ph) < (unsigned)a.Hi) bits |= 1 << (3 + d);
          if (d < a.kw && (unsigned)(x + d - a.pw) < (unsigned)a.Wi) bits |= 1 << (6 + d);
        }
      }
      vbits[i] = bits;
      pixchunk[i] = ((t * a.Hi + y) * a.Wi + x) * cin8;
      m += 8;
      x += 8;
      while (x >= a.Wo) {
        x -= a.Wo;
        if (++y == a.Ho) {
          y = 0;
          ++t;
        }
      }
    }
  }
  // W: 12 pieces (96 rows x 128 B): waves 0-3 stage pieces 2w, 2w+1; waves 4-7 stage piece 8 + (w - 4)
  const int nwp = wid < 4 ? 2 : 1;
  const int wp0 = wid < 4 ? 2 * wid : 4 + wid;
  int wchunk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wp0 + i) + (lane >> 3);
    const int co = min(n0 + row, a.Cout - 1);
    wchunk[i] = co * taps * (a.Cin >> 3);
  }
  // scalar description of the two 32-channel sub-tiles of super-tile kt2 (computed ONCE per tile, wave-uniform)
  struct Sub {
    int toff, sel, woff;
    bool ok;
  };
  // sub-tile iterator (wave-uniform scalars): walks (tap, 32-channel chunk) pairs in order without integer divisions
  int it_kidx, it_kc, it_dx, it_dy, it_dt;
  auto it_seek = [&](int kidx) {
    it_kidx = kidx;
    const int kk = kidx < nk ? kidx : 0;
    const int tap = kk / kc_per_tap;
    it_kc = kk - tap * kc_per_tap;
    it_dx = tap % a.kw;
    it_dy = (tap / a.kw) % a.kh;
    it_dt = tap / (a.kw * a.kh);
  };
  auto describe_next = [&](Sub (&d)[2]) {  // describes sub-tiles it_kidx, it_kidx + 1 and advances by two
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      d[s2].ok = it_kidx < nk;
      const int cin0c = it_kc * (CBK >> 3);
      d[s2].toff = (((it_dt - a.pt) * a.Hi + (it_dy - a.ph)) * a.Wi + (it_dx - a.pw)) * cin8 + cin0c;
      d[s2].sel = (1 << it_dt) | (1 << (3 + it_dy)) | (1 << (6 + it_dx));
      d[s2].woff = ((it_dt * a.kh + it_dy) * a.kw + it_dx) * cin8 + cin0c;
      ++it_kidx;
      if (++it_kc == kc_per_tap) {
        it_kc = 0;
        if (++it_dx == a.kw) {
          it_dx = 0;
          if (++it_dy == a.kh) {
            it_dy = 0;
            ++it_dt;
          }
        }
      }
    }
  };
  auto dma_piece = [&](int kt2, const Sub (&d)[2], int i) {  // i in 0..7: X pieces of this wave; 8, 9: its W pieces
    unsigned char* xb = smem + (kt2 & 1) * QBUF + wid * 8192;
    unsigned char* wb = smem + (kt2 & 1) * QBUF + QX_BYTES + wp0 * 1024;
    if (i < 8) {
      const int sb = (i & 1) ? (sub_even ^ 1) : sub_even;
      const int to = sb ? d[1].toff : d[0].toff;
      const int se = sb ? d[1].sel : d[0].sel;
      const int ok = (int)(sb ? d[1].ok : d[0].ok) & (int)((vbits[i] & se) == se);
      glds16(sel_ptr(ok, a.in + ((long)(pixchunk[i] + to + cc) << 3), pa.zeros), xb + i * 1024);
    } else if (i - 8 < nwp) {
      const int sb = ((wp0 + i - 8) & 1) ? (sub_even ^ 1) : sub_even;
      const int ok = (int)(sb ? d[1].ok : d[0].ok);
      glds16(sel_ptr(ok, a.w + ((long)(wchunk[i - 8] + (sb ? d[1].woff : d[0].woff) + cc) << 3), pa.zeros), wb + (i - 8) * 1024);
    }
  };
  auto dma_tile = [&](int kt2) {  // all pieces at once (prologue); the iterator must stand at sub-tile 2*kt2
    Sub d[2];
    describe_next(d);
#pragma unroll
    for (int i = 0; i < 10; ++i) dma_piece(kt2, d, i);
  };

  // ---- fragment addressing --------------------------------------------------------------------------------------------------
  int fxoff[2], fxsw[2], fwoff[3], fwsw[3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wid * 64 + j * 32 + l31;
    fxoff[j] = row * 128;
    fxsw[j] = (row >> 1) & 7;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = i * 32 + l31;
    fwoff[i] = QX_BYTES + row * 128;
    fwsw[i] = (row >> 1) & 7;
  }
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4 fw[2][3], fx[2][2];
  auto read_half = [&](int kt2, int half) {
    const unsigned char* base = smem + (kt2 & 1) * QBUF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 4 * half + 2 * ks + hi;
#pragma unroll
      for (int i = 0; i < 3; ++i) fw[ks][i] = *reinterpret_cast<const u32x4*>(base + fwoff[i] + ((c ^ fwsw[i]) << 4));
#pragma unroll
      for (int j = 0; j < 2; ++j) fx[ks][j] = *reinterpret_cast<const u32x4*>(base + fxoff[j] + ((c ^ fxsw[j]) << 4));
    }
  };
  auto mma_half = [&](int dma_kt2) {
    Sub d[2];
    if (dma_kt2 >= 0) describe_next(d);  // tiles are described in increasing order by each wave
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = mfma32t<F16>(fw[ks][i], fx[ks][j], acc[i][j]);
          const int idx = ks * 6 + i * 2 + j;
          if (dma_kt2 >= 0 && idx < 10) dma_piece(dma_kt2, d, idx);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  it_seek(0);
  dma_tile(0);  // iterator now at tile 1
  drain();
  bar();
  if (!groupB) {
    for (int kt2 = 0; kt2 < nk2; ++kt2) {
      read_half(kt2, 0);
      bar();
      mma_half(kt2 + 1 < nk2 ? kt2 + 1 : -1);
      bar();
      read_half(kt2, 1);
      bar();
      mma_half(-1);
      drain();
      bar();
    }
    bar();
  } else {
    if (nk2 > 1) dma_tile(1);  // iterator now at tile 2
    bar();
    for (int kt2 = 0; kt2 < nk2; ++kt2) {
      read_half(kt2, 0);
      bar();
      mma_half(-1);
      bar();
      read_half(kt2, 1);
      drain();
      bar();
      mma_half(kt2 + 2 < nk2 ? kt2 + 2 : -1);
      bar();
    }
  }

  // ---- epilogue (as k_conv) ---------------------------------------------------------------------------------------------------
  conv_store_dispatch(a, acc, m0 + wid * 64, M, n0, l31, hi, [&](long m) { return out_pixel(a, m); });
}

// ------------------------------------------------------------------------------------------------------------------
// Direct (VALU) convolution for the few layers whose channel counts do not fill an MFMA tile: encoder conv1 (3 -> 96),
// decoder conv1 (16 -> 384), decoder head (96 -> 3), encoder head (384 -> 32), quant convs (1x1x1).  vae.py:288, 392, 421, 316.
// in: channels-last f32 or bf16; w: [taps][Cin][Cout] f32 (Cout fastest, so consecutive lanes read consecutive weights).
// One thread per (pixel, co).
// ------------------------------------------------------------------------------------------------------------------
// Issue schedule of one tap of k_conv_w4's slice loop, shared by the code that emits it and the code that counts its waits.
//   M = 4 NCB MFMAs per tap, MFMA m works on output block m / 4 and pixel block m % 4; the "gap" m is what follows MFMA m.
//   weight fragment of block cb: reloaded in gap 4 cb + 3 (tap t loads tap t + 3), first used by MFMA 4 cb three taps later
//   LDS-DMA pieces (taps 0..7, ND per tap), next tap's 4 pixel fragment reads: the remaining gaps, one memory instruction per gap
template <int NCB, int DMA_ON, int W_ON>
struct W4Sched {
  static constexpr int M = 4 * NCB;
  // which of the tap's two pieces (0 / 1) goes out in gap m, or -1
  static constexpr int dma_at(int m) { return NCB == 3 ? (m == 5 ? 0 : m == 9 ? 1 : -1) : (m == 1 ? 0 : m == 2 ? 1 : -1); }
  // which pixel fragment of the next tap is read in gap m, or -1
  static constexpr int bread_at(int m) { return NCB == 3 ? (m < 3 ? m : m == 4 ? 3 : -1) : (m < 4 ? m : -1); }
  static constexpr int wload_at(int m) { return (m & 3) == 3 ? m / 4 : -1; }  // block whose fragment is reloaded in gap m
  // VMEM operations issued after the load of fragment (tap, cb) -- in gap 4 cb + 3 of tap - 3 -- and before MFMA 4 cb of `tap`.  Taps < 0
  // are the previous slice's 24..26 (no pieces there); the prologue issues W(0..2, .) in the same order.
  static constexpr int younger(int tap, int cb) {
    int n = 0;
    for (int tt = tap - 3; tt <= tap; ++tt)
      for (int g = 0; g < M; ++g) {
        const bool after_issue = tt > tap - 3 || g > 4 * cb + 3;
        const bool before_use = tt < tap || g < 4 * cb;
        if (!after_issue || !before_use) continue;
        if (W_ON && wload_at(g) >= 0) ++n;
        if (DMA_ON && dma_at(g) >= 0 && tt >= 0 && tt < 8) ++n;
      }
    return n;
  }
};

// k_conv_w4: 3x3x3 causal convolution (stride 1) with the input patch resident in LDS, one wave per SIMD.
//
// The implicit-GEMM kernels above gather every (tap, channel-slice) operand tile from L2, so each input pixel crosses the
// L2 -> LDS path 27 times: 76 KiB per 1536 MFMA-cycles per CU, ~50 B/clk/CU -- they sit at 650-710 TFLOP/s on every VAE width,
// bound by that path, not by the matrix pipe.  Here a workgroup computes an 8 x 64 pixel tile of ONE output frame for 96 output
// channels and keeps, per 16-channel slice, the whole (3 frames) x (8+2 rows) x (64+2 columns) input patch in LDS (63 KiB,
// double-buffered): all 27 taps read it at shifted addresses (tap offsets are ds_read immediates; 32 consecutive pixels of a
// fragment are 1 KiB contiguous: no swizzle, no bank conflicts), so a pixel crosses L2 -> LDS ~1.2 times.  Weights never touch
// LDS: the 96 x 16 weight fragment of a tap is 3 global loads per lane from a re-packed copy [27][Cin/16][Cout][16] (each load is
// 1 KiB contiguous per wave; from the [Cout][27][Cin] layout the same loads touch 32 cache lines each and the kernel runs at the
// L2 -> L1 line rate, 3x slower), prefetched 2 taps ahead in a register ring.  4 waves, wave = 2 rows x 64 pixels x 96 channels = 4 x 3 MFMA tiles = 192 accumulator registers in
// AGPRs (inline-asm MFMAs, see attention.hip); per tap 12 MFMAs with 4 LDS reads + 3 global loads + < 1 LDS-DMA piece in their issue
// shadows.  Out-of-range taps (causal / spatial padding, ragged tile edges) come from a zeroed page; one barrier per slice.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WY = 8, WX = 64, W4T = 256;
constexpr int PR = WY + 2, PC = WX + 2;
constexpr int PATCH_PX = 3 * PR * PC;      // 1980 pixels of 32 B
constexpr int PATCH_BUF = 64 * 1024;       // 64 pieces of 1 KiB (1980 * 32 B = 61.9 KiB, the tail of the last piece is padding)
constexpr int W4_LDS = 2 * PATCH_BUF;
// LDS regions of k_conv_w4 and what overlays them:
//   [0, PATCH_BUF)              patch buffer 0: slices 0, 2, 4 ... of a tile
//   [PATCH_BUF, 2 * PATCH_BUF)  patch buffer 1: slices 1, 3, 5 ...; the epilogue's per-wave staging quarters overlay it (store_tile_lds:
//                               dead behind the last slice's barrier, and only the wave's OWN later pieces land in its quarter)
static_assert(PATCH_PX * 32 <= PATCH_BUF, "a 16-channel slice of the (3 frames) x (8 + 2 rows) x (64 + 2 cols) patch must fit one patch buffer");
static_assert(PATCH_BUF == (W4T / 64) * 16 * 1024, "4 waves x 16 LDS-DMA pieces of 1 KiB fill a patch buffer exactly (stage_piece: wid * 16 + j)");
static_assert(W4_LDS <= 160 * 1024, "two patch buffers must fit the LDS of a CU");
static_assert((PATCH_BUF / (W4T / 64)) % 1024 == 0, "a wave's epilogue staging quarter = the LDS range of its own 16 pieces: whole pieces");
#ifndef WF_CONV_DIRECT_STORE
#define WF_CONV_DIRECT_STORE 0  // lab only: 1 = the epilogue of rounds 1-3 (stores straight from the accumulator layout)
#endif

#ifdef WF_CONV_TIMING
__device__ unsigned long long g_conv_cycles[8];
__device__ unsigned long long g_conv_slice[32];
#endif
struct ConvW4Args {
  ConvArgs c;
  const uint16_t* zeros;
  int tiles_x, tiles_y;
  // input addressing (elements): pixel (t, y, x), stored 16-channel slice s -> ((t * Hi + y) * row_stride + x * pix_stride + s * slice_stride)
  //   pixel-major  [T][Hi][Wi][Cs]:        row_stride = Wi * Cs,        pix_stride = Cs, slice_stride = 16
  //   slice-major  [T][Hi][Cs/16][Wi][16]: row_stride = (Cs/16)*Wi*16,  pix_stride = 16, slice_stride = Wi * 16
  //     (a patch row of one slice is Wi * 32 contiguous bytes: every 1 KiB LDS-DMA piece reads 8 whole cache lines instead of 32
  //      32-byte fragments of 32 different pixels)
  // nsa = stored slices Cs / 16; K slice cs reads stored slice cs mod nsa: the fp32-class three-term operand [hi | lo | hi] is stored
  // as [hi | lo] (Cs = 2/3 Cin) and its hi slices are read twice.
  long row_stride, slice_stride;
  int pix_stride, nsa;
  int walk;  // 0 = one contiguous chunk of the tile list per workgroup, 1 = XCD-cooperative (see k_conv_w4)
};

// DBG (ablation builds only, -DWF_CONV_ABLATE + WF_CONV_DEBUG=<bits>; wrong results): bit 0 = no in-loop LDS-DMA, bit 1 = no in-loop weight
// loads, bit 2 = no in-loop LDS fragment reads.  Compile-time, so that the production instantiation (DBG = 0) carries no branches.
// NCB = 32-channel output blocks per workgroup: 3 (96 output channels, the ResidualBlock layers) or 1 (the 96 -> 3 decoder head and the
// 384 -> 32 encoder head, zero-padded to 32 output channels: a third of the MFMAs of the 96-wide tile they used to be computed with).
template <int DBG, int NCB = 3, bool F16 = false>
__global__ __launch_bounds__(W4T, 1) void k_conv_w4(ConvW4Args pa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvArgs& a = pa.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  // persistent: every workgroup walks its own contiguous chunk of the tile list; blockIdx.y -> 96-channel output block.
  // The first patch of the NEXT tile is staged during the last channel slice of the current one, so that only the very first tile
  // of a workgroup pays the cold-start DMA latency, and the epilogue stores run under that flight.
  const int ntile = pa.tiles_x * pa.tiles_y * a.To;
  const int n0 = blockIdx.y * (32 * NCB);
  const int Cin = a.Cin;
  const int ns = Cin / 16;  // even (Cin % 32 == 0): a tile starts in LDS buffer 0 and ends in buffer 1
  int t, y0, x0;
  // tile list in (spatial tile, frame) order, frame fastest, cut into one contiguous chunk per workgroup: consecutive tiles of a
  // workgroup are consecutive frames of the same 8 x 64 window, so two of the three patch frames were fetched by this CU just before
  auto decode = [&](int tile, int& tt, int& yy, int& xx) {
    const int sidx = tile / a.To;
    tt = tile - sidx * a.To;
    xx = (sidx % pa.tiles_x) * WX;
    yy = (sidx / pa.tiles_x) * WY;
  };
  int tile_begin, tile_end, tile_step = 1;
  if (pa.walk == 1 && (gridDim.x & 7) == 0) {
    // XCD-cooperative walk: workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8 -- gridDim.x is a multiple of 8, so the
    // blockIdx.y planes land on the same XCD); an XCD takes one contiguous eighth of the list and its workgroups take CONSECUTIVE
    // items of it (consecutive frames of one window) at the same time, stride = workgroups per XCD: the three patch frames of a tile
    // are wanted by the two neighbours on the same L2 at about the same moment instead of 3 x from HBM / MALL a whole tile apart
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, nl = (int)gridDim.x >> 3;
    const int per = (ntile + 7) / 8;
    tile_begin = xcd * per + j;
    tile_end = min(ntile, (xcd + 1) * per);
    tile_step = nl;
  } else {
    const int chunk = (ntile + (int)gridDim.x - 1) / (int)gridDim.x;
    tile_begin = (int)blockIdx.x * chunk;
    tile_end = min(ntile, tile_begin + chunk);
  }
  if (tile_begin >= tile_end) return;

#ifdef WF_CONV_TIMING
  const unsigned long long tc0 = __builtin_readcyclecounter();
#endif
  // ---- LDS-DMA sources of this wave's 16 patch pieces: lane-load q = (16 w + j) * 64 + lane -> patch pixel q >> 1, chunk q & 1 ----
  const uint16_t* psrc[16];
  auto compute_psrc = [&](int tt0, int yy0, int xx0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = (wid * 16 + j) * 64 + lane;
      const int p = q >> 1, ch = q & 1;
      const int f = p / (PR * PC), rem = p - f * (PR * PC);
      const int r = rem / PC, cc = rem - r * PC;
      const int tt = tt0 - 2 + f, yy = yy0 - a.ph + r, xx = xx0 - 1 + cc;
      const bool ok = p < PATCH_PX && tt >= 0 && yy >= 0 && yy < a.Hi && xx >= 0 && xx < a.Wi;
      psrc[j] = ok ? a.in + ((size_t)tt * a.Hi + yy) * pa.row_stride + (size_t)xx * pa.pix_stride + ch * 8 : pa.zeros;
    }
  };
  decode(tile_begin, t, y0, x0);
  compute_psrc(t, y0, x0);
  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_offset(smem));
  auto dma_piece_c = [&](int cs, auto JC) {  // the same for a compile-time piece number (slice loop): M0 = uniform base + immediate
    (void)&psrc;
    constexpr int j = decltype(JC)::value;
    const int acs = cs >= pa.nsa ? cs - pa.nsa : cs;
    const uint16_t* src = psrc[j] + (size_t)acs * pa.slice_stride;
    glds16_async_m0_imm<j * 1024>(src, smem_base + (uint32_t)((cs & 1) * PATCH_BUF) + (uint32_t)(wid * 16 * 1024));
  };
  auto dma_piece = [&](int cs, int j) {  // piece j of the patch of K slice cs -> buffer cs & 1 (stored slice cs mod nsa)
    const int acs = cs >= pa.nsa ? cs - pa.nsa : cs;
    // ONE wave-uniform slice offset for every lane (a single 64-bit add per piece, no per-lane select): the lanes of out-of-range pixels
    // point at the zero page, which the caller makes at least (nsa - 1) * slice_stride elements + 16 bytes long (checked by wf_conv3d_333)
    const uint16_t* src = psrc[j] + (size_t)acs * pa.slice_stride;
    // issued from inline asm: with the builtin hipcc drains vmcnt(0) in front of every later ds_read (it cannot prove that the DMA's
    // LDS destination does not alias it), i.e. every few MFMAs here.  Ordering is ours: vmcnt + barrier at the end of the slice.
    glds16_async_m0(src, smem_base + (uint32_t)((cs & 1) * PATCH_BUF + (wid * 16 + j) * 1024));
  };

  // ---- fragment addressing ----
  // B (pixels): wave rows 2w, 2w+1 of the tile, two 32-pixel column blocks; patch pixel (dt, row + dy, col + dx), 32 B per pixel.
  // (All 16 lanes of a ds_read_b128 service group read the same channel half, i.e. only the even or the odd 16-byte slots: a 2-way bank
  // conflict on every fragment read -- PMC: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE.  Swapping the two halves of a pixel by bit
  // 3 of its patch column (applied to the DMA source chunk, undone by one offset register per dx) removes them, and measured 3 % SLOWER
  // on every VAE shape: the LDS array is ~20 % busy here, the extra address registers cost more than the conflicts.  Not kept.)
  uint32_t boff[4];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    const int rr = pb >> 1, xb = pb & 1;
    boff[pb] = (uint32_t)(((2 * wid + rr) * PC + xb * 32 + l31) * 32 + hi * 16);
  }
  // A (weights): packed [27][Cin/16][Cout][16]; lane row co = n0 + 32 cb + l31, channels 8 hi .. + 7 of the slice
  const uint32_t aoff = (uint32_t)((l31 * 16 + 8 * hi) * 2);
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.w) + (size_t)n0 * 32;
  const size_t wslice = (size_t)a.Cout * 32;  // bytes per (tap, slice)
  // Weight loads are issued from inline asm (saddr form: wave-uniform 64-bit base + the lane's constant 32-bit offset) and waited for by
  // hand.  The LDS-DMA pieces are inline asm too, i.e. invisible to the compiler's s_waitcnt insertion: with compiler-managed weight
  // loads it emitted vmcnt(N) for N outstanding loads IT knew of, while the hardware counter also held the younger DMA pieces -- every
  // wait then drained loads issued barely one tap earlier and the MFMA stream stalled on L2 latency (43 instead of 32 cycles per MFMA).
  // Round 4: the address of the fragment of (slice, tap, cb) was recomputed per load -- (tap * ns + slice) * wslice as a 64-bit scalar
  // multiply-add, 12 SALU per tap -- in a wave that is alone on its SIMD and has ~12 free issue cycles per MFMA (issue-slot lab, DESIGN
  // section 4c).  The loads of a slice visit (slice, 2), (slice, 3) ... (slice, 26), (next slice, 0), (next slice, 1): a RUNNING uniform
  // pointer `wp` advances by `tapstride` per tap (one 64-bit scalar add) and jumps to the next slice's tap 0 once per slice; the output
  // block cb is an immediate offset of the load.
  const size_t tapstride = (size_t)ns * wslice;  // bytes between two taps of one slice
  const unsigned char* wp = wbase;              // -> (slice 0, tap 0)
  auto wload_cur = [&](u32x4& dst, auto CBC) {
    (void)&aoff, (void)&wp;
    constexpr int cb = decltype(CBC)::value;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(aoff), "s"(wp), "n"(cb * 1024) : "memory");
  };
  // Round 4, schedule of a tap (W4Sched below): the MFMAs of a tap run OUTPUT-BLOCK-major (m -> cb = m / 4, pb = m % 4), so the fragment of
  // block cb is dead after gap 4 cb + 3 -- and is reloaded right there with the fragment of the tap THREE ahead (same ring slot: 27 = 9 x 3),
  // first needed at MFMA 4 cb of that tap: 32 MFMAs ~ 1024 pipe cycles of flight for every fragment where the tap-major order (reload two
  // taps ahead in gaps 4..6, one wait per tap) gave 20 -- with no more registers.  One hand-counted wait per fragment, just before its first
  // MFMA; W4Sched counts the VMEM operations younger than the fragment from the issue order itself.
  using SCH = W4Sched<NCB, (DBG & 1) ? 0 : 1, (DBG & 2) ? 0 : 1>;

  f32x16 acc[4][NCB];
  auto mma = [&](f32x16& c, const u32x4& wv, const u32x4& xv) {
    if constexpr (F16)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(wv), "v"(xv));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(wv), "v"(xv));
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  u32x4 bf[2][4];  // pixel fragments: tap parity (27 taps: tap 26 and the next slice's tap 0 share slot 0, refilled behind the barrier)
  u32x4 af[3][NCB];  // weight fragments: ring over taps (27 = 9 x 3: the slot of a tap is tap % 3 in every slice), each reloaded 3 taps ahead
  auto bread = [&](int buf, int tap, int pb) {
    const int dt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    const unsigned char* base = smem + buf * PATCH_BUF + ((dt * PR + dy) * PC + dx) * 32;
    return *reinterpret_cast<const u32x4*>(base + boff[pb]);
  };
  // the same read in the slice loop: the lane's address INCLUDING the buffer offset sits in a register for the whole slice (bslice) and
  // the tap's patch offset (<= 46 528 B) is the ds_read's immediate -- the per-tap v_add_u32 per fragment is gone
  uint32_t bslice[4];
  // (the destination is the fragment register itself, by reference: a local result copied into it afterwards would be copied BEFORE the
  // data arrives -- the compiler does not know that an inline-asm load completes later)
  auto bread_c = [&](u32x4& dst, auto TC, int pb) {
    (void)&bslice;
    constexpr int tap = decltype(TC)::value;
    constexpr int dt = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(bslice[pb]), "n"(((dt * PR + dy) * PC + dx) * 32) : "memory");
  };

  // prologue: patch of slice 0, weights of taps 0, 1, then the pixel fragments of tap 0
#pragma unroll
  for (int j = 0; j < 16; ++j) dma_piece(0, j);
  for_const<3>([&](auto TP) {
    (void)&af, (void)&wp, (void)&tapstride, (void)&wload_cur;
    for_const<NCB>([&](auto CBC) {
      (void)&af, (void)&wload_cur;
      wload_cur(af[decltype(TP)::value][decltype(CBC)::value], CBC);
    });
    wp += tapstride;
  });  // wp -> (slice 0, tap 3); issue order W(0,0) W(0,1) ... W(2,NCB-1) as in the steady state
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NCB) : "memory");  // the 16 pieces landed (the 3 NCB weight loads were issued after them)
  bar();
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) bf[0][pb] = bread(0, 0, pb);

#ifdef WF_CONV_TIMING
  const unsigned long long tc1 = __builtin_readcyclecounter();
  unsigned long long t_wait = 0, t_main = 0, t_epi = 0, n_tiles = 0;
#endif
  for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
#ifdef WF_CONV_TIMING
  const unsigned long long tt0 = __builtin_readcyclecounter();
#endif
  const int tile_n = tile + tile_step;
  const bool has_next = tile_n < tile_end;
  int tn = t, yn = y0, xn = x0;
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pb][cb][r] = 0.f;
      asm volatile("" : "+a"(acc[pb][cb]));
    }
  for (int cs = 0; cs < ns; ++cs) {
#ifdef WF_CONV_TIMING
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
    const int buf = cs & 1;
    if (cs == ns - 1 && has_next) {  // the sources of the next tile's first patch replace this tile's (no longer needed)
      decode(tile_n, tn, yn, xn);
      compute_psrc(tn, yn, xn);
    }
    // next slice's patch; last slice: slice 0 of the next tile into buffer 0 (or a harmless re-stage when there is none)
    const int csn = cs + 1 < ns ? cs + 1 : (has_next ? 0 : cs);
    // per-slice uniforms / registers of the round-4 address paths: where the weight pointer jumps to after tap 26's fragments (the next
    // slice's tap 0; past the last slice: slice 0 of the next tile, same weights), and the fragment read addresses with the buffer folded in
    const unsigned char* w_next = wbase + (size_t)(cs + 1 < ns ? cs + 1 : 0) * wslice;
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      bslice[pb] = lds_offset(smem) + (uint32_t)(buf * PATCH_BUF) + boff[pb];
      asm volatile("" : "+v"(bslice[pb]));  // one register each for the whole slice (not re-materialised per tap)
    }
    for_const<27>([&](auto TC) {
      constexpr int tap = decltype(TC)::value;
      // this tap's four pixel fragments were read (inline asm: the compiler does not count them) in the first gaps of the previous tap, >= 7
      // MFMAs ago; tap 0's come from the compiler-managed reads behind the slice barrier
      if constexpr (tap > 0 && !(DBG & 4))
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[tap & 1][0]), "+v"(bf[tap & 1][1]), "+v"(bf[tap & 1][2]), "+v"(bf[tap & 1][3]));
      // gap m of the tap: MFMA (output block m / 4, pixel block m % 4); in its shadow, per W4Sched: the next tap's 4 pixel fragments, the
      // weight fragment of the block just finished for the tap three ahead (VMEM), the 2 LDS-DMA pieces of the next slice's patch (VMEM,
      // taps 0..7) -- the VMEM issue ORDER is what the hand-counted waits assume
      constexpr int M = SCH::M;
      for_const<M>([&](auto MC) {
        (void)&acc, (void)&af, (void)&bf, (void)&wp, (void)&w_next, (void)&tapstride, (void)&bslice, (void)&psrc, (void)&wload_cur, (void)&bread_c,
            (void)&dma_piece_c;
        constexpr int m = decltype(MC)::value;
        constexpr int cb = m / 4, pb = m % 4;
        if constexpr (pb == 0 && !(DBG & 2))  // first MFMA on this block's fragment: it was issued 3 taps ago, SCH counts what is younger
          asm volatile("s_waitcnt vmcnt(%1)" : "+v"(af[tap % 3][cb]) : "n"(SCH::younger(tap, cb)));
        mma(acc[pb][cb], af[tap % 3][cb], bf[tap & 1][pb]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SCH::bread_at(m) >= 0 && tap < 26 && !(DBG & 4))  // next tap's pixel fragments
          bread_c(bf[(tap + 1) & 1][SCH::bread_at(m)], std::integral_constant<int, tap + 1>{}, SCH::bread_at(m));
        if constexpr (pb == 3 && !(DBG & 2)) {  // block cb is done for this tap: its slot takes the fragment of the tap three ahead
          wload_cur(af[tap % 3][cb], std::integral_constant<int, cb>{});
          if constexpr (cb == NCB - 1) {
            if constexpr (tap == 23)
              wp = w_next;       // tap 26's fragments were the slice's last: on to (next slice, tap 0)
            else
              wp += tapstride;
          }
        }
        if constexpr (SCH::dma_at(m) >= 0 && tap < 8 && !(DBG & 1)) dma_piece_c(csn, std::integral_constant<int, 2 * tap + SCH::dma_at(m)>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
#ifdef WF_CONV_TIMING
    const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NCB) : "memory");  // patch of slice cs+1 landed (only the 3 NCB prefetched weight loads are younger)
    bar();
#ifdef WF_CONV_TIMING
    t_wait += __builtin_readcyclecounter() - tw0;
    if (lane == 0 && wid == 0 && cs < 32) atomicAdd(&g_conv_slice[cs], __builtin_readcyclecounter() - ts0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the atomic is a VMEM op the hand-counted vmcnt waits of the next slice do not know
#endif
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) bf[0][pb] = bread(buf ^ 1, 0, pb);
  }

#ifdef WF_CONV_TIMING
  const unsigned long long tt1 = __builtin_readcyclecounter();
#endif
  // ---- epilogue ----
  // The optional operands (bias, residual, the two output forms) are tile-uniform: the store loop is instantiated for the combinations
  // the VAE uses and dispatched ONCE per tile.  Tested inside the loop they were ~4 scalar branches per store group, two of them taken in
  // the common case -- and a taken branch costs a wave that is alone on its SIMD ~130 cycles of instruction fetch: 12 of the epilogue's
  // 16.7 k cycles (tools/conv_timing.py).
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
  auto store_tile = [&](auto RC, auto FC, auto HC, auto BC) {
    // RC / FC / HC / BC: std::integral_constant<int, v>, v = 0 absent, 1 present, 2 decide at run time (generic fallback)
    constexpr int R = decltype(RC)::value, F = decltype(FC)::value, H = decltype(HC)::value, BI = decltype(BC)::value;
    const bool has_resid = R == 2 ? a.resid != nullptr : R == 1;
    const bool has_f32 = F == 2 ? a.out_f32 != nullptr : F == 1;
    const bool has_bf16 = H == 2 ? a.out_bf16 != nullptr : H == 1;
    const bool has_bias = BI == 2 ? a.bias != nullptr : BI == 1;
    int lane_ = lane;  // opaque per tile (see store_tile_lds): no per-lane address of the epilogue lives across the slice loop
    asm volatile("" : "+v"(lane_));
    const int hi = lane_ >> 5, l31 = lane_ & 31;
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      const int y = y0 + 2 * wid + (pb >> 1), x = x0 + (pb & 1) * 32 + l31;
      const bool inb = y < a.Ho && x < a.Wo;
      const size_t m = ((size_t)t * a.Ho + y) * a.Wo + x;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        f32x16 av = acc[pb][cb];
        asm volatile("" : "+v"(av));
        if (!inb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = n0 + cb * 32 + 8 * g + 4 * hi;
          if (__builtin_expect(co >= a.Cout, 0)) continue;
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = F16 ? av[4 * g + q] * a.acc_scale : av[4 * g + q];
          if (has_bias) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += bb[q];
          }
          const size_t o = m * a.Cout + co;
          if (has_resid) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(a.resid + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += rr[q];
          }
          if (has_f32) {
            f32x4 ov = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(a.out_f32 + o) = ov;
          }
          if (has_bf16) {
            u32x2 pk = {pack16x2(a.f16, v[0], v[1]), pack16x2(a.f16, v[2], v[3])};
            *reinterpret_cast<u32x2*>(a.out_bf16 + o) = pk;
          }
        }
      }
    }
  };
  // Round 4: the common case (bias, fp32 out, optional residual) of the 96-channel block goes THROUGH LDS.  In the accumulator layout a lane
  // owns one pixel and quads of channels: a store instruction's 64 lanes hit 64 different lines (lane stride = Cout * 4 B), the CU's address
  // pipe takes them one line per cycle, and 4 waves x 48 stores x 64 = 12.3 k of the epilogue's 14.8-15.9 k cycles (7.5 % of a 96-wide x3 tile,
  // profiles/r4_q_conv_cycles_ablation.txt).  Each wave stages one pixel block at a time (32 pixels x 96 channels, rows padded to 400 B: the
  // transposing ds_write_b128 are conflict-free) in ITS OWN 16 KiB quarter of patch buffer 1 -- dead behind the last slice's barrier, and the
  // only LDS-DMA writes that will land there are this wave's own pieces of the next tile's slice 1, later in its program order: no
  // workgroup barrier -- and reads it back pixel-row-major: 16 B per lane, 24 lanes per pixel, 384 B runs (the whole 12 KiB when Cout = 96).
  // Same arithmetic in the same order (acc + bias, then + residual): bit-identical.
  auto store_tile_lds = [&](auto RC) {  // FULL tiles only (every pixel and channel of the block in range): no predicate anywhere
    constexpr int R = decltype(RC)::value;
    constexpr int ROW = 400;
    static_assert(32 * ROW <= PATCH_BUF / (W4T / 64), "a pixel block must fit the wave's own quarter of the patch buffer");
    unsigned char* stg = smem + PATCH_BUF + wid * (PATCH_BUF / (W4T / 64));
    int lane_ = lane;  // opaque per tile: the per-chunk pixel / channel / offset values below are a few VALU each -- hoisted out of the tile loop
    asm volatile("" : "+v"(lane_));  // as loop invariants they were 36 registers live across the slice loop (78 spills)
    const int hi_ = lane_ >> 5, l31_ = lane_ & 31;
    // row phase geometry: chunk c = 64 i + lane of a pixel block = pixel c / 24, channels 4 (c % 24) .. + 3.  64 = 16 (mod 24): the channel
    // quad of step i depends on i % 3 only -- THREE bias quads per lane serve the whole tile (loaded once, one wait)
    int pc[3], kc[3];
    f32x4 bq[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = j * 64 + lane_;
      pc[j] = c / 24;
      kc[j] = c - 24 * pc[j];
      bq[j] = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * kc[j]);
    }
    // step i = 3 q + j: c = 192 q + (64 j + lane) -> pixel 8 q + pc[j], channel quad kc[j]
    auto gofs = [&](int pb, int i) -> size_t {
      const int y = y0 + 2 * wid + (pb >> 1), x = x0 + (pb & 1) * 32 + 8 * (i / 3) + pc[i % 3];
      return (((size_t)t * a.Ho + y) * a.Wo + x) * a.Cout + n0 + 4 * kc[i % 3];
    };
    f32x4 rr[2][4];
    auto resid_group = [&](int slot, int pb, int g4) {  // the 4 residual quads of steps 4 g4 .. 4 g4 + 3 of pixel block pb
#pragma unroll
      for (int u = 0; u < 4; ++u) rr[slot][u] = *reinterpret_cast<const f32x4*>(a.resid + gofs(pb, 4 * g4 + u));
    };
    if constexpr (R) resid_group(0, 0, 0);
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        f32x16 av = acc[pb][cb];
        asm volatile("" : "+v"(av));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {av[4 * g], av[4 * g + 1], av[4 * g + 2], av[4 * g + 3]};
          *reinterpret_cast<f32x4*>(stg + l31_ * ROW + (cb * 32 + 8 * g + 4 * hi_) * 4) = v;
        }
        __builtin_amdgcn_sched_barrier(0);  // one accumulator block (16 registers) at a time
      }
#pragma unroll
      for (int g4 = 0; g4 < 3; ++g4) {  // four chunks per lane at a time; the NEXT group's residual quads are in flight meanwhile
        const int slot = (pb * 3 + g4) & 1;
        if constexpr (R) {
          if (g4 < 2)
            resid_group(slot ^ 1, pb, g4 + 1);
          else if (pb < 3)
            resid_group(slot ^ 1, pb + 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 lv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = 4 * g4 + u;
          lv[u] = *reinterpret_cast<const f32x4*>(stg + (8 * (i / 3) + pc[i % 3]) * ROW + kc[i % 3] * 16);
        }
        __builtin_amdgcn_sched_barrier(0);  // the four LDS reads as one batch (one wait), not read - wait - store four times
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = 4 * g4 + u;
          f32x4 v = lv[u];
          const f32x4 bb = bq[i % 3];
          if constexpr (F16) {  // fp16 weight operands are stored scaled by a power of two (exact): one fma instead of the add
            const float sc = a.acc_scale;
            v = f32x4{v[0] * sc + bb[0], v[1] * sc + bb[1], v[2] * sc + bb[2], v[3] * sc + bb[3]};
          } else {
            v = f32x4{v[0] + bb[0], v[1] + bb[1], v[2] + bb[2], v[3] + bb[3]};
          }
          if constexpr (R) {
            const f32x4 r4 = rr[slot][u];
            v = f32x4{v[0] + r4[0], v[1] + r4[1], v[2] + r4[2], v[3] + r4[3]};
          }
          *reinterpret_cast<f32x4*>(a.out_f32 + gofs(pb, i)) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    const bool common = a.bias && a.out_f32 && !a.out_bf16;  // every ResidualBlock conv of the VAE: bias, fp32 activations out
    const bool full = y0 + WY <= a.Ho && x0 + WX <= a.Wo && n0 + 32 * NCB <= a.Cout;  // ragged edge tiles keep the direct stores
    if (NCB == 3 && !WF_CONV_DIRECT_STORE && common && full) {
      if (!a.resid)
        store_tile_lds(I0{});
      else
        store_tile_lds(I1{});
    } else {
      if (common && !a.resid)
        store_tile(I0{}, I1{}, I0{}, I1{});
      else if (common)
        store_tile(I1{}, I1{}, I0{}, I1{});
      else
        store_tile(I2{}, I2{}, I2{}, I2{});
    }
  }
#ifdef WF_CONV_TIMING
  t_main += tt1 - tt0;
  t_epi += __builtin_readcyclecounter() - tt1;
  ++n_tiles;
#endif
  t = tn;
  y0 = yn;
  x0 = xn;
  }  // tile loop
#ifdef WF_CONV_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && wid == 0) {
    atomicAdd(&g_conv_cycles[0], tc1 - tc0);
    atomicAdd(&g_conv_cycles[1], t_main);
    atomicAdd(&g_conv_cycles[2], t_epi);
    atomicAdd(&g_conv_cycles[3], t_wait);
    atomicAdd(&g_conv_cycles[4], (unsigned long long)ns * n_tiles);
    atomicAdd(&g_conv_cycles[6], n_tiles);
    atomicAdd(&g_conv_cycles[5], 1ull);
  }
#endif
}

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

static int conv3d_cl_impl(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_bf16,
                          int Ti, int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st,
                          int ss, int pt, int ph, int pw, int up2, int tsplit, const void* zero_page, const int* scatter, void* stream,
                          int f16 = 0, float acc_scale = 1.0f) {
  WF_CHECK_ARG(acc_scale > 0.0f && acc_scale < 3.0e38f, "wf_conv3d_cl: acc_scale must be a positive finite number (1 = none)");
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
  a.up2 = up2; a.tsplit = tsplit; a.acc_scale = acc_scale;
  a.f16 = f16;
  a.osy = a.ooy = a.osx = a.oox = a.oH = a.oW = 0;
  if (scatter) {  // {out_H, out_W, sy, oy, sx, ox}
    WF_CHECK_ARG(!tsplit && scatter[2] >= 1 && scatter[4] >= 1 && scatter[3] >= 0 && scatter[5] >= 0 &&
                     scatter[2] * (Ho - 1) + scatter[3] < scatter[0] && scatter[4] * (Wo - 1) + scatter[5] < scatter[1],
                 "wf_conv3d_cl_scatter: the scattered grid does not fit the output tensor");
    a.oH = scatter[0]; a.oW = scatter[1]; a.osy = scatter[2]; a.ooy = scatter[3]; a.osx = scatter[4]; a.oox = scatter[5];
  }
  // stride-1, no upsample / frame interleave, big enough to fill the chip: the 512-pixel ping-pong kernel (needs a zero page
  // for the out-of-range taps of its DMA gather and chunk-granular 32-bit offsets)
  const long in_chunks = (long)Ti * Hi * Wi * (Cin / 8);
  if (zero_page && st == 1 && ss == 1 && !up2 && !tsplit && kt <= 3 && kh <= 3 && kw <= 3 && M >= 64 * QM &&
      in_chunks < (1L << 31) - (1L << 24) && (long)Cout * kt * kh * kw * (Cin / 8) < (1L << 31)) {
    ConvPPArgs pa;
    pa.c = a;
    pa.zeros = (const uint16_t*)zero_page;
    dim3 grid((unsigned)((M + QM - 1) / QM), (unsigned)((Cout + QN - 1) / QN));
    if (f16)
      hipLaunchKernelGGL(k_conv_pp<true>, grid, dim3(QT), 2 * QBUF, (hipStream_t)stream, pa);
    else
      hipLaunchKernelGGL(k_conv_pp<false>, grid, dim3(QT), 2 * QBUF, (hipStream_t)stream, pa);
    WF_LAUNCH_CHECK("wf_conv3d_cl");
    return WF_OK;
  }
  dim3 grid((unsigned)((M + CBM - 1) / CBM), (unsigned)((Cout + CBN - 1) / CBN));
  if (f16)
    hipLaunchKernelGGL(k_conv<true>, grid, dim3(CNT), 2 * CBUF, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_conv<false>, grid, dim3(CNT), 2 * CBUF, (hipStream_t)stream, a);
  WF_LAUNCH_CHECK("wf_conv3d_cl");
  return WF_OK;
}

extern "C" int wf_conv3d_cl(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_bf16,
                            int Ti, int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st,
                            int ss, int pt, int ph, int pw, int up2, int tsplit, const void* zero_page, void* stream) {
  return conv3d_cl_impl(in, w, bias, resid, out_f32, out_bf16, Ti, Hi, Wi, Cin, To, Ho, Wo, Cout, kt, kh, kw, st, ss, pt, ph, pw, up2, tsplit,
                        zero_page, nullptr, stream);
}

extern "C" int wf_conv3d_cl_scatter(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_bf16,
                                    int Ti, int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st,
                                    int ss, int pt, int ph, int pw, const void* zero_page, int out_H, int out_W, int sy, int oy, int sx,
                                    int ox, void* stream) {
  const int scatter[6] = {out_H, out_W, sy, oy, sx, ox};
  return conv3d_cl_impl(in, w, bias, resid, out_f32, out_bf16, Ti, Hi, Wi, Cin, To, Ho, Wo, Cout, kt, kh, kw, st, ss, pt, ph, pw, 0, 0,
                        zero_page, scatter, stream);
}

// The same convolutions on fp16 operands (in, w and the optional 16-bit output copy are fp16; v_mfma_f32_32x32x16_f16): the VAE's
// "fp16x3" operand format -- hi = fp16(x), lo = fp16(x - hi), 2^-22 per product where the bf16 split gives 2^-16 -- and its one-term "fp16" mode.
extern "C" int wf_conv3d_cl_f16(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_f16,
                                int Ti, int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st,
                                int ss, int pt, int ph, int pw, int up2, int tsplit, const void* zero_page, float acc_scale, void* stream) {
  return conv3d_cl_impl(in, w, bias, resid, out_f32, out_f16, Ti, Hi, Wi, Cin, To, Ho, Wo, Cout, kt, kh, kw, st, ss, pt, ph, pw, up2, tsplit,
                        zero_page, nullptr, stream, 1, acc_scale);
}

extern "C" int wf_conv3d_cl_scatter_f16(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_f16,
                                        int Ti, int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st,
                                        int ss, int pt, int ph, int pw, const void* zero_page, int out_H, int out_W, int sy, int oy, int sx,
                                        int ox, float acc_scale, void* stream) {
  const int scatter[6] = {out_H, out_W, sy, oy, sx, ox};
  return conv3d_cl_impl(in, w, bias, resid, out_f32, out_f16, Ti, Hi, Wi, Cin, To, Ho, Wo, Cout, kt, kh, kw, st, ss, pt, ph, pw, 0, 0,
                        zero_page, scatter, stream, 1, acc_scale);
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

// weight re-pack for k_conv_w4: [Cout][27][Cin] -> [27][Cin/16][Cout][16]
namespace {
__global__ void k_pack333(const uint16_t* __restrict__ w, uint16_t* __restrict__ o, int Cout, int Cin) {
  const size_t n = (size_t)Cout * 27 * Cin;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int tap = (int)((i / Cin) % 27);
    const int co = (int)(i / ((size_t)Cin * 27));
    o[(((size_t)tap * (Cin / 16) + ci / 16) * Cout + co) * 16 + (ci & 15)] = w[i];
  }
}
}  // namespace

extern "C" int wf_conv3d_pack333(const void* w, void* w_packed, int Cout, int Cin, void* stream) {
  WF_CHECK_ARG(w && w_packed, "wf_conv3d_pack333: null pointer");
  WF_CHECK_ARG(Cin % 16 == 0 && Cout > 0, "wf_conv3d_pack333: Cin (%d) must be a multiple of 16", Cin);
  hipLaunchKernelGGL(k_pack333, dim3(512), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)w, (uint16_t*)w_packed, Cout, Cin);
  WF_LAUNCH_CHECK("wf_conv3d_pack333");
  return WF_OK;
}

extern "C" size_t wf_conv3d_333_zero_page_bytes(int Wi, int Cin_stored, int layout) {
  // the lanes of padding pixels read the zero page at the same wave-uniform slice offset as the lanes of real pixels read the input
  const size_t slice_stride = layout == 1 ? (size_t)Wi * 16 : (size_t)16;
  const size_t nsa = Cin_stored > 0 ? (size_t)Cin_stored / 16 : 1;
  return (nsa - 1) * slice_stride * 2 + 64;
}

static int conv3d_333_impl(const void* in, const void* w_packed, const float* bias, const float* resid, float* out_f32,
                           void* out_bf16, int T, int Hi, int Wi, int Cin, int Ho, int Cout, int ph, const void* zero_page,
                           size_t zero_page_bytes, int layout, int Cin_stored, void* stream, int f16, float acc_scale = 1.0f) {
  WF_CHECK_ARG(in && w_packed && zero_page && (out_f32 || out_bf16), "wf_conv3d_333: null pointer");
  WF_CHECK_ARG(acc_scale > 0.0f && acc_scale < 3.0e38f, "wf_conv3d_333: acc_scale must be a positive finite number (1 = none)");
  {
    const size_t need = wf_conv3d_333_zero_page_bytes(Wi, Cin_stored, layout);
    WF_CHECK_ARG(zero_page_bytes >= need, "wf_conv3d_333: zero page of %zu bytes, this shape needs %zu (wf_conv3d_333_zero_page_bytes)",
                 zero_page_bytes, need);
  }
  WF_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0, "wf_conv3d_333: Cin (%d) and Cout (%d) must be multiples of 32", Cin, Cout);
  WF_CHECK_ARG(layout == 0 || layout == 1, "wf_conv3d_333: layout must be 0 (pixel-major) or 1 (slice-major)");
  WF_CHECK_ARG(Cin_stored == Cin || (Cin % 3 == 0 && Cin_stored == Cin / 3 * 2 && Cin_stored % 16 == 0),
               "wf_conv3d_333: Cin_stored (%d) must be Cin (%d) or, for the three-term operand stored as [hi | lo], 2/3 of it", Cin_stored, Cin);
  WF_CHECK_ARG((long)27 * Cin * Cout * 2 < (1L << 31), "wf_conv3d_333: weight tensor too large");
  if ((long)T * Ho * Wi == 0) return WF_OK;
  ConvW4Args wa;
  ConvArgs& a = wa.c;
  a.in = (const uint16_t*)in;
  a.w = (const uint16_t*)w_packed;
  a.bias = bias;
  a.resid = resid;
  a.out_f32 = out_f32;
  a.out_bf16 = (uint16_t*)out_bf16;
  a.Ti = T; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin;
  a.To = T; a.Ho = Ho; a.Wo = Wi; a.Cout = Cout;
  a.kt = a.kh = a.kw = 3;
  a.st = a.ss = 1; a.pt = 2; a.ph = ph; a.pw = 1;
  a.up2 = a.tsplit = 0;
  a.acc_scale = acc_scale;
  a.f16 = f16;
  a.osy = a.ooy = a.osx = a.oox = a.oH = a.oW = 0;
  wa.zeros = (const uint16_t*)zero_page;
  wa.tiles_x = (Wi + WX - 1) / WX;
  wa.tiles_y = (Ho + WY - 1) / WY;
  wa.nsa = Cin_stored / 16;
  wa.walk = 1;  // XCD-cooperative tile walk (round 2: +14 % on the 96-wide layer over one contiguous chunk per workgroup, walk = 0)

  if (layout == 0) {
    wa.row_stride = (long)Wi * Cin_stored; wa.pix_stride = Cin_stored; wa.slice_stride = 16;
  } else {
    wa.row_stride = (long)wa.nsa * Wi * 16; wa.pix_stride = 16; wa.slice_stride = (long)Wi * 16;
  }
  const bool thin = Cout <= 32;  // one 32-channel output block per workgroup (the VAE heads)
  const int ny = thin ? 1 : (Cout + 95) / 96, ntile = wa.tiles_x * wa.tiles_y * T;
  const int gx = std::min(ntile, std::max(1, 256 / ny));  // one persistent workgroup per CU
  dim3 grid((unsigned)gx, (unsigned)ny);
#ifdef WF_CONV_ABLATE
  {
    const char* dbg = getenv("WF_CONV_DEBUG");
    const int d = dbg ? atoi(dbg) : 0;
    if (d == 1) hipLaunchKernelGGL(k_conv_w4<1>, grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
    else if (d == 2) hipLaunchKernelGGL(k_conv_w4<2>, grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
    else if (d == 3) hipLaunchKernelGGL(k_conv_w4<3>, grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
    else if (d == 7) hipLaunchKernelGGL(k_conv_w4<7>, grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
    else hipLaunchKernelGGL(k_conv_w4<0>, grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
  }
#else
  if (thin && f16)
    hipLaunchKernelGGL((k_conv_w4<0, 1, true>), grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
  else if (thin)
    hipLaunchKernelGGL((k_conv_w4<0, 1>), grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
  else if (f16)
    hipLaunchKernelGGL((k_conv_w4<0, 3, true>), grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
  else
    hipLaunchKernelGGL((k_conv_w4<0, 3>), grid, dim3(W4T), W4_LDS, (hipStream_t)stream, wa);
#endif
  WF_LAUNCH_CHECK("wf_conv3d_333");
  return WF_OK;
}

extern "C" int wf_conv3d_333(const void* in, const void* w_packed, const float* bias, const float* resid, float* out_f32,
                             void* out_bf16, int T, int Hi, int Wi, int Cin, int Ho, int Cout, int ph, const void* zero_page,
                             size_t zero_page_bytes, int layout, int Cin_stored, void* stream) {
  return conv3d_333_impl(in, w_packed, bias, resid, out_f32, out_bf16, T, Hi, Wi, Cin, Ho, Cout, ph, zero_page, zero_page_bytes, layout,
                         Cin_stored, stream, 0);
}

extern "C" int wf_conv3d_333_f16(const void* in, const void* w_packed, const float* bias, const float* resid, float* out_f32,
                                 void* out_f16, int T, int Hi, int Wi, int Cin, int Ho, int Cout, int ph, const void* zero_page,
                                 size_t zero_page_bytes, int layout, int Cin_stored, float acc_scale, void* stream) {
  return conv3d_333_impl(in, w_packed, bias, resid, out_f32, out_f16, T, Hi, Wi, Cin, Ho, Cout, ph, zero_page, zero_page_bytes, layout,
                         Cin_stored, stream, 1, acc_scale);
}

#ifdef WF_CONV_TIMING
extern "C" int wf_debug_conv_slices(unsigned long long* out32) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_conv_slice), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  unsigned long long z[32] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_slice), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
extern "C" int wf_debug_conv_cycles(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_conv_cycles), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_cycles), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
