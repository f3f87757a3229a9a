// Dense optical flow of the FLF gate on the GPU: the job cv2.calcOpticalFlowFarneback does in the reference
// (scheduling_unipc_multistep_clean.py:156-248, called per latent channel from :382-389 and :466-476 with the fixed parameters of
// :220-224: pyr_scale .5, levels 3, winsize 15, iterations 3, poly_n 5, poly_sigma 1.2, flags 0).  The reference moves every
// channel to the host, quantises it to uint8 and loops over frame pairs in OpenCV (32 D2H copies and 640 CPU flow calls per gate
// at the 81-frame config); here all C x (T-1) frame pairs of a latent tensor are processed in one batch of small launches and
// nothing leaves the device.
//
// Algorithm (G. Farneback, SCIA 2003, as published in OpenCV's optflowgf.cpp): per pyramid level, Gaussian pre-blur + bilinear
// resize of both frames, quadratic polynomial expansion (separable weighted least squares with a 11-tap Gaussian applicability),
// then 3 rounds of [per-pixel 2x2 normal equations from the expansion coefficients at the displaced position -> 15x15 box average
// -> solve].  Numerics follow the C++ float / double split (float images and matrices, double accumulators for the horizontal
// expansion pass, the box sums and the solve); this file is compiled with -ffp-contract=off so every product / sum rounds
// separately as in the oracle (oracle/farneback.py).  PARITY IS UNPINNED against a real cv2 (none available offline).
//
// HBM-bound, tiny: at 81 f x 480p a latent tensor is 336 images of 60 x 104 pixels (one pyramid level); every kernel is one thread
// per pixel with coalesced planar / interleaved rows, the whole working set (< 100 MB) lives in L2 / MALL.
#include "common.h"

#include <algorithm>
#include <cmath>

namespace {

using namespace wf;

constexpr int POLY_N = 5;
constexpr int WIN = 15;
constexpr int WIN_M = WIN / 2;
constexpr int ITERS = 3;
constexpr int MAX_LEVELS = 3;
constexpr int MIN_SIZE = 32;
constexpr int MAX_KSIZE = 19;  // pre-blur kernel at the deepest level (scale 1/8: sigma 3.5 -> cvRound(17.5) | 1 = 19)
constexpr int NT = 256;

struct Blur {
  int ksize;
  float k[MAX_KSIZE];
};
struct Poly {
  float g[POLY_N + 1], xg[POLY_N + 1], xxg[POLY_N + 1];  // taps 0..n (g symmetric, xg antisymmetric)
  double ig11, ig03, ig33, ig55;
};

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i = (i < 0 ? -i : i) % p;
  return i >= n ? p - i : i;
}
__device__ __forceinline__ int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

// ---- global min / range of the tensor (SCHED:376-378, 462-464): exact whatever the reduction order --------------------------
// blockIdx.y = group: the whole tensor (Wan: one global range) or one channel (LongCat: a range per channel); n elements per group
__global__ void k_minmax_partial(TView x, size_t n, float* part) {
  float mn = INFINITY, mx = -INFINITY;
  const size_t g0 = (size_t)blockIdx.y * n;
  part += (size_t)blockIdx.y * 2 * gridDim.x;
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const float v = tload(x, g0 + i);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  __shared__ float smn[NT / 64], smx[NT / 64];
  mn = wave_min(mn);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) smn[threadIdx.x >> 6] = mn, smx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < NT / 64; ++i) mn = fminf(mn, smn[i]), mx = fmaxf(mx, smx[i]);
    part[2 * blockIdx.x] = mn;
    part[2 * blockIdx.x + 1] = mx;
  }
}
__global__ void k_minmax_final(const float* part, int nb, float* mm, int bf16_math) {
  float mn = INFINITY, mx = -INFINITY;
  part += (size_t)blockIdx.x * 2 * nb;
  mm += 2 * blockIdx.x;
  for (int i = threadIdx.x; i < nb; i += 64) mn = fminf(mn, part[2 * i]), mx = fmaxf(mx, part[2 * i + 1]);
  mn = wave_min(mn);
  mx = wave_max(mx);
  if (threadIdx.x == 0) {
    mm[0] = mn;
    // range + 1e-8 in the tensor's arithmetic: fp32 (Wan: the channel was promoted first), or bf16 ops on a bf16 tensor (LongCat)
    mm[1] = bf16_math ? rbf(rbf(mx - mn) + 1e-8f) : (mx - mn) + 1e-8f;
  }
}

// ---- quantise to the uint8 grey level the reference hands to OpenCV, fused with the horizontal pass of the pre-blur -----------
// mode 0 (Wan, SCHED:165-201): one global range, uint8(n * 255).  mode 1 (LongCat, LSCHED:105-121, 290-297): a range per channel,
// normalised in the tensor's own dtype (bf16 ops round after every step), then -- the normalised value lies in [0, 1], inside the
// reference's "[-1.1, 1.1]" test -- uint8(clip((n + 1) * 127.5, 0, 255)) in fp32.
template <int MODE>
__device__ __forceinline__ float grey(const TView& x, size_t i, float mn, float rg) {
  if constexpr (MODE == 0) {
    const float nrm = (tload(x, i) - mn) / rg;       // (channel_rgb - min) / range            SCHED:388 / 474
    const float s = nrm * 255.0f;                    // video_np * 255                         SCHED:175
    return (float)(unsigned char)(int)s;             // .astype(np.uint8): truncation; RGB2GRAY of 3 equal channels is the identity
  } else {
    float nrm;
    if (x.dt == WF_BF16)
      nrm = rbf(rbf(tload(x, i) - mn) / rg);
    else
      nrm = (tload(x, i) - mn) / rg;
    const float s = fminf(fmaxf((nrm + 1.0f) * 127.5f, 0.0f), 255.0f);
    return (float)(unsigned char)(int)s;
  }
}
template <int MODE>
__global__ void k_quant_blur_rows(TView x, const float* mm, float* out, int N, int h, int w, Blur b, int frames_per_group) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= (size_t)N * h * w) return;
  const int xx = (int)(i % w);
  const size_t row = i - xx;
  const size_t grp = MODE == 0 ? 0 : i / ((size_t)frames_per_group * h * w);
  const float mn = mm[2 * grp], rg = mm[2 * grp + 1];
  const int r = b.ksize / 2;
  float acc = 0.f;
  for (int j = 0; j < b.ksize; ++j) acc = acc + b.k[j] * grey<MODE>(x, row + reflect101(xx + j - r, w), mn, rg);
  out[i] = acc;
}
__global__ void k_blur_cols(const float* __restrict__ in, float* __restrict__ out, int N, int h, int w, Blur b) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= (size_t)N * h * w) return;
  const int xx = (int)(i % w);
  const int y = (int)((i / w) % h);
  const size_t img = i - (size_t)y * w - xx;
  const int r = b.ksize / 2;
  float acc = 0.f;
  for (int j = 0; j < b.ksize; ++j) acc = acc + b.k[j] * in[img + (size_t)reflect101(y + j - r, h) * w + xx];
  out[i] = acc;
}

// ---- cv::resize INTER_LINEAR (centre-aligned, clamped), ch interleaved channels, optional post-scale (flow upsampling) ------
__device__ __forceinline__ void lin_tap(int d, int dst, int src, int& s0, int& s1, float& f) {
  const double scale = (double)src / (double)dst;
  f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f = f - (float)s;
  if (s < 0) f = 0.f, s = 0;
  if (s >= src - 1) f = 0.f, s = src - 1;
  s0 = s;
  s1 = s + 1 < src ? s + 1 : src - 1;
}
__global__ void k_resize_linear(const float* __restrict__ in, float* __restrict__ out, int N, int hi, int wi, int ho, int wo, int ch,
                                float post) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= (size_t)N * ho * wo * ch) return;
  const int c = (int)(i % ch);
  const int xx = (int)((i / ch) % wo);
  const int y = (int)((i / ((size_t)ch * wo)) % ho);
  const size_t n = i / ((size_t)ch * wo * ho);
  int x0, x1, y0, y1;
  float fx, fy;
  lin_tap(xx, wo, wi, x0, x1, fx);
  lin_tap(y, ho, hi, y0, y1, fy);
  const float* a = in + n * (size_t)hi * wi * ch + c;
  const float r0 = a[((size_t)y0 * wi + x0) * ch] * (1.f - fx) + a[((size_t)y0 * wi + x1) * ch] * fx;
  const float r1 = a[((size_t)y1 * wi + x0) * ch] * (1.f - fx) + a[((size_t)y1 * wi + x1) * ch] * fx;
  const float v = r0 * (1.f - fy) + r1 * fy;
  out[i] = post != 1.f ? v * post : v;
}

// ---- FarnebackPolyExp: vertical pass (float) -> V[n,y,x,3]; horizontal pass (double accumulators) -> R[n,y,x,5] -------------
__global__ void k_poly_v(const float* __restrict__ src, float* __restrict__ V, int N, int h, int w, Poly p) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= (size_t)N * h * w) return;
  const int xx = (int)(i % w);
  const int y = (int)((i / w) % h);
  const float* im = src + (i - (size_t)y * w - xx) + xx;
  float t0 = im[(size_t)y * w] * p.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int k = 1; k <= POLY_N; ++k) {
    const float s0 = im[(size_t)max(y - k, 0) * w], s1 = im[(size_t)min(y + k, h - 1) * w];
    const float q = s0 + s1;
    t0 = t0 + p.g[k] * q;
    t1 = t1 + p.xg[k] * (s1 - s0);
    t2 = t2 + p.xxg[k] * q;
  }
  V[3 * i] = t0;
  V[3 * i + 1] = t1;
  V[3 * i + 2] = t2;
}
__global__ void k_poly_h(const float* __restrict__ V, float* __restrict__ R, int N, int h, int w, Poly p) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i >= (size_t)N * h * w) return;
  const int xx = (int)(i % w);
  const float* row = V + 3 * (i - xx);
  const float g0 = p.g[0];
  double b1 = (double)(row[3 * xx] * g0), b2 = 0, b3 = (double)(row[3 * xx + 1] * g0), b4 = 0, b5 = (double)(row[3 * xx + 2] * g0), b6 = 0;
#pragma unroll
  for (int k = 1; k <= POLY_N; ++k) {
    const float* rp = row + 3 * min(xx + k, w - 1);   // replicated borders
    const float* rm = row + 3 * max(xx - k, 0);
    const double tg = (double)(rp[0] + rm[0]);
    b1 += tg * (double)p.g[k];
    b4 += tg * (double)p.xxg[k];
    b2 += (double)((rp[0] - rm[0]) * p.xg[k]);
    b3 += (double)((rp[1] + rm[1]) * p.g[k]);
    b6 += (double)((rp[1] - rm[1]) * p.xg[k]);
    b5 += (double)((rp[2] + rm[2]) * p.g[k]);
  }
  float* d = R + 5 * i;
  d[1] = (float)(b2 * p.ig11);
  d[0] = (float)(b3 * p.ig11);
  d[3] = (float)(b1 * p.ig03 + b4 * p.ig33);
  d[2] = (float)(b1 * p.ig03 + b5 * p.ig33);
  d[4] = (float)(b6 * p.ig55);
}

// ---- FarnebackUpdateMatrices: pair (c, t) uses R of frames t and t+1 of channel c -> M[pair,y,x,5] ---------------------------
__device__ __forceinline__ float border_lo(int i) { return i < 2 ? 0.14f : 0.4472f; }  // border[5] = {.14, .14, .4472, .4472, .4472}
__device__ __forceinline__ float border_w(int i, int n) {
  return (i < 5 ? border_lo(i) : 1.f) * (i >= n - 5 ? border_lo(n - i - 1) : 1.f);
}
__global__ void k_update_matrices(const float* __restrict__ R, const float* __restrict__ flow, float* __restrict__ M, int C, int T,
                                  int h, int w) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  const size_t hw = (size_t)h * w;
  if (i >= (size_t)C * (T - 1) * hw) return;
  const int xx = (int)(i % w);
  const int y = (int)((i / w) % h);
  const size_t pair = i / hw;
  const size_t c = pair / (T - 1), t = pair % (T - 1);
  const float* R0 = R + ((c * T + t) * hw + (size_t)y * w + xx) * 5;
  const float* R1 = R + (c * T + t + 1) * hw * 5;
  const float dx = flow[2 * i], dy = flow[2 * i + 1];
  float fx = (float)xx + dx, fy = (float)y + dy;
  const float flx = floorf(fx), fly = floorf(fy);
  // cvFloor + the unsigned range test of the C++: inside <=> 0 <= x1 < w-1 and 0 <= y1 < h-1 (NaN / huge values fall outside)
  const bool inside = flx >= 0.f && flx < (float)(w - 1) && fly >= 0.f && fly < (float)(h - 1);
  float r2, r3, r4, r5, r6;
  if (inside) {
    const int x1 = (int)flx, y1 = (int)fly;
    fx = fx - flx;
    fy = fy - fly;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    const float* p = R1 + ((size_t)y1 * w + x1) * 5;
    const size_t st = (size_t)w * 5;
    r2 = a00 * p[0] + a01 * p[5] + a10 * p[st] + a11 * p[st + 5];
    r3 = a00 * p[1] + a01 * p[6] + a10 * p[st + 1] + a11 * p[st + 6];
    r4 = a00 * p[2] + a01 * p[7] + a10 * p[st + 2] + a11 * p[st + 7];
    r5 = a00 * p[3] + a01 * p[8] + a10 * p[st + 3] + a11 * p[st + 8];
    r6 = a00 * p[4] + a01 * p[9] + a10 * p[st + 4] + a11 * p[st + 9];
    r4 = (R0[2] + r4) * 0.5f;
    r5 = (R0[3] + r5) * 0.5f;
    r6 = (R0[4] + r6) * 0.25f;
  } else {
    r2 = r3 = 0.f;
    r4 = R0[2];
    r5 = R0[3];
    r6 = R0[4] * 0.5f;
  }
  r2 = (R0[0] - r2) * 0.5f;
  r3 = (R0[1] - r3) * 0.5f;
  r2 += r4 * dy + r6 * dx;
  r3 += r6 * dy + r5 * dx;
  // border attenuation with the C++'s unsigned range test (for w or h < 10 it is NOT "within 5 pixels of an edge")
  const bool edge = (unsigned)(xx - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10);
  float sc = 1.f;
  if (edge) sc = border_w(xx, w) * (y < 5 ? border_lo(y) : 1.f) * (y >= h - 5 ? border_lo(h - y - 1) : 1.f);
  r2 *= sc;
  r3 *= sc;
  r4 *= sc;
  r5 *= sc;
  r6 *= sc;
  float* m = M + 5 * i;
  m[0] = r4 * r4 + r6 * r6;
  m[1] = (r4 + r5) * r6;
  m[2] = r5 * r5 + r6 * r6;
  m[3] = r4 * r2 + r6 * r3;
  m[4] = r6 * r2 + r5 * r3;
}

// ---- FarnebackUpdateFlow_Blur: 15 x 15 box sums with replicated borders in double, then the regularised 2 x 2 solve ---------
__global__ void k_box_v(const float* __restrict__ M, double* __restrict__ S, size_t npair, int h, int w) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;   // one thread per (pair, y, x, comp)
  if (i >= npair * h * w * 5) return;
  const size_t rowlen = (size_t)w * 5;
  const int y = (int)((i / rowlen) % h);
  const size_t base = i - (size_t)y * rowlen;
  double s = 0;
  for (int d = -WIN_M; d <= WIN_M; ++d) s += (double)M[base + (size_t)clampi(y + d, 0, h - 1) * rowlen];
  S[i] = s;
}
__global__ void k_box_h_solve(const double* __restrict__ S, float* __restrict__ flow, size_t npair, int h, int w, int planar) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  const size_t hw = (size_t)h * w;
  if (i >= npair * hw) return;
  const int xx = (int)(i % w);
  const double* row = S + (i - xx) * 5;
  double g11 = 0, g12 = 0, g22 = 0, h1 = 0, h2 = 0;
  for (int d = -WIN_M; d <= WIN_M; ++d) {
    const double* q = row + 5 * clampi(xx + d, 0, w - 1);
    g11 += q[0];
    g12 += q[1];
    g22 += q[2];
    h1 += q[3];
    h2 += q[4];
  }
  const double scale = 1.0 / (WIN * WIN);
  g11 *= scale;
  g12 *= scale;
  g22 *= scale;
  h1 *= scale;
  h2 *= scale;
  const double idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3);
  const float u = (float)((g11 * h2 - g12 * h1) * idet);
  const float v = (float)((g22 * h1 - g12 * h2) * idet);
  if (planar) {  // final layout of the reference: flows.transpose(0, 3, 1, 2) -> [pair][2][h][w]   (SCHED:240-241)
    const size_t pair = i / hw, px = i % hw;
    flow[(pair * 2) * hw + px] = u;
    flow[(pair * 2 + 1) * hw + px] = v;
  } else {
    flow[2 * i] = u;
    flow[2 * i + 1] = v;
  }
}

// ---- host-side plan ----------------------------------------------------------------------------------------------------------------
struct Level {
  double scale, sigma;
  int ksize, w, h;
};
int cv_round(double v) { return (int)std::nearbyint(v); }  // round half to even under the default rounding mode

int plan_levels(int rows, int cols, Level* lv) {
  int k = 0;
  double scale = 1.0;
  while (k < MAX_LEVELS) {
    scale *= 0.5;
    if (cols * scale < MIN_SIZE || rows * scale < MIN_SIZE) break;
    ++k;
  }
  int n = 0;
  for (int l = k; l >= 0; --l) {
    double s = 1.0;
    for (int i = 0; i < l; ++i) s *= 0.5;
    Level& L = lv[n++];
    L.scale = s;
    L.sigma = (1.0 / s - 1.0) * 0.5;
    L.ksize = std::max(cv_round(L.sigma * 5) | 1, 3);
    L.w = cv_round(cols * s);
    L.h = cv_round(rows * s);
  }
  return n;
}

void gaussian_kernel(int ksize, double sigma, Blur* b) {
  static const float small3[3] = {0.25f, 0.5f, 0.25f};
  b->ksize = ksize;
  float tmp[MAX_KSIZE];
  double sum = 0;
  for (int i = 0; i < ksize; ++i) {
    double t;
    if (sigma <= 0 && ksize == 3) {
      t = small3[i];
    } else {
      const double s = sigma > 0 ? sigma : ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
      const double x = i - (ksize - 1) * 0.5;
      t = std::exp(-0.5 / (s * s) * x * x);
    }
    tmp[i] = (float)t;
    sum += tmp[i];
  }
  sum = 1.0 / sum;
  for (int i = 0; i < ksize; ++i) b->k[i] = (float)(tmp[i] * sum);
}

void prepare_poly(Poly* p) {
  const int n = POLY_N;
  const double sigma = 1.2;
  float g[2 * POLY_N + 1];
  double s = 0;
  for (int x = -n; x <= n; ++x) {
    g[x + n] = (float)std::exp(-x * x / (2 * sigma * sigma));
    s += g[x + n];
  }
  s = 1.0 / s;
  for (int x = -n; x <= n; ++x) g[x + n] = (float)(g[x + n] * s);
  for (int k = 0; k <= n; ++k) {
    p->g[k] = g[n + k];
    p->xg[k] = (float)(k * g[n + k]);
    p->xxg[k] = (float)(k * k * g[n + k]);
  }
  double G[6][12] = {};
  for (int y = -n; y <= n; ++y)
    for (int x = -n; x <= n; ++x) {
      const double wgt = (double)g[y + n] * (double)g[x + n];
      G[0][0] += wgt;
      G[1][1] += wgt * x * x;
      G[3][3] += wgt * x * x * x * x;
      G[5][5] += wgt * x * x * y * y;
    }
  G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
  G[4][4] = G[3][3];
  G[3][4] = G[4][3] = G[5][5];
  for (int i = 0; i < 6; ++i) G[i][6 + i] = 1.0;
  for (int c = 0; c < 6; ++c) {  // Gauss-Jordan with partial pivoting (G is symmetric positive definite, 6 x 6)
    int piv = c;
    for (int r = c + 1; r < 6; ++r)
      if (std::fabs(G[r][c]) > std::fabs(G[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 12; ++j) std::swap(G[c][j], G[piv][j]);
    const double d = 1.0 / G[c][c];
    for (int j = 0; j < 12; ++j) G[c][j] *= d;
    for (int r = 0; r < 6; ++r)
      if (r != c) {
        const double f = G[r][c];
        if (f != 0)
          for (int j = 0; j < 12; ++j) G[r][j] -= f * G[c][j];
      }
  }
  p->ig11 = G[1][7];
  p->ig03 = G[0][9];
  p->ig33 = G[3][9];
  p->ig55 = G[5][11];
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
inline unsigned blocks(size_t n) { return (unsigned)((n + NT - 1) / NT); }

}  // namespace

extern "C" size_t wf_farneback_workspace_bytes(int C, int T, int h, int w) {
  if (C <= 0 || T <= 1 || h <= 0 || w <= 0) return 0;
  const size_t N = (size_t)C * T, P = (size_t)C * (T - 1), hw = (size_t)h * w;
  size_t b = 8192;                        // min / range per group + block partials
  b += 3 * al256(N * hw * 4);             // row-blurred, blurred, level image
  b += al256(N * hw * 3 * 4);             // vertical expansion pass
  b += al256(N * hw * 5 * 4);             // R
  b += al256(P * hw * 5 * 4);             // M
  b += al256(P * hw * 5 * 8);             // box column sums (double)
  b += 2 * al256(P * hw * 2 * 4);         // flow (current level, previous level)
  return b;
}

extern "C" int wf_farneback_flows(const void* x, int dt, float* flow, int C, int T, int h, int w, int quant_mode, void* ws, void* stream) {
  WF_CHECK_ARG(x && flow && ws, "wf_farneback_flows: null pointer");
  WF_CHECK_ARG(quant_mode == 0 || (quant_mode == 1 && C <= 64), "wf_farneback_flows: quant_mode must be 0 (Wan) or 1 (LongCat, C <= 64)");
  WF_CHECK_ARG(dt == WF_F32 || dt == WF_BF16, "wf_farneback_flows: dtype %d", dt);
  WF_CHECK_ARG(C > 0 && T > 1 && h > 0 && w > 0, "wf_farneback_flows: bad shape C=%d T=%d h=%d w=%d", C, T, h, w);
  hipStream_t st = (hipStream_t)stream;
  const size_t N = (size_t)C * T, P = (size_t)C * (T - 1), hw = (size_t)h * w;
  char* base = (char*)ws;
  float* mm = (float*)base;             // per group: [0] min, [1] range (one group, or one per channel: <= 64)
  float* part = mm + 128;               // per group up to 480 / groups block partials
  base += 8192;
  auto take = [&](size_t bytes) {
    char* p = base;
    base += al256(bytes);
    return p;
  };
  float* rowb = (float*)take(N * hw * 4);
  float* blur = (float*)take(N * hw * 4);
  float* limg = (float*)take(N * hw * 4);
  float* V = (float*)take(N * hw * 3 * 4);
  float* R = (float*)take(N * hw * 5 * 4);
  float* M = (float*)take(P * hw * 5 * 4);
  double* S = (double*)take(P * hw * 5 * 8);
  float* fl[2] = {(float*)take(P * hw * 2 * 4), (float*)take(P * hw * 2 * 4)};

  TView xv{const_cast<void*>(x), dt};
  const int groups = quant_mode == 1 ? C : 1;
  const size_t per_group = N * hw / groups;
  const int nb = (int)std::max<size_t>(1, std::min<size_t>(480 / groups, (per_group + NT - 1) / NT));
  hipLaunchKernelGGL(k_minmax_partial, dim3(nb, groups), dim3(NT), 0, st, xv, per_group, part);
  hipLaunchKernelGGL(k_minmax_final, dim3(groups), dim3(64), 0, st, part, nb, mm, quant_mode == 1 && dt == WF_BF16 ? 1 : 0);

  Level lv[MAX_LEVELS + 1];
  const int nl = plan_levels(h, w, lv);
  Poly poly;
  prepare_poly(&poly);
  int cur = 0;
  int ph = 0, pw = 0;  // previous (coarser) level size
  for (int li = 0; li < nl; ++li) {
    const Level& L = lv[li];
    const size_t lhw = (size_t)L.h * L.w;
    Blur b;
    gaussian_kernel(L.ksize, L.sigma, &b);
    if (quant_mode == 1)
      hipLaunchKernelGGL(k_quant_blur_rows<1>, dim3(blocks(N * hw)), dim3(NT), 0, st, xv, mm, rowb, (int)N, h, w, b, T);
    else
      hipLaunchKernelGGL(k_quant_blur_rows<0>, dim3(blocks(N * hw)), dim3(NT), 0, st, xv, mm, rowb, (int)N, h, w, b, T);
    hipLaunchKernelGGL(k_blur_cols, dim3(blocks(N * hw)), dim3(NT), 0, st, rowb, blur, (int)N, h, w, b);
    const float* img = blur;
    if (L.h != h || L.w != w) {
      hipLaunchKernelGGL(k_resize_linear, dim3(blocks(N * lhw)), dim3(NT), 0, st, blur, limg, (int)N, h, w, L.h, L.w, 1, 1.f);
      img = limg;
    }
    hipLaunchKernelGGL(k_poly_v, dim3(blocks(N * lhw)), dim3(NT), 0, st, img, V, (int)N, L.h, L.w, poly);
    hipLaunchKernelGGL(k_poly_h, dim3(blocks(N * lhw)), dim3(NT), 0, st, V, R, (int)N, L.h, L.w, poly);
    float* f = fl[cur];
    if (li == 0) {
      hipMemsetAsync(f, 0, P * lhw * 2 * 4, st);
    } else {
      hipLaunchKernelGGL(k_resize_linear, dim3(blocks(P * lhw * 2)), dim3(NT), 0, st, fl[cur ^ 1], f, (int)P, ph, pw, L.h, L.w, 2, 2.0f);
    }
    hipLaunchKernelGGL(k_update_matrices, dim3(blocks(P * lhw)), dim3(NT), 0, st, R, f, M, C, T, L.h, L.w);
    for (int it = 0; it < ITERS; ++it) {
      const bool last = li == nl - 1 && it == ITERS - 1;
      hipLaunchKernelGGL(k_box_v, dim3(blocks(P * lhw * 5)), dim3(NT), 0, st, M, S, P, L.h, L.w);
      hipLaunchKernelGGL(k_box_h_solve, dim3(blocks(P * lhw)), dim3(NT), 0, st, S, last ? flow : f, P, L.h, L.w, last ? 1 : 0);
      if (it < ITERS - 1)
        hipLaunchKernelGGL(k_update_matrices, dim3(blocks(P * lhw)), dim3(NT), 0, st, R, f, M, C, T, L.h, L.w);
    }
    ph = L.h;
    pw = L.w;
    cur ^= 1;
  }
  WF_LAUNCH_CHECK("wf_farneback_flows");
  return WF_OK;
}
