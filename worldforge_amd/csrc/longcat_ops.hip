// HBM-bound companions of the GEMM / attention kernels for the LongCat-Video DiT
// (longcat_for_worldforge/longcat_video/modules: longcat_video_dit.py = LCD, attention.py = LCA, blocks.py = LCB, rope_3d.py = LCR).
// The LongCat residual stream is bf16 (LCD:104, 120: x.to(x_dtype)) and the AdaLN parameters are per latent FRAME (LCD:85-88), so
// the Wan kernels of dit_ops.hip (fp32 stream, one modulation vector per forward) do not apply:
//   wf_lc_ln_modulate   : LayerNorm_FP32 + per-frame modulate (LCB:133-141, LCD:91, 114) or affine (pre_crs_attn_norm, LCD:111)
//   wf_lc_gate_residual : x = bf16(x + gate[frame] * y)  (LCD:102-104, 117-120), gate NULL: x = x + y (LCD:111)
//   wf_lc_norm_heads    : RMSNorm_FP32 over each head's 128 channels (LCB:40-52; LCA:111, 231) + interleaved 3D RoPE
//                         (LCR:32-36, 101-120), written in the attention layout [H][Lout][128]
//   wf_lc_swiglu        : silu(w1 x) * w3 x (LCB:36-37) on the fused [w1 | w3] projection
// One pass each, 16-byte accesses, compiled with -ffp-contract=off (the reference's separate torch ops never fuse a multiply-add); rounding points follow the reference's bf16 flow (every torch op on bf16 tensors rounds its result).
#include "common.h"
#include "mfma.h"

using namespace wf;

namespace {

__device__ __forceinline__ float block_sum_4(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(v[k] << 16);
    f[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* y) {
  return u32x4{pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
}

// x bf16 [L, C] -> out bf16 [L, C]: (x - mean) * rstd * (plus_one + mul[g][c]) + add[g][c], g = row / rows_per_group.
// Two-pass statistics over register-resident data, biased variance (F.layer_norm).
template <int VPT>  // chunks of 8 channels per thread: C <= 2048 * VPT
__global__ __launch_bounds__(256) void k_lc_ln(const uint16_t* __restrict__ x, const float* __restrict__ mul,
                                               const float* __restrict__ add, long mod_ld, int rows_per_group, long row0,
                                               const int* __restrict__ gidx, float plus_one, uint16_t* __restrict__ out, int C,
                                               float eps) {
  __shared__ float sm[4];
  const size_t row = blockIdx.x;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + row * C);
  const int nch = C >> 3;
  float v[VPT][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int id = threadIdx.x + 256 * i;
    if (id < nch) {
      unpack8(xr[id], v[i]);
#pragma unroll
      for (int k = 0; k < 8; k += 2) s += v[i][k] + v[i][k + 1];
    }
  }
  const float mean = block_sum_4(s, sm) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int id = threadIdx.x + 256 * i;
    if (id < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = v[i][k] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(block_sum_4(q, sm) / (float)C + eps);
  const size_t g = gidx ? (size_t)gidx[row0 + row] : (rows_per_group > 0 ? (row0 + row) / rows_per_group : 0);
  const float* mg = mul ? mul + g * mod_ld : nullptr;
  const float* ag = add ? add + g * mod_ld : nullptr;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int id = threadIdx.x + 256 * i;
    if (id < nch) {
      float m[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, y[8];
      if (mg) {
        const float4 m0 = reinterpret_cast<const float4*>(mg)[2 * id], m1 = reinterpret_cast<const float4*>(mg)[2 * id + 1];
        m[0] = m0.x, m[1] = m0.y, m[2] = m0.z, m[3] = m0.w, m[4] = m1.x, m[5] = m1.y, m[6] = m1.z, m[7] = m1.w;
      }
      if (ag) {
        const float4 a0 = reinterpret_cast<const float4*>(ag)[2 * id], a1 = reinterpret_cast<const float4*>(ag)[2 * id + 1];
        a[0] = a0.x, a[1] = a0.y, a[2] = a0.z, a[3] = a0.w, a[4] = a1.x, a[5] = a1.y, a[6] = a1.z, a[7] = a1.w;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) y[k] = (v[i][k] - mean) * rstd * (plus_one + m[k]) + a[k];
      reinterpret_cast<u32x4*>(out + row * C)[id] = pack8(y);
    }
  }
}

// x[r][c] = bf16(x[r][c] + gate[r / rows_per_group][c] * y[r][c]);  n8 = L * C / 8 chunks
__global__ void k_lc_gate_resid(uint16_t* __restrict__ x, const uint16_t* __restrict__ y, long ldy, const float* __restrict__ gate,
                                long gate_ld, int rows_per_group, long row0, const int* __restrict__ gidx, int C, size_t n8) {
  const int cpr = C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / cpr;
    const int ch = (int)(i - row * cpr);
    float xv[8], yv[8], g[8] = {1, 1, 1, 1, 1, 1, 1, 1}, o[8];
    u32x4* xp = reinterpret_cast<u32x4*>(x + row * C) + ch;
    unpack8(*xp, xv);
    unpack8(reinterpret_cast<const u32x4*>(y + row * ldy)[ch], yv);
    if (gate) {
      const size_t grp = gidx ? (size_t)gidx[row0 + row] : (row0 + row) / rows_per_group;
      const float* gp = gate + grp * gate_ld + ch * 8;
      const float4 g0 = reinterpret_cast<const float4*>(gp)[0], g1 = reinterpret_cast<const float4*>(gp)[1];
      g[0] = g0.x, g[1] = g0.y, g[2] = g0.z, g[3] = g0.w, g[4] = g1.x, g[5] = g1.y, g[6] = g1.z, g[7] = g1.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = xv[k] + g[k] * yv[k];  // two torch ops in the reference: built with -ffp-contract=off
    *xp = pack8(o);
  }
}

// in bf16 [L, ld] (head h at columns h*128) -> out bf16 [H][Lout][128].  16 lanes per (row, head), 8 channels per lane.
// y = bf16(bf16(x * rsqrt(mean(x^2) + eps)) * w); RoPE on pairs (2p, 2p+1) with angle table entries cos/sin[row][p] in fp32 -> bf16.
__global__ __launch_bounds__(256) void k_lc_heads(const uint16_t* __restrict__ in, long ld, const float* __restrict__ w,
                                                  const float* __restrict__ cs, const float* __restrict__ sn,
                                                  uint16_t* __restrict__ out, int L, int Lout, int H, float eps, float out_scale) {
  const int row = blockIdx.x;
  const int head = blockIdx.y * 16 + (threadIdx.x >> 4), within = threadIdx.x & 15;
  if (head >= H) return;
  float v[8], y[8];
  unpack8(*reinterpret_cast<const u32x4*>(in + (size_t)row * ld + head * 128 + within * 8), v);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) ss += v[k] * v[k];
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
  const float rinv = rsqrtf(ss * (1.0f / 128.0f) + eps);
  const float4 w0 = reinterpret_cast<const float4*>(w)[2 * within], w1 = reinterpret_cast<const float4*>(w)[2 * within + 1];
  const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) y[k] = rbf(rbf(v[k] * rinv) * ww[k]);
  if (cs) {
    const float4 c4 = reinterpret_cast<const float4*>(cs + (size_t)row * 64)[within];
    const float4 s4 = reinterpret_cast<const float4*>(sn + (size_t)row * 64)[within];
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float re = y[2 * k] * cc[k] - y[2 * k + 1] * sv[k];
      const float im = y[2 * k + 1] * cc[k] + y[2 * k] * sv[k];
      y[2 * k] = re;
      y[2 * k + 1] = im;
    }
  }
  if (out_scale != 1.0f) {  // pre-scaled Q for the attention kernel's exp2-domain form (wf_attn_fwd softmax_scale = 0): folded in front of the one rounding
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] *= out_scale;
  }
  *reinterpret_cast<u32x4*>(out + ((size_t)head * Lout + row) * 128 + within * 8) = pack8(y);
}

// in bf16 [L, ld]: columns [0, Hd) = w1 x, [Hd, 2 Hd) = w3 x -> out bf16 [L, Hd] = bf16(bf16(silu(a)) * b)
__global__ void k_lc_swiglu(const uint16_t* __restrict__ in, long ld, uint16_t* __restrict__ out, int Hd, size_t n8) {
  const int cpr = Hd >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / cpr;
    const int ch = (int)(i - row * cpr);
    float a[8], b[8], o[8];
    unpack8(reinterpret_cast<const u32x4*>(in + row * ld)[ch], a);
    unpack8(reinterpret_cast<const u32x4*>(in + row * ld + Hd)[ch], b);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = rbf(a[k] / (1.0f + __expf(-a[k]))) * b[k];
    reinterpret_cast<u32x4*>(out + row * (size_t)Hd)[ch] = pack8(o);
  }
}

// in bf16 [H][L][128] -> out bf16 [H][L / 128][128]: mean over each block of 128 consecutive rows, fp32 accumulation, one rounding
// (torch's mean on a bf16 tensor; bsa_interface.py:169-179).  grid (L / 128, H), 256 threads = 16 row groups x 16 column chunks.
__global__ __launch_bounds__(256) void k_lc_mean_pool(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int L, int block) {
  __shared__ float sm[16][128];
  const int blk = blockIdx.x, head = blockIdx.y;
  const int ch = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const uint16_t* base = in + ((size_t)head * L + (size_t)blk * block) * 128 + ch * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < block / 16; ++i) {
    float v[8];
    unpack8(*reinterpret_cast<const u32x4*>(base + (size_t)(grp + 16 * i) * 128), v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[grp][ch * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 128) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sm[g][threadIdx.x];
    out[((size_t)head * (L / block) + blk) * 128 + threadIdx.x] = f32_to_bf16(t / (float)block);
  }
}

// out[i][:] = in[idx[i]][:]  (bf16 rows of C elements, 16-byte chunks)
__global__ void k_gather_rows(const uint16_t* __restrict__ in, int64_t ld_in, const int* __restrict__ idx, uint16_t* __restrict__ out,
                              int64_t ld_out, int C, size_t n8) {
  const int cpr = C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / cpr;
    const int ch = (int)(i - row * cpr);
    reinterpret_cast<u32x4*>(out + row * ld_out)[ch] = reinterpret_cast<const u32x4*>(in + (size_t)idx[row] * ld_in)[ch];
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Block selection of the sparse attention (bsa_interface.py:211-224): top n_sel of n_k block scores per query block, turned straight
// into the per-workgroup lists wf_attn_bsa_fwd walks (worldforge_amd/bsa.py group_lists: union of the g query blocks of a workgroup in
// ascending block order, entry = physical block * 2^g + selection flags).  One workgroup per (head, group of g query blocks).
// Per query block: exact n_sel-th largest bf16 score by a two-pass radix select on the 16 score bits (histogram of the high byte, then
// of the low byte inside the boundary bin), everything above it selected, ties at it by ascending block index (what torch's radix
// top-k does on a GPU).  No sort, no host round trip; replaces torch.topk + ~15 small integer tensor ops per layer.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int TK_MAXK = 2048;  // key blocks per row (98 560 tokens / 64 = 1540)
__device__ __forceinline__ uint32_t bf16_sort_key(uint16_t b) { return (b & 0x8000u) ? (uint16_t)~b : (uint16_t)(b | 0x8000u); }  // ascending
// exclusive scan of one int per thread over the 256 threads of the workgroup; returns the prefix, *total = sum of all
__device__ __forceinline__ int block_exscan(int v, int* sm /* >= 4 ints */, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  __syncthreads();  // sm may still be read by the previous scan
  if (lane == 63) sm[wv] = inc;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < wv; ++i) base += sm[i];
  *total = sm[0] + sm[1] + sm[2] + sm[3];
  return base + inc - v;
}
// inclusive scan of one float per thread over the 256 threads (fixed order: deterministic); *total = sum of all
__device__ __forceinline__ float block_incscan_f(float v, float* sm /* >= 4 floats */, float* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  __syncthreads();
  if (lane == 63) sm[wv] = inc;
  __syncthreads();
  float base = 0.f;
  for (int i = 0; i < wv; ++i) base += sm[i];
  *total = ((sm[0] + sm[1]) + sm[2]) + sm[3];
  return base + inc;
}
// bsa_interface.py:226-263 (get_select_indices_cdf / _cdf_topk): per (head, query block) row, the number of key blocks the cdf rule takes =
// how many blocks, in descending weight order, have a cumulative softmax(score / sqrt(128)) weight <= cdf_threshold (searchsorted right),
// at least n_min, at most n_k.  The reference evaluates this chain on its bf16 score tensor, and eager torch rounds to bf16 after every
// op (pinned by tests/golden/g14c_bsa_cdf_bf16.npz, recorded from get_select_indices_cdf_from_score on bf16 scores):
//     x = bf16(score * c)   w = bf16(exp(x - max x) / sum)   [softmax: fp32 inside, bf16 out]   cdf_k = bf16(fp32 running sum of the sorted w)
// and compares float(cdf_k) <= float32(threshold).  Near 0.9 the bf16 spacing is 2^-8, so many consecutive cdf values collapse and the
// count differs from the fp32 rule by several blocks -- the rounding points are reproduced here.  The row's 16-bit score keys are sorted
// in LDS (bitonic, descending; x and w are monotone in the score, so this is the weight order), the weights scanned in that order with a
// fixed tree: every partial sum of bf16 weights >= 2^-17 is exact in fp32 whatever the order, and once a weight is smaller than that the
// cumulative sum is already > 0.97 (n_k <= 2048) -- so for thresholds <= 0.97 the scan equals the sequential cumsum bit for bit.  What can
// still differ from a torch run is one bf16 ulp of a weight where exp / the row sum round differently (about 1 row in 10^3 changes its
// count).  Which blocks = the top `need` of the row: k_bsa_topk_lists with need_rows.
__global__ __launch_bounds__(256) void k_bsa_cdf_need(const uint16_t* __restrict__ scores, long ld, int n_q, int n_k, float thr, int n_min,
                                                      int* __restrict__ need_rows) {
  __shared__ uint16_t key[TK_MAXK];
  __shared__ float smf[4];
  __shared__ int smi[4];
  const int qb = blockIdx.x, head = blockIdx.y, tid = threadIdx.x;
  const uint16_t* row = scores + ((size_t)head * n_q + qb) * ld;
  for (int b = tid; b < TK_MAXK; b += 256) key[b] = b < n_k ? (uint16_t)bf16_sort_key(row[b]) : (uint16_t)0;  // padding sorts last
  __syncthreads();
  for (int k2 = 2; k2 <= TK_MAXK; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < TK_MAXK; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const uint16_t a = key[i], c = key[p];
          const bool desc = (i & k2) == 0;
          if (desc ? a < c : a > c) key[i] = c, key[p] = a;
        }
      }
      __syncthreads();
    }
  auto score_of = [](uint16_t kx) {  // inverse of bf16_sort_key -> float
    const uint16_t b = (kx & 0x8000u) ? (uint16_t)(kx & 0x7fffu) : (uint16_t)~kx;
    return __uint_as_float((uint32_t)b << 16);
  };
  constexpr int PER = TK_MAXK / 256;
  const float c = 0.08838834764831845f;  // 1 / sqrt(128)
  const float mx = rbf(score_of(key[0]) * c);  // bf16(score * sm_scale), the softmax input
  float e[PER], part = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int p = tid * PER + i;
    e[i] = p < n_k ? expf(rbf(score_of(key[p]) * c) - mx) : 0.f;
    part += e[i];
  }
  float total;
  block_incscan_f(part, smf, &total);
  float wsum = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    e[i] = rbf(e[i] / total);  // the softmax weight, as torch.softmax divides; stored as bf16
    wsum += e[i];
  }
  float tot2;
  float run = block_incscan_f(wsum, smf, &tot2) - wsum;  // cumulative weight in front of this thread's first position
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    run += e[i];
    cnt += (tid * PER + i < n_k) && rbf(run) <= thr;  // torch.cumsum on bf16: fp32 running sum, every output rounded to bf16
  }
  int tot;
  block_exscan(cnt, smi, &tot);
  if (tid == 0) need_rows[(size_t)head * n_q + qb] = min(max(tot, n_min), n_k);
}
__global__ __launch_bounds__(256) void k_bsa_topk_lists(const uint16_t* __restrict__ scores, long ld, int n_q, int n_k, int n_sel, int gs,
                                                        int bps, int heads, int* __restrict__ lists, int* __restrict__ counts,
                                                        int max_entries, uint32_t* __restrict__ sel_mask,
                                                        const int* __restrict__ need_rows /* per-row counts (cdf rule) or null */) {
  __shared__ uint16_t key[TK_MAXK];
  __shared__ uint8_t flags[TK_MAXK];
  __shared__ int hist[256];
  __shared__ int sc[4];
  __shared__ int pick[2];  // boundary bin, elements still to take inside it
  const int grp = blockIdx.x, head = blockIdx.y, tid = threadIdx.x;
  const int per = (n_k + 255) / 256;  // contiguous chunk of block indices per thread
  const int b0 = tid * per, b1 = min(b0 + per, n_k);
  for (int b = tid; b < n_k; b += 256) flags[b] = 0;
  for (int r = 0; r < gs; ++r) {
    const int qb = grp * gs + r;
    if (qb >= n_q) break;  // (uniform)
    const uint16_t* row = scores + ((size_t)head * n_q + qb) * ld;
    __syncthreads();
    for (int b = tid; b < n_k; b += 256) key[b] = (uint16_t)bf16_sort_key(row[b]);
    // two radix passes: byte 1 over all keys, byte 0 inside the boundary bin
    int need = need_rows ? need_rows[(size_t)head * n_q + qb] : n_sel;  // (uniform)
    uint32_t prefix = 0;  // high byte of the boundary value after pass 0
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      hist[tid] = 0;
      __syncthreads();
      for (int b = tid; b < n_k; b += 256) {
        const uint32_t kx = key[b];
        if (pass == 0)
          atomicAdd(&hist[kx >> 8], 1);
        else if ((kx >> 8) == prefix)
          atomicAdd(&hist[kx & 255], 1);
      }
      __syncthreads();
      if (tid == 0) {  // walk the 256 bins from the top: 256 steps, once per row and pass
        int left = need, bin = 255;
        while (bin > 0 && hist[bin] < left) left -= hist[bin--];
        pick[0] = bin;
        pick[1] = left;
      }
      __syncthreads();
      if (pass == 0) prefix = (uint32_t)pick[0];
      need = pick[1];
      __syncthreads();
    }
    const uint32_t T = (prefix << 8) | (uint32_t)pick[0];  // the n_sel-th largest key; `need` of the keys equal to it are taken
    int eq = 0;
    for (int b = b0; b < b1; ++b) eq += key[b] == T;
    int tot;
    int before = block_exscan(eq, sc, &tot);
    for (int b = b0; b < b1; ++b) {
      const uint32_t kx = key[b];
      bool take = kx > T;
      if (kx == T) take = before++ < need;
      if (take) flags[b] |= (uint8_t)(1u << r);  // this thread owns block b: no race
    }
    if (sel_mask) {
      __syncthreads();
      const int nw = (n_k + 31) / 32;
      for (int w = tid; w < nw; w += 256) {
        uint32_t m = 0;
        for (int i = 0; i < 32 && w * 32 + i < n_k; ++i) m |= (uint32_t)((flags[w * 32 + i] >> r) & 1u) << i;
        sel_mask[((size_t)head * n_q + qb) * nw + w] = m;
      }
    }
  }
  __syncthreads();
  int mine = 0;
  for (int b = b0; b < b1; ++b) mine += flags[b] != 0;
  int total;
  int pos = block_exscan(mine, sc, &total);
  int* out = lists + ((size_t)head * gridDim.x + grp) * max_entries;
  for (int b = b0; b < b1; ++b)
    if (flags[b]) {
      const int phys = (b / bps) * (heads * bps) + head * bps + b % bps;
      out[pos++] = (phys << gs) | (int)flags[b];
    }
  for (int i = total + tid; i < max_entries; i += 256) out[i] = 0;  // never walked (count below): a valid block all the same
  if (tid == 0) counts[head * gridDim.x + grp] = total;
}

}  // namespace

extern "C" int wf_lc_mean_pool_blocks(const void* in, void* out, int H, int L, int block, void* stream) {
  WF_CHECK_ARG(in && out, "wf_lc_mean_pool_blocks: null pointer");
  WF_CHECK_ARG(block == 128 || block == 64, "wf_lc_mean_pool_blocks: block must be 128 or 64");
  WF_CHECK_ARG(H > 0 && L > 0 && L % block == 0, "wf_lc_mean_pool_blocks: L (%d) must be whole %d-token blocks", L, block);
  hipLaunchKernelGGL(k_lc_mean_pool, dim3(L / block, H), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)out, L, block);
  WF_LAUNCH_CHECK("wf_lc_mean_pool_blocks");
  return WF_OK;
}

extern "C" int wf_bsa_topk_lists(const void* scores, int64_t ld, int heads, int n_q, int n_k, int n_sel, int block, int blocks_per_segment,
                                 int* lists, int* counts, int max_entries, uint32_t* sel_mask, void* stream) {
  WF_CHECK_ARG(scores && lists && counts, "wf_bsa_topk_lists: null pointer");
  WF_CHECK_ARG(block == 128 || block == 64, "wf_bsa_topk_lists: block must be 128 or 64");
  WF_CHECK_ARG(heads > 0 && n_q > 0 && n_k > 0 && n_k <= TK_MAXK && ld >= n_k, "wf_bsa_topk_lists: n_k=%d must be in 1..%d, ld >= n_k", n_k, TK_MAXK);
  WF_CHECK_ARG(n_sel >= 1 && n_sel <= n_k, "wf_bsa_topk_lists: n_sel=%d must be in 1..n_k=%d", n_sel, n_k);
  const int gs = 256 / block;
  WF_CHECK_ARG(blocks_per_segment > 0 && max_entries >= ((long)gs * n_sel < n_k ? gs * n_sel : n_k),
               "wf_bsa_topk_lists: max_entries=%d must hold min(g * n_sel, n_k) entries", max_entries);
  hipLaunchKernelGGL(k_bsa_topk_lists, dim3((n_q + gs - 1) / gs, heads), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)scores, (long)ld,
                     n_q, n_k, n_sel, gs, blocks_per_segment, heads, lists, counts, max_entries, sel_mask, (const int*)nullptr);
  WF_LAUNCH_CHECK("wf_bsa_topk_lists");
  return WF_OK;
}

extern "C" int wf_bsa_cdf_lists(const void* scores, int64_t ld, int heads, int n_q, int n_k, float cdf_threshold, int n_min, int block,
                                int blocks_per_segment, int* lists, int* counts, int max_entries, uint32_t* sel_mask, int* row_counts,
                                void* stream) {
  WF_CHECK_ARG(scores && lists && counts && row_counts, "wf_bsa_cdf_lists: null pointer");
  WF_CHECK_ARG(block == 128 || block == 64, "wf_bsa_cdf_lists: block must be 128 or 64");
  WF_CHECK_ARG(heads > 0 && n_q > 0 && n_k > 0 && n_k <= TK_MAXK && ld >= n_k, "wf_bsa_cdf_lists: n_k=%d must be in 1..%d, ld >= n_k", n_k, TK_MAXK);
  WF_CHECK_ARG(cdf_threshold >= 0.f && n_min >= 0 && n_min <= n_k, "wf_bsa_cdf_lists: cdf_threshold >= 0, n_min in 0..n_k");
  WF_CHECK_ARG(blocks_per_segment > 0 && max_entries >= n_k, "wf_bsa_cdf_lists: max_entries=%d must hold n_k=%d entries (a row may take every block)",
               max_entries, n_k);
  const int gs = 256 / block;
  hipLaunchKernelGGL(k_bsa_cdf_need, dim3(n_q, heads), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)scores, (long)ld, n_q, n_k,
                     cdf_threshold, n_min, row_counts);
  hipLaunchKernelGGL(k_bsa_topk_lists, dim3((n_q + gs - 1) / gs, heads), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)scores, (long)ld,
                     n_q, n_k, 0, gs, blocks_per_segment, heads, lists, counts, max_entries, sel_mask, (const int*)row_counts);
  WF_LAUNCH_CHECK("wf_bsa_cdf_lists");
  return WF_OK;
}

extern "C" int wf_gather_rows_bf16(const void* in, int64_t ld_in, const int* index, void* out, int64_t ld_out, int n_rows, int C,
                                   void* stream) {
  WF_CHECK_ARG(in && out && index, "wf_gather_rows_bf16: null pointer");
  WF_CHECK_ARG(C % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0, "wf_gather_rows_bf16: C, ld_in, ld_out must be multiples of 8");
  WF_CHECK_ARG((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "wf_gather_rows_bf16: 16-byte alignment");
  const size_t n8 = (size_t)n_rows * (C / 8);
  if (n8 == 0) return WF_OK;
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, ld_in, index,
                     (uint16_t*)out, ld_out, C, n8);
  WF_LAUNCH_CHECK("wf_gather_rows_bf16");
  return WF_OK;
}

extern "C" int wf_lc_ln_modulate(const void* x, const float* mul, const float* add, int64_t mod_ld, int rows_per_group, int64_t row0,
                                 const int* group_index, int plus_one, void* out, int L, int C, float eps, void* stream) {
  WF_CHECK_ARG(x && out, "wf_lc_ln_modulate: null pointer");
  WF_CHECK_ARG(C % 8 == 0 && C > 0 && C <= 8192, "wf_lc_ln_modulate: C=%d must be a multiple of 8 and <= 8192", C);
  WF_CHECK_ARG(rows_per_group >= 0 && mod_ld % 4 == 0 && row0 >= 0, "wf_lc_ln_modulate: rows_per_group >= 0, mod_ld %% 4 == 0, row0 >= 0");
  WF_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)mul | (uintptr_t)add) & 15) == 0, "wf_lc_ln_modulate: 16-byte alignment");
  if (L == 0) return WF_OK;
  hipStream_t s = (hipStream_t)stream;
  const float p1 = plus_one ? 1.0f : 0.0f;
  const uint16_t* xi = (const uint16_t*)x;
  uint16_t* oo = (uint16_t*)out;
  if (C <= 2048)
    hipLaunchKernelGGL(k_lc_ln<1>, dim3(L), dim3(256), 0, s, xi, mul, add, mod_ld, rows_per_group, row0, group_index, p1, oo, C, eps);
  else if (C <= 4096)
    hipLaunchKernelGGL(k_lc_ln<2>, dim3(L), dim3(256), 0, s, xi, mul, add, mod_ld, rows_per_group, row0, group_index, p1, oo, C, eps);
  else
    hipLaunchKernelGGL(k_lc_ln<4>, dim3(L), dim3(256), 0, s, xi, mul, add, mod_ld, rows_per_group, row0, group_index, p1, oo, C, eps);
  WF_LAUNCH_CHECK("wf_lc_ln_modulate");
  return WF_OK;
}

extern "C" int wf_lc_gate_residual(void* x, const void* y, int64_t ldy, const float* gate, int64_t gate_ld, int rows_per_group,
                                   int64_t row0, const int* group_index, int L, int C, void* stream) {
  WF_CHECK_ARG(x && y, "wf_lc_gate_residual: null pointer");
  WF_CHECK_ARG(C % 8 == 0 && ldy % 8 == 0 && gate_ld % 4 == 0, "wf_lc_gate_residual: C, ldy %% 8, gate_ld %% 4");
  WF_CHECK_ARG(!gate || rows_per_group > 0 || group_index, "wf_lc_gate_residual: a gate needs rows_per_group > 0 or a group index");
  WF_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gate) & 15) == 0, "wf_lc_gate_residual: 16-byte alignment");
  const size_t n8 = (size_t)L * (C / 8);
  if (n8 == 0) return WF_OK;
  hipLaunchKernelGGL(k_lc_gate_resid, dim3(grid_for(n8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)x,
                     (const uint16_t*)y, ldy, gate, gate_ld, rows_per_group > 0 ? rows_per_group : 1, row0, group_index, C, n8);
  WF_LAUNCH_CHECK("wf_lc_gate_residual");
  return WF_OK;
}

extern "C" int wf_lc_norm_heads(const void* in, int64_t ld, const float* weight, const float* cos_tab, const float* sin_tab, void* out,
                                int L, int Lout, int H, float eps, float out_scale, void* stream) {
  WF_CHECK_ARG(in && weight && out, "wf_lc_norm_heads: null pointer");
  WF_CHECK_ARG(H > 0 && ld % 8 == 0 && Lout >= L, "wf_lc_norm_heads: H > 0, ld %% 8 == 0, Lout >= L");
  WF_CHECK_ARG((cos_tab == nullptr) == (sin_tab == nullptr), "wf_lc_norm_heads: cos/sin must both be given or both null");
  WF_CHECK_ARG((((uintptr_t)in | (uintptr_t)out | (uintptr_t)weight | (uintptr_t)cos_tab | (uintptr_t)sin_tab) & 15) == 0,
               "wf_lc_norm_heads: 16-byte alignment");
  if (L == 0) return WF_OK;
  hipLaunchKernelGGL(k_lc_heads, dim3(L, (H + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, ld, weight, cos_tab,
                     sin_tab, (uint16_t*)out, L, Lout, H, eps, out_scale);
  WF_LAUNCH_CHECK("wf_lc_norm_heads");
  return WF_OK;
}

extern "C" int wf_lc_swiglu(const void* in, int64_t ld, void* out, int L, int Hd, void* stream) {
  WF_CHECK_ARG(in && out, "wf_lc_swiglu: null pointer");
  WF_CHECK_ARG(Hd % 8 == 0 && ld % 8 == 0 && ld >= 2 * (int64_t)Hd, "wf_lc_swiglu: Hd %% 8, ld %% 8, ld >= 2 Hd");
  WF_CHECK_ARG((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "wf_lc_swiglu: 16-byte alignment");
  const size_t n8 = (size_t)L * (Hd / 8);
  if (n8 == 0) return WF_OK;
  hipLaunchKernelGGL(k_lc_swiglu, dim3(grid_for(n8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, ld,
                     (uint16_t*)out, Hd, n8);
  WF_LAUNCH_CHECK("wf_lc_swiglu");
  return WF_OK;
}
