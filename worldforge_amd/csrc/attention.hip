// Fused (flash-style) bidirectional attention forward for the Wan DiT on gfx950:  O = softmax(Q K^T / sqrt(128)) V.
//
// Replaces flash_attention() as called by WanSelfAttention.forward (wan/modules/model.py:149-154 -> attention.py:24-130;
// Lq = Lk = 32 760 tokens at 480p, 40 heads x 128) and by WanI2VCrossAttention.forward (model.py:220-222; Lk = 257 / 512).
// This is the roofline-graded kernel: 4*Lq*Lk*128 flop per head, MFMA-bound.
//
// Layouts (produced by wf_qk_norm_rope / wf_v_transpose in dit_ops.hip):
//   Q  [H][Lq ][128] bf16     K [H][Lkp][128] bf16 (rows >= kv_len are zero)
//   Vt [H][Lkp/64][128][64] bf16  -- V transposed and blocked by 64 keys, so that a KV tile is 16 KiB contiguous and the
//                                    P.V MFMA reads its A operand (d x keys) with plain 16-byte LDS reads, no transpose
//   O  [Lq][H*128] bf16 (token-major: the A operand of the o-projection GEMM)
//
// Structure: one workgroup = 8 waves = 256 query rows of one head; each wave owns 32 query rows for the whole KV sweep.
//   * "swapped" QK^T: S^T[key][q] = mfma(A = K tile, B = Q^T), so a lane owns ONE query column and 32 of the 64 scores of
//     a KV tile: the row max / row sum are in-lane reductions plus one exchange with lane^32;
//   * the K rows are fed to the MFMA with index bits 2 and 3 swapped, which makes the scores a lane holds in registers
//     8m..8m+7 exactly the 8 consecutive keys the P^T B-operand of the P.V MFMA needs: P never leaves registers and
//     needs no cross-lane movement (cvt to bf16 only);
//   * O^T[d][q] += mfma(A = V^T tile, B = P^T): the lane ends up with quads of consecutive head-dim values of its query
//     row -> 8-byte stores;
//   * K and V^T tiles are double-buffered in LDS (64 KiB), staged global -> VGPR -> LDS with the next tile's loads issued
//     before the current tile's MFMAs; 16-byte chunk c of row r is stored at c ^ (r & 15) (K, 256-B rows) resp.
//     c ^ ((r >> 1) & 7) (V^T, 128-B rows): conflict-free ds_read_b128 fragment reads;
//   * online softmax in fp32 with exp2 and the 1/sqrt(d)*log2(e) scale folded into one FMA; O is rescaled only when some
//     row maximum in the wave actually grew (exact, threshold 0);
//   * XCD-aware grid: workgroup b runs on XCD b % 8; all query blocks of a head are given to one XCD so its 32 CUs share
//     that head's K / V^T stream through their L2.
#include "common.h"
#include "mfma.h"

using namespace wf;

namespace {

constexpr int D = 128;
constexpr int QB = 256;  // query rows per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int NT = 512;
constexpr int K_TILE_BYTES = KB * D * 2;  // 16 KiB
constexpr int V_TILE_BYTES = D * KB * 2;  // 16 KiB

struct AttnArgs {
  const uint16_t* Q;
  const uint16_t* K;
  const uint16_t* Vt;
  uint16_t* O;
  int H, Lq, Lkp, kv_len, ldo;
  int seg_len;  // keys per K/V segment (= Lkp, or the per-rank shard length when K/V were all-gathered: [P][H][seg_len][128])
  int n_qblk;
  float scale_log2;  // softmax_scale * log2(e)
  int accumulate;    // O += result (second cross-attention, model.py:227)
};

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

__global__ __launch_bounds__(NT, 2) void k_attn(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [buf0: K tile | V^T tile][buf1: K tile | V^T tile]
  constexpr int BUF_BYTES = K_TILE_BYTES + V_TILE_BYTES;

  // ---- XCD-aware (head, q-block) assignment: heads are dealt round-robin to the 8 XCDs -----------------------------
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int hslot = j / a.n_qblk;  // how many heads this XCD has completed
  const int head = hslot * 8 + xcd;
  const int qblk = j % a.n_qblk;
  if (head >= a.H) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int q_row = qblk * QB + wid * 32 + l31;
  const int q_ld = min(q_row, a.Lq - 1);

  // ---- Q^T B-operand fragments: 8 k-steps x 8 bf16 ---------------------------------------------------------------------
  bf16x8 qf[8];
  {
    const uint16_t* qp = a.Q + ((size_t)head * a.Lq + q_ld) * D;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = as_bf16x8(*reinterpret_cast<const u32x4*>(qp + 16 * s + 8 * hi));
  }

  // ---- staging: K tile 1024 chunks, V tile 1024 chunks, 512 threads -> 2 + 2 per thread -------------------------------
  // K/V may be a concatenation of P per-rank segments [P][H][seg_len][128] (sequence-parallel all-gather); seg_len % 64 == 0,
  // so a 64-key tile never straddles two segments.  Single GPU: seg_len == Lkp, one segment.
  const int tiles_per_seg = a.seg_len / KB;
  u32x4 rK[2], rV[2];
  int koff[2], voff[2], ksrc[2], vsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int id = tid + NT * i;
    int kr = id >> 4, kc = id & 15;  // K: 64 rows x 16 chunks
    ksrc[i] = kr * D + kc * 8;
    koff[i] = kr * 256 + ((kc ^ (kr & 15)) << 4);
    int vr = id >> 3, vc = id & 7;  // V^T: 128 rows x 8 chunks
    vsrc[i] = vr * KB + vc * 8;
    voff[i] = vr * 128 + ((vc ^ ((vr >> 1) & 7)) << 4);
  }
  auto gload = [&](int t) {
    const int seg = t / tiles_per_seg;
    const size_t tile_off = ((size_t)(seg * a.H + head) * tiles_per_seg + (t - seg * tiles_per_seg)) * (KB * D);
    const uint16_t* kp = a.K + tile_off;
    const uint16_t* vp = a.Vt + tile_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rK[i] = *reinterpret_cast<const u32x4*>(kp + ksrc[i]);
      rV[i] = *reinterpret_cast<const u32x4*>(vp + vsrc[i]);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(smem + buf * BUF_BYTES + koff[i]) = rK[i];
      *reinterpret_cast<u32x4*>(smem + buf * BUF_BYTES + K_TILE_BYTES + voff[i]) = rV[i];
    }
  };

  // fragment read offsets
  int krow_off[2], krow_sw[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    int kr = kb * 32 + swap23(l31);  // K row fed to MFMA row l31 (bits 2,3 swapped)
    krow_off[kb] = kr * 256;
    krow_sw[kb] = kr & 15;
  }
  int vrow_off[4], vrow_sw[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    int vr = db * 32 + l31;
    vrow_off[db] = vr * 128;
    vrow_sw[db] = (vr >> 1) & 7;
  }

  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = a.scale_log2;

  const int ntiles = (a.kv_len + KB - 1) / KB;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const unsigned char* sKb = smem + buf * BUF_BYTES;
    const unsigned char* sVb = sKb + K_TILE_BYTES;
    if (t + 1 < ntiles) gload(t + 1);

    // ---- S^T = K Q^T : two 32-key blocks ------------------------------------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int ch = 2 * st + hi;
        bf16x8 kf = as_bf16x8(*reinterpret_cast<const u32x4*>(sKb + krow_off[kb] + ((ch ^ krow_sw[kb]) << 4)));
        s[kb] = mfma32(kf, qf[st], s[kb]);
      }
    }
    // lane holds, for query column l31: register r of block kb  <->  key  t*64 + 32*kb + 16*(r>>3) + 8*hi + (r&7)
    if (t == ntiles - 1 && (a.kv_len & (KB - 1))) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = t * KB + 32 * kb + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= a.kv_len) s[kb][r] = -INFINITY;
        }
    }
    // ---- online softmax ----------------------------------------------------------------------------------------------
    float mloc = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kb][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(c * (m_run - m_new));
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      l_run *= alpha;
      m_run = m_new;
    }
    const float mc = c * m_run;
    float lsum = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(c * s[kb][r] - mc);
        lsum += p[r];
      }
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        u32x4 pk = {pack_bf16x2(p[8 * m2 + 0], p[8 * m2 + 1]), pack_bf16x2(p[8 * m2 + 2], p[8 * m2 + 3]),
                    pack_bf16x2(p[8 * m2 + 4], p[8 * m2 + 5]), pack_bf16x2(p[8 * m2 + 6], p[8 * m2 + 7])};
        pf[kb * 2 + m2] = as_bf16x8(pk);
      }
    }
    l_run += lsum;

    // ---- O^T += V^T P^T : 4 head-dim blocks x 4 key steps ------------------------------------------------------------
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        const int ch = 2 * m4 + hi;
        bf16x8 vf = as_bf16x8(*reinterpret_cast<const u32x4*>(sVb + vrow_off[db] + ((ch ^ vrow_sw[db]) << 4)));
        o[db] = mfma32(vf, pf[m4], o[db]);
      }
    }
    if (t + 1 < ntiles) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- finish: combine the two half-wave partial sums, normalise, store ------------------------------------------------
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  if (q_row < a.Lq) {
    uint16_t* op = a.O + (size_t)q_row * a.ldo + head * D;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hi;
        float v0 = o[db][4 * g + 0] * inv, v1 = o[db][4 * g + 1] * inv, v2 = o[db][4 * g + 2] * inv,
              v3 = o[db][4 * g + 3] * inv;
        if (a.accumulate) {
          u32x2 old = *reinterpret_cast<const u32x2*>(op + d);
          v0 += __uint_as_float(old[0] << 16);
          v1 += __uint_as_float(old[0] & 0xffff0000u);
          v2 += __uint_as_float(old[1] << 16);
          v3 += __uint_as_float(old[1] & 0xffff0000u);
        }
        u32x2 pk = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
        *reinterpret_cast<u32x2*>(op + d) = pk;
      }
    }
  }
}

}  // namespace

extern "C" int wf_attn_fwd(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int kv_len, int seg_len,
                           int ldo, float softmax_scale, int accumulate, void* stream) {
  WF_CHECK_ARG(Q && K && Vt && O, "wf_attn_fwd: null pointer");
  WF_CHECK_ARG(H > 0 && Lq > 0 && kv_len > 0, "wf_attn_fwd: empty problem");
  WF_CHECK_ARG(Lkp % KB == 0 && kv_len <= Lkp, "wf_attn_fwd: Lkp (%d) must be a multiple of 64 and >= kv_len (%d)", Lkp,
               kv_len);
  WF_CHECK_ARG(seg_len > 0 && seg_len % KB == 0 && Lkp % seg_len == 0, "wf_attn_fwd: seg_len (%d) must be a multiple of 64 dividing Lkp",
               seg_len);
  WF_CHECK_ARG(ldo % 4 == 0 && ldo >= H * D, "wf_attn_fwd: bad ldo %d", ldo);
  WF_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O) & 15) == 0, "wf_attn_fwd: 16-byte alignment");
  AttnArgs a;
  a.Q = (const uint16_t*)Q;
  a.K = (const uint16_t*)K;
  a.Vt = (const uint16_t*)Vt;
  a.O = (uint16_t*)O;
  a.H = H;
  a.Lq = Lq;
  a.Lkp = Lkp;
  a.kv_len = kv_len;
  a.seg_len = seg_len;
  a.ldo = ldo;
  a.n_qblk = ceil_div(Lq, QB);
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  a.accumulate = accumulate;
  const int hslots = (H + 7) / 8;
  const int grid = hslots * a.n_qblk * 8;
  hipLaunchKernelGGL(k_attn, dim3(grid), dim3(NT), 2 * (K_TILE_BYTES + V_TILE_BYTES), (hipStream_t)stream, a);
  WF_LAUNCH_CHECK("wf_attn_fwd");
  return WF_OK;
}
