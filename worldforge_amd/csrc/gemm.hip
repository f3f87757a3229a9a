// bf16 MFMA GEMM for the DiT / VAE linear layers:  out[M,N] = epilogue( X[M,K] . W[N,K]^T + bias[N] )
//
// Replaces every nn.Linear of wan/modules/model.py (q/k/v/o :123-126, ffn :271-273, embeddings :456-464, head :332) and
// the 1x1 convolutions of wan/modules/vae.py (:199, :234-235).  MFMA-bound (2*M*N*K flop).
//
// Design (gfx950, 64-lane waves):
//   * workgroup tile 128(tokens) x 128(features), BK = 64, 4 waves in a 2x2 grid, each wave 64x64 = 2x2 MFMA 32x32x16 tiles;
//     the WEIGHT rows are the MFMA A operand and the TOKEN rows the B operand, so each lane ends up owning one token row
//     and quads of consecutive output features (see mfma.h) -> 8/16-byte epilogue stores and vector bias / gate loads;
//   * both operands are K-contiguous (nn.Linear stores W as [N,K]); tiles are staged global -> VGPR -> LDS with 16-byte
//     accesses, next tile's global loads issued before the current tile's MFMAs and written to the other LDS buffer
//     after them (one barrier per K tile);
//   * LDS rows are 128 B; 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7): conflict-free for the
//     ds_read_b128 lane groups of the fragment reads and for the ds_write_b128 staging writes;
//   * XCD-aware rasterisation: workgroups of one XCD (blockIdx % 8) walk 8x8 super-tiles so that the 32 CUs sharing an
//     L2 re-use the same 8 token panels and 8 weight panels;
//   * ragged M / N are handled by clamping load rows and predicating stores; K tail chunks are zero-filled (K % 8 == 0).
#include "common.h"
#include "mfma.h"
#include <stdlib.h>

using namespace wf;

namespace {

constexpr int BM = 128;  // token rows per workgroup tile
constexpr int BN = 128;  // output features per workgroup tile
constexpr int BK = 64;
constexpr int NTHREADS = 256;
constexpr int ROW_BYTES = BK * 2;             // 128
constexpr int TILE_BYTES = BM * ROW_BYTES;    // 16 KiB per operand tile
constexpr int CHUNKS_PER_THREAD = (BM * 8) / NTHREADS;  // 4 x 16-byte chunks per operand per thread

struct GemmArgs {
  const uint16_t* X;  // [M, ldx] bf16
  const uint16_t* W;  // [N, K] bf16
  const float* bias;  // [N] or null
  void* out;          // bf16 / f32 [M, ldo]
  const float* gate;  // [N] f32 (EPI_RESID) or null
  int M, N, K, ldx, ldw, ldo;
  int mt, nt;  // tile counts
  // batched launches of k_gemm (blockIdx.y = batch b): X, W, out of problem b start b * bsx / bsw / bso ELEMENTS further (0 = not batched)
  long bsx, bsw, bso;
  int f16;  // X, W and a 16-bit output are fp16 instead of bf16 (wf_gemm_f16: the VAE's fp16 operand formats)
  int batch;  // gridDim.y of a batched launch (1 otherwise)
  float acc_scale;  // F16 instantiations only: out = epilogue(acc * acc_scale + bias) -- 2^-k when the fp16 weight operand is stored scaled by 2^k
};

enum { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32 = 2, EPI_RESID = 3, EPI_F32_ACC = 4 };

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ float gelu_tanh(float x) {
  // nn.GELU(approximate='tanh') (model.py:272): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).  0.5 (1 + tanh u) is the logistic
  // function of 2u, so gelu = x / (1 + exp(-2u)) = x * rcp(1 + exp2(c x (1 + 0.044715 x^2))) with c = -2 sqrt(2/pi) log2(e): six VALU and
  // two transcendentals per value where the textbook form took about twelve and two (round 4: the GELU was 13 k of the 22 k cycles of
  // the FFN-up epilogue, profiles/r4_c_gemm_pp_cycles.md).  x -> +inf: exp2 -> 0, result x; x -> -inf: exp2 -> inf, rcp -> 0, result -0.
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f;
  const float p = __builtin_fmaf(k1, x * x, 1.0f);
  const float e = __builtin_amdgcn_exp2f((c * x) * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int EPI, bool F16 = false>
__global__ __launch_bounds__(NTHREADS, 2) void k_gemm(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // ---- XCD-aware tile assignment --------------------------------------------------------------------------
  const int smt = (a.mt + 7) >> 3, snt = (a.nt + 7) >> 3;
  const int nsuper = smt * snt;
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int gid = (j >> 6) * 8 + xcd;
  if (gid >= nsuper) return;
  const int within = j & 63;
  const int tm = (gid / snt) * 8 + (within >> 3);
  const int tn = (gid % snt) * 8 + (within & 7);
  if (tm >= a.mt || tn >= a.nt) return;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 1, wm = wid & 1;  // wave's 64x64 sub-tile: features wn*64.., tokens wm*64..
  const int l31 = lane & 31, hi = lane >> 5;

  // LDS: [buf0: W tile | X tile][buf1: W tile | X tile]

  // ---- staging geometry: chunk id = tid + 256*i -> (row = id >> 3, chunk = id & 7) ----------------------------------
  const int kchunks = a.K >> 3;
  u32x4 rW[CHUNKS_PER_THREAD], rX[CHUNKS_PER_THREAD];
  const uint16_t* pW[CHUNKS_PER_THREAD];
  const uint16_t* pX[CHUNKS_PER_THREAD];
  int ldsoff[CHUNKS_PER_THREAD];
  const int ck = tid & 7;
#pragma unroll
  for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
    int row = (tid >> 3) + 32 * i;
    int wr = min(n0 + row, a.N - 1);
    int xr = min(m0 + row, a.M - 1);
    pW[i] = a.W + (size_t)blockIdx.y * a.bsw + (size_t)wr * a.ldw + ck * 8;
    pX[i] = a.X + (size_t)blockIdx.y * a.bsx + (size_t)xr * a.ldx + ck * 8;
    ldsoff[i] = row * ROW_BYTES + swz(row, ck) * 16;
  }
  bool stage_ok = true;
  auto gload = [&](int kt) {
    // K-tail chunks (only in the last K tile of a K % 64 != 0 problem) must read as zero.  The load always targets a valid
    // (clamped) address and the zero-select is applied when the registers are written to LDS, AFTER the MFMAs of the current
    // tile: neither exec-mask branches nor an early use of the load result sit between the load issue and those MFMAs.
    const int kc = kt * 8 + ck;
    stage_ok = kc < kchunks;
    const size_t koff = (size_t)min(kc, kchunks - 1) * 8 - (size_t)ck * 8;
#pragma unroll
    for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
      rW[i] = *reinterpret_cast<const u32x4*>(pW[i] + koff);
      rX[i] = *reinterpret_cast<const u32x4*>(pX[i] + koff);
    }
  };
  auto lstore = [&](int buf) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < CHUNKS_PER_THREAD; ++i) {
      *reinterpret_cast<u32x4*>(smem + buf * (2 * TILE_BYTES) + ldsoff[i]) = stage_ok ? rW[i] : z;
      *reinterpret_cast<u32x4*>(smem + buf * (2 * TILE_BYTES) + TILE_BYTES + ldsoff[i]) = stage_ok ? rX[i] : z;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

  // fragment row offsets (constant over K)
  int offW[2], offX[2], rswW[2], rswX[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int rw = wn * 64 + i * 32 + l31;
    int rx = wm * 64 + i * 32 + l31;
    offW[i] = rw * ROW_BYTES;
    offX[i] = rx * ROW_BYTES;
    rswW[i] = (rw >> 1) & 7;
    rswX[i] = (rx >> 1) & 7;
  }

  const int nk = (a.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const unsigned char* sWb = smem + buf * (2 * TILE_BYTES);
    const unsigned char* sXb = sWb + TILE_BYTES;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int c = 2 * s + hi;
      u32x4 fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fw[i] = *reinterpret_cast<const u32x4*>(sWb + offW[i] + ((c ^ rswW[i]) << 4));
        fx[i] = *reinterpret_cast<const u32x4*>(sXb + offX[i] + ((c ^ rswX[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = mfma32t<F16>(fw[i], fx[jn], acc[i][jn]);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns token row (m) and feature quads -----------------------------------------------------------
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
    const int m = m0 + wm * 64 + jn * 32 + l31;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * hi;
        if (n >= a.N) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = F16 ? acc[i][jn][4 * g + q] * a.acc_scale : acc[i][jn][4 * g + q];
        if (a.bias) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += bb[q];
        }
        const size_t o = (size_t)m * a.ldo + n;
        if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
          if constexpr (EPI == EPI_BF16_GELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = gelu_tanh(v[q]);
          }
          u32x2 pk;
          if constexpr (F16)
            pk = u32x2{pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3])};
          else
            pk = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + (size_t)blockIdx.y * a.bso + o) = pk;
        } else if constexpr (EPI == EPI_F32) {
          f32x4 ov = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)blockIdx.y * a.bso + o) = ov;
        } else if constexpr (EPI == EPI_F32_ACC) {
          float* po = reinterpret_cast<float*>(a.out) + (size_t)blockIdx.y * a.bso + o;
          f32x4 old = *reinterpret_cast<const f32x4*>(po);
          f32x4 ov = {old[0] + v[0], old[1] + v[1], old[2] + v[2], old[3] + v[3]};
          *reinterpret_cast<f32x4*>(po) = ov;
        } else {  // EPI_RESID: x += (acc + bias) * gate   (model.py:306, 310, 313)
          float* po = reinterpret_cast<float*>(a.out) + (size_t)blockIdx.y * a.bso + o;
          f32x4 old = *reinterpret_cast<const f32x4*>(po);
          f32x4 gg = {1.f, 1.f, 1.f, 1.f};
          if (a.gate) gg = *reinterpret_cast<const f32x4*>(a.gate + n);
          f32x4 ov = {old[0] + v[0] * gg[0], old[1] + v[1] * gg[1], old[2] + v[2] * gg[2], old[3] + v[3] * gg[3]};
          *reinterpret_cast<f32x4*>(po) = ov;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Large-problem variant: 256 x 256 tile, 8 waves (2 token-halves x 4 feature-quarters, wave tile 128 tokens x 64 features =
// 4 x 2 MFMA 32x32x16 tiles), BK = 64, LDS-DMA staging into a double buffer (128 KiB), two-group ping-pong schedule.
//
// Each K tile is processed in four phases per wave: R0 (LDS reads of k-steps 0,1 -> 12 fragment quads), M0 (16 MFMAs, registers
// only), R1, M1.  Waves 0-3 (group A) and 4-7 (group B, one wave of each group per SIMD) run the same phase sequence ONE PHASE
// APART, phase-locked by raw s_barrier, so that on every SIMD one wave's MFMA phase always coincides with the other wave's
// LDS-read / DMA-issue phase: the matrix pipe never waits for LDS latency.  All waves issue the LDS-DMA of tile t+1 right after
// barrier 4t (its buffer was last read two intervals earlier) and drain it (vmcnt(0)) before barrier 4t+4.
// Requirements: K % 64 == 0 (no K tail in the DMA path); ragged M / N handled by clamped source rows + predicated stores.
// ------------------------------------------------------------------------------------------------------------------
#ifndef WF_GEMM_ABLATE
#define WF_GEMM_ABLATE 0  // lab only (tools/gemm_pp_cycles.py, wrong results): 1 no LDS-DMA pieces in the K loop of k_gemm_pp, 2 only the K tile's
#endif                    // last barrier (no phase hand-over barriers), 4 no LDS fragment reads in the loop, 8 no s_setprio around the MFMA phases
#ifndef WF_GEMM_DMA_PHASE
// k_gemm_pp, where / how a wave issues the LDS-DMA pieces of the next K tile.  0 = rounds 1-3: in the gaps of its MFMA phase through the
// builtin (per piece a 64-bit VALU address add, three scalar instructions and a branch: ~60 issue cycles where an MFMA leaves ~12 free --
// 42 pipe cycles lost per piece, 12.7 % of the K loop: profiles/r4_f_gemm_ablate.md).  1 = in the wave's own READ phase (lab negative: the
// read phase then outlasts the partner's MFMA phase, K loop 84.5 -> 70 %).  2 = round 4: in the MFMA gaps in the saddr form of the attention
// kernel (wave-uniform 64-bit base + constant 32-bit lane offset, M0 in one s_add with an immediate, no branch per piece).
#define WF_GEMM_DMA_PHASE 2
#endif
#ifndef WF_GEMM_DMA_RSPLIT
#define WF_GEMM_DMA_RSPLIT 5  // (PHASE 2) how many of a wave's 9 (8) pieces go to the tail of its first READ phase instead of the MFMA gaps (lab: 0 -> 89.2 %, 3 -> 90.8, 5 -> 92-93.5, 6 / 7 -> 92.4, 9 -> 86.6 % of the pipe)
#endif
#ifdef WF_GEMM_TIMING
// lab builds (WF_EXTRA_HIPCC_FLAGS=-DWF_GEMM_TIMING): slots [0..7] belonged to k_gemm_w4 (removed in round 5); k_gemm_pp adds, per workgroup (its
// wave 0): [8] prologue, [9] K loop, [10] epilogue shader cycles, [11] workgroups, [12] cycles of wave 4 (group B) K loop
__device__ unsigned long long g_gemm_cycles[16];
#endif
constexpr int PM = 256, PN = 256, PK = 64, PT = 512;
constexpr int P_TILE = PM * PK * 2;  // 32 KiB per operand tile
constexpr int P_BUF = 2 * P_TILE;    // W tile | X tile


// Tile geometry.  NI = 32-feature MFMA tiles per wave, NJ = 32-token MFMA tiles per wave:
//   NI = 2, NJ = 4: 256 tokens x 256 features; waves = 2 token halves (= ping-pong group) x 4 feature quarters;
//   NI = 5, NJ = 2: 256 tokens x 320 features; waves = 4 token quarters x 2 feature halves (= ping-pong group).  N = 5120 is
//   16 x 320: with 4096 / 8192 tokens per rank (8 / 4 ranks) that is exactly 1 / 2 rounds of 256 workgroups where the 256-wide tile
//   needs 1.25 / 2.5; the wider tile also reads 0.70 fragment quads per MFMA instead of 0.75.
template <int NI>
struct PPGeom {
  static constexpr int NJ = NI == 2 ? 4 : 2;
  static constexpr int TSPLIT = PM / (NJ * 32);       // waves along tokens
  static constexpr int FSPLIT = 8 / TSPLIT;           // waves along features
  static constexpr int PNT = FSPLIT * NI * 32;        // features per workgroup tile
  static constexpr int W_TILE = PNT * PK * 2;         // bytes
  static constexpr int X_TILE = PM * PK * 2;
  static constexpr int BUF = W_TILE + X_TILE;
  static constexpr int NWP = PNT / 64;                // 1 KiB W pieces per wave
  static constexpr int NP = NWP + 4;                  // LDS-DMA pieces per wave per K tile
  static constexpr int STG = NJ * 32 * 144;           // epilogue staging bytes per wave
  // + a 1 KiB dummy LDS-DMA target per wave (k_gemm_pp) BEHIND both uses of the rest: for NI = 2 the epilogue staging (8 x 18 KiB) is
  // larger than the operand buffers, and a dummy region at 2 * BUF lay inside wave 7's staging rows -- a group-B wave's last dummy pieces
  // could land there after wave 7 had begun its epilogue (seen as a rare wrong tile when several streams shared the GPU)
  static constexpr int DUMMY = 2 * BUF > 8 * STG ? 2 * BUF : 8 * STG;
  static constexpr int LDS = DUMMY + 8 * 1024;
};

template <int EPI, int NI, bool F16 = false>
__global__ __launch_bounds__(PT, 2) void k_gemm_pp(GemmArgs a) {
  using G = PPGeom<NI>;
  constexpr int NJ = G::NJ, NWP = G::NWP, NP = G::NP, W_TILE = G::W_TILE, BUF = G::BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // batched launch (wf_gemm_*_batched): problem blockIdx.y starts bsx / bsw / bso ELEMENTS further (all zero, and gridDim.y = 1, otherwise)
  a.X += (size_t)blockIdx.y * a.bsx;
  a.W += (size_t)blockIdx.y * a.bsw;
  a.out = reinterpret_cast<unsigned char*>(a.out) + (size_t)blockIdx.y * a.bso * ((EPI == EPI_BF16 || EPI == EPI_BF16_GELU) ? 2 : 4);
  // XCD-aware tile assignment (4 x 4 super-tiles of 256 x 256 tiles per XCD pass)
  const int smt = (a.mt + 3) >> 2, snt = (a.nt + 3) >> 2;
  const int nsuper = smt * snt;
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int gid = (j >> 4) * 8 + xcd;
  if (gid >= nsuper) return;
  const int within = j & 15;
  const int tm = (gid / snt) * 4 + (within >> 2);
  const int tn = (gid % snt) * 4 + (within & 3);
  if (tm >= a.mt || tn >= a.nt) return;
  const int m0 = tm * PM, n0 = tn * G::PNT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const bool groupB = wid >= 4;
  const int wfi = NJ == 4 ? (wid & 3) : (wid >> 2);  // wave index along features
  const int wti = NJ == 4 ? (wid >> 2) : (wid & 3);  // wave index along tokens
  const int wf0 = wfi * NI * 32, wt0 = wti * NJ * 32;  // wave tile origin inside the workgroup tile

  // ---- LDS-DMA geometry: an operand tile = rows x 128 B = pieces of 1 KiB (8 rows); wave w moves pieces NWP*w.. of W and
  // 4w..4w+3 of X.  lane -> (row = 8*piece + lane/8, slot = lane%8) receives source chunk slot ^ ((row >> 1) & 7).
  const uint16_t* srcW[NWP];
  const uint16_t* srcX[4];
#pragma unroll
  for (int i = 0; i < NWP; ++i) {
    const int row = 8 * (wid * NWP + i) + (lane >> 3), slot = lane & 7;
    srcW[i] = a.W + (size_t)min(n0 + row, a.N - 1) * a.ldw + (slot ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (wid * 4 + i) + (lane >> 3), slot = lane & 7;
    srcX[i] = a.X + (size_t)min(m0 + row, a.M - 1) * a.ldx + (slot ^ ((row >> 1) & 7)) * 8;
  }
  auto dma_piece = [&](int kt, int i) {  // i in 0..NP-1: W pieces first, then the 4 X pieces
    unsigned char* base = smem + (kt & 1) * BUF;
    if (i < NWP)
      glds16(srcW[i] + (size_t)kt * PK, base + (wid * NWP + i) * 1024);
    else
      glds16(srcX[i - NWP] + (size_t)kt * PK, base + W_TILE + (wid * 4 + i - NWP) * 1024);
  };
  auto dma = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) dma_piece(kt, i);
  };
#if WF_GEMM_DMA_PHASE == 2
  // saddr form: per-lane BYTE offsets from a.W / a.X (constant over K; the launcher guarantees N * ldw * 2 and M * ldx * 2 < 4 GiB)
  uint32_t voffW[NWP], voffX[4];
#pragma unroll
  for (int i = 0; i < NWP; ++i) voffW[i] = (uint32_t)(reinterpret_cast<const unsigned char*>(srcW[i]) - reinterpret_cast<const unsigned char*>(a.W));
#pragma unroll
  for (int i = 0; i < 4; ++i) voffX[i] = (uint32_t)(reinterpret_cast<const unsigned char*>(srcX[i]) - reinterpret_cast<const unsigned char*>(a.X));
  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_offset(smem));
  const int nk_ = a.K / PK;
  // Past the last K tile the pieces are still issued (no branch in the MFMA stream, and ONE code path for the MFMA phase: two copies of it
  // behind a branch made the register allocator spill accumulators): they re-read tile nk-1 into a wave-private 1 KiB dummy region behind the
  // operand buffers (G::LDS reserves it), which nobody reads.
  auto piece_s = [&](int i, int kt) {  // i is a constant after unrolling; kt may be >= nk
    const bool live = kt < nk_;
    const int ks_ = live ? kt : nk_ - 1;
    const uint32_t buf = smem_base + (uint32_t)((kt & 1) * BUF);
    static_assert(G::DUMMY >= 2 * BUF && G::DUMMY >= 8 * G::STG && G::LDS >= G::DUMMY + 8 * 1024, "the dummy LDS-DMA targets overlap live LDS");
    const uint32_t dummy = smem_base + (uint32_t)(G::DUMMY + wid * 1024);
    if (i < NWP)
      glds16_saddr(reinterpret_cast<const unsigned char*>(a.W) + (size_t)ks_ * (PK * 2), voffW[i],
                   live ? buf + (uint32_t)((wid * NWP + i) * 1024) : dummy);
    else
      glds16_saddr(reinterpret_cast<const unsigned char*>(a.X) + (size_t)ks_ * (PK * 2), voffX[i - NWP],
                   live ? buf + (uint32_t)(W_TILE + (wid * 4 + i - NWP) * 1024) : dummy);
  };
#endif

  // ---- fragment addressing ------------------------------------------------------------------------------------------------
  // every fragment row of a wave is its lane's row l31 plus a multiple of 32: one swizzle term serves all of them
  const int sw = (l31 >> 1) & 7;
  const int offW = (wf0 + l31) * 128, offX = W_TILE + (wt0 + l31) * 128;
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int jx = 0; jx < NJ; ++jx)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jx][r] = 0.f;

  u32x4 fw[2][NI], fx[2][NJ];
  auto read_half = [&](int kt, int half) {
    const unsigned char* base = smem + (kt & 1) * BUF;
    if ((WF_GEMM_ABLATE & 4) && kt > 0) return;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ((2 * (2 * half + ks) + hi) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[ks][i] = *reinterpret_cast<const u32x4*>(base + offW + i * 4096 + c);
#pragma unroll
      for (int jx = 0; jx < NJ; ++jx) fx[ks][jx] = *reinterpret_cast<const u32x4*>(base + offX + jx * 4096 + c);
    }
  };
  // 2*NI*NJ MFMAs on register operands; optionally the NP LDS-DMA pieces of tile `dma_kt` are issued in the gaps (one behind
  // every second MFMA: the MFMA pipe hides their issue cost, and the read phases stay pure LDS reads)
#if WF_GEMM_DMA_PHASE == 2
  // MFMA phase WITH the pieces of tile `kt_next` (compile-time: the caller branches ONCE per phase between this and the plain phase)
  auto mma_half_dma = [&](int kt_next) {
    if (!(WF_GEMM_ABLATE & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jx = 0; jx < NJ; ++jx) {
          acc[i][jx] = mfma32t<F16>(fw[ks][i], fx[ks][jx], acc[i][jx]);
          const int idx = (ks * NI + i) * NJ + jx;
          constexpr int RS = WF_GEMM_DMA_RSPLIT < NP ? WF_GEMM_DMA_RSPLIT : NP;
          if ((idx & 1) && (idx >> 1) < NP - RS && !(WF_GEMM_ABLATE & 1)) {
            piece_s(RS + (idx >> 1), kt_next);
            __builtin_amdgcn_sched_barrier(0);  // the piece stays in this gap
          }
        }
    __builtin_amdgcn_s_setprio(0);
  };
#endif
  auto mma_half = [&](int dma_kt) {
    if (WF_GEMM_ABLATE & 1) dma_kt = -1;
    if (!(WF_GEMM_ABLATE & 8)) __builtin_amdgcn_s_setprio(1);  // the MFMA phase outranks the co-resident wave's LDS phase at the issue arbiter
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jx = 0; jx < NJ; ++jx) {
          acc[i][jx] = mfma32t<F16>(fw[ks][i], fx[ks][jx], acc[i][jx]);
          const int idx = (ks * NI + i) * NJ + jx;
          if (dma_kt >= 0 && (idx & 1) && (idx >> 1) < NP) dma_piece(dma_kt, idx >> 1);
        }
    if (dma_kt >= 0) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  const int nk = a.K / PK;
#ifdef WF_GEMM_TIMING
  const unsigned long long tt0 = __builtin_readcyclecounter();
#endif
  dma(0);
  drain();
  bar();
#ifdef WF_GEMM_TIMING
  const unsigned long long tt1 = __builtin_readcyclecounter();
#endif
  auto pbar = [&]() {  // a phase hand-over barrier (lab: WF_GEMM_ABLATE & 2 drops them, keeping the K tile's buffer hand-over)
    if (!(WF_GEMM_ABLATE & 2)) bar();
  };
#if WF_GEMM_DMA_PHASE == 1
  // Round 4.  A wave in its MFMA phase has ~12 free issue cycles per MFMA; an LDS-DMA piece (M0 set-up, 64-bit address, the VMEM issue)
  // needs ~60, so every piece placed between two MFMAs cost ~42 cycles of matrix pipe: 9 pieces x 8 waves = 12.7 % of the K loop
  // (cycle-counted ablation, profiles/r4_f_gemm_ablate.md).  The partner wave on the same SIMD, meanwhile, sits in its READ phase: 14 LDS
  // reads, then the hand-over barrier.  Each wave therefore stages ITS pieces of tile kt+1 at the tail of its own first read phase of tile
  // kt -- group A in phase 1, group B in phase 2 -- where their issue time overlaps the other group's MFMA phase; the MFMA phases are
  // pure MFMA streams.  Hazards: buffer (kt+1)&1 was last read before barrier 4kt (both groups' R1 of tile kt-1), and every wave drains
  // its pieces before barrier 4kt+4, behind which tile kt+1 is first read.
  if (!groupB) {
    for (int kt = 0; kt < nk; ++kt) {
      read_half(kt, 0);
      if (kt + 1 < nk && !(WF_GEMM_ABLATE & 1)) dma(kt + 1);
      pbar();  // 4kt+1
      mma_half(-1);
      pbar();  // 4kt+2
      read_half(kt, 1);
      pbar();  // 4kt+3
      mma_half(-1);
      drain();
      bar();  // 4kt+4
    }
    bar();
  } else {
    bar();  // 1
    for (int kt = 0; kt < nk; ++kt) {
      read_half(kt, 0);
      if (kt + 1 < nk && !(WF_GEMM_ABLATE & 1)) dma(kt + 1);
      pbar();  // 4kt+2
      mma_half(-1);
      pbar();  // 4kt+3
      read_half(kt, 1);
      drain();
      bar();  // 4kt+4
      mma_half(-1);
      pbar();  // 4kt+5
    }
  }
#elif WF_GEMM_DMA_PHASE == 2
  auto rphase_pieces = [&](int kt_next) {  // the first WF_GEMM_DMA_RSPLIT pieces, behind the read phase's LDS reads
#pragma unroll
    for (int i = 0; i < (WF_GEMM_DMA_RSPLIT < NP ? WF_GEMM_DMA_RSPLIT : NP); ++i)
      if (!(WF_GEMM_ABLATE & 1)) piece_s(i, kt_next);
  };
  if (!groupB) {
    for (int kt = 0; kt < nk; ++kt) {
      read_half(kt, 0);
      rphase_pieces(kt + 1);
      pbar();  // 4kt+1
      mma_half_dma(kt + 1);  // the pieces of tile kt+1 ride in the MFMA gaps (its buffer is free since barrier 4kt)
      pbar();  // 4kt+2
      read_half(kt, 1);
      pbar();  // 4kt+3
      mma_half(-1);
      drain();
      bar();  // 4kt+4
    }
    bar();
  } else {
    if (nk > 1) dma(1);
    bar();  // 1
    for (int kt = 0; kt < nk; ++kt) {
      read_half(kt, 0);
      if (kt > 0) rphase_pieces(kt + 1);  // group B's share of tile kt+1 that its previous MFMA phase (in phase 1) left over; tile 1 came whole from the prologue
      pbar();  // 4kt+2
      mma_half(-1);
      pbar();  // 4kt+3
      read_half(kt, 1);
      drain();
      bar();  // 4kt+4
      mma_half_dma(kt + 2);
      pbar();  // 4kt+5
    }
    drain();  // the dummy pieces of the last two phases
  }
#else
  if (!groupB) {
    for (int kt = 0; kt < nk; ++kt) {
      read_half(kt, 0);
      pbar();  // 4kt+1
      mma_half(kt + 1 < nk ? kt + 1 : -1);  // DMA of tile kt+1 rides in the MFMA gaps (its buffer is free since barrier 4kt)
      pbar();  // 4kt+2
      read_half(kt, 1);
      pbar();  // 4kt+3
      mma_half(-1);
      drain();
      bar();  // 4kt+4
    }
    bar();
  } else {
    if (nk > 1) dma(1);
    bar();  // 1
    for (int kt = 0; kt < nk; ++kt) {
      read_half(kt, 0);
      pbar();  // 4kt+2
      mma_half(-1);
      pbar();  // 4kt+3
      read_half(kt, 1);
      drain();
      bar();  // 4kt+4
      mma_half(kt + 2 < nk ? kt + 2 : -1);
      pbar();  // 4kt+5
    }
  }
#endif

#ifdef WF_GEMM_TIMING
  const unsigned long long tt2 = __builtin_readcyclecounter();
#endif
  // ---- epilogue through LDS: row-contiguous global accesses ------------------------------------------------------------------
  // In the accumulator layout a lane owns one token row and quads of features, so a store instruction touches 32-64 different rows
  // (8 / 16 bytes each): 64 such instructions per lane made the epilogue ~20 k cycles per tile, 9 % of a K = 5120 GEMM.  Each wave
  // therefore transposes its tile through a private LDS region (the operand buffers are free behind the last barrier) in passes of
  // [NJ*32 tokens][128 B]  (64 bf16 features = two MFMA tiles, or 32 fp32 features = one; an odd last bf16 tile makes a 64-byte
  // pass) and reads / writes global memory in whole rows of a pass.  Rows are padded by 16 B (144-byte stride) so that neither the
  // column-wise writes nor the row-wise reads conflict.  No workgroup barrier: the region is wave-private.
  {
    constexpr int RS = 144;  // padded row stride in bytes
    constexpr int ROWS = NJ * 32;
    unsigned char* stg = smem + wid * G::STG;
    if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
#pragma unroll
      for (int i0 = 0; i0 < NI; i0 += 2) {
        const int nti = (NI - i0) >= 2 ? 2 : 1;  // feature tiles in this pass (compile-time after unrolling)
        // the pass's bias quads in one batch (one wait) -- per store group they were serialized L2 round trips; features >= N are never
        // stored, so their (clamped) bias value does not matter
        f32x4 bq[2][4];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int g = 0; g < 4; ++g) bq[ii][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              if (ii < nti) bq[ii][g] = *reinterpret_cast<const f32x4*>(a.bias + min(n0 + wf0 + (i0 + ii) * 32 + 8 * g + 4 * hi, a.N - 4));
        }
#pragma unroll
        for (int jx = 0; jx < NJ; ++jx)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            if (ii >= nti) continue;
            const int i = i0 + ii;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nl = ii * 32 + 8 * g + 4 * hi;  // feature within the pass
              const int n = n0 + wf0 + i0 * 32 + nl;
              float v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = (F16 ? acc[i][jx][4 * g + q] * a.acc_scale : acc[i][jx][4 * g + q]) + bq[ii][g][q];
              if constexpr (EPI == EPI_BF16_GELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_tanh(v[q]);
              }
              u32x2 pk;
              if constexpr (F16)
                pk = u32x2{pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3])};
              else
                pk = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
              *reinterpret_cast<u32x2*>(stg + (jx * 32 + l31) * RS + nl * 2) = pk;
            }
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // row phase: 8 (4) lanes x 16 B per token row, 8 (16) rows per instruction
        const int lpr = nti * 4;
        const int lrow = lane / lpr, lch = lane % lpr;
#pragma unroll
        for (int r8 = 0; r8 < ROWS * nti / 16; ++r8) {
          const int row = r8 * (64 / lpr) + lrow;
          const int m = m0 + wt0 + row;
          const int n = n0 + wf0 + i0 * 32 + lch * 8;
          const u32x4 val = *reinterpret_cast<const u32x4*>(stg + row * RS + lch * 16);
          if (m < a.M && n < a.N) {  // N % 4 == 0: a chunk of 8 features may straddle the edge
            uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (size_t)m * a.ldo + n;
            if (n + 8 <= a.N)
              *reinterpret_cast<u32x4*>(op) = val;
            else
              *reinterpret_cast<u32x2*>(op) = u32x2{val[0], val[1]};
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next pass overwrites the staging rows
      }
    } else {
      const int lrow = lane >> 3, lch = lane & 7;  // row phase: 8 rows x 8 chunks of 16 B per instruction
      // fp32 outputs: one pass of [NJ*32 tokens][32 features] f32 = 128 B per row for every feature tile
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        // the pass's four bias quads in one batch (one wait): tested and loaded per store group they were four serialized L2 round trips.
        // Features >= N are never stored, so their (clamped) bias value does not matter.
        f32x4 bq[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (a.bias) {
#pragma unroll
          for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(a.bias + min(n0 + wf0 + i * 32 + 8 * g + 4 * hi, a.N - 4));
        }
#pragma unroll
        for (int jx = 0; jx < NJ; ++jx)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = 8 * g + 4 * hi;
            const float sc = F16 ? a.acc_scale : 1.0f;  // (folds away in the bf16 instantiations)
            f32x4 v = {acc[i][jx][4 * g + 0] * sc + bq[g][0], acc[i][jx][4 * g + 1] * sc + bq[g][1], acc[i][jx][4 * g + 2] * sc + bq[g][2],
                       acc[i][jx][4 * g + 3] * sc + bq[g][3]};
            *reinterpret_cast<f32x4*>(stg + (jx * 32 + l31) * RS + nl * 4) = v;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // Read-modify-write epilogues: ALL the old values of a chunk of rows are loaded before the first store.  Written as one loop
        // (load, add, store per row) the compiler must keep every load behind the previous row's store -- it cannot know they do not
        // alias -- and each row waited a full HBM round trip: 40 serialized round trips per wave and tile.  The gate row (the same
        // features for every row of the pass) is loaded once per pass for the same reason.
        const int n = n0 + wf0 + i * 32 + lch * 4;
        f32x4 gg = {1.f, 1.f, 1.f, 1.f};
        if constexpr (EPI == EPI_RESID) {
          if (a.gate) gg = *reinterpret_cast<const f32x4*>(a.gate + min(n, a.N - 4));
        }
        constexpr int CH = 4;  // rows-of-8 per chunk: 4 x 16 B per lane in flight (8 spill: the accumulators of the later passes are still live)
#pragma unroll
        for (int c0 = 0; c0 < ROWS / 8; c0 += CH) {
          f32x4 oldv[CH];
          if constexpr (EPI != EPI_F32) {
#pragma unroll
            for (int r8 = 0; r8 < CH; ++r8) {  // unconditional, clamped: a guard per load would put each in its own block with its own wait
              const int m = min(m0 + wt0 + (c0 + r8) * 8 + lrow, a.M - 1);
              oldv[r8] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.out) + (size_t)m * a.ldo + min(n, a.N - 4));
            }
          }
#pragma unroll
          for (int r8 = 0; r8 < CH; ++r8) {
            const int row = (c0 + r8) * 8 + lrow;
            const int m = m0 + wt0 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * RS + lch * 16);
            if (m < a.M && n < a.N) {
              float* po = reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n;
              if constexpr (EPI == EPI_F32) {
                *reinterpret_cast<f32x4*>(po) = v;
              } else if constexpr (EPI == EPI_F32_ACC) {
                const f32x4 old = oldv[r8];
                *reinterpret_cast<f32x4*>(po) = f32x4{old[0] + v[0], old[1] + v[1], old[2] + v[2], old[3] + v[3]};
              } else {
                const f32x4 old = oldv[r8];
                *reinterpret_cast<f32x4*>(po) =
                    f32x4{old[0] + v[0] * gg[0], old[1] + v[1] * gg[1], old[2] + v[2] * gg[2], old[3] + v[3] * gg[3]};
              }
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next pass overwrites the staging rows
      }
    }
  }
#ifdef WF_GEMM_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the epilogue's stores have left the wave
  const unsigned long long tt3 = __builtin_readcyclecounter();
  if (lane == 0 && wid == 0) {
    atomicAdd(&g_gemm_cycles[8], tt1 - tt0);
    atomicAdd(&g_gemm_cycles[9], tt2 - tt1);
    atomicAdd(&g_gemm_cycles[10], tt3 - tt2);
    atomicAdd(&g_gemm_cycles[11], 1ull);
  }
  if (lane == 0 && wid == 4) atomicAdd(&g_gemm_cycles[12], tt2 - tt1);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Removed in round 5 (superseded kernels kept alive behind environment switches): `k_gemm_pp16`, the same ping-pong
// kernel on v_mfma_f32_16x16x32_bf16 (round 3, WF_GEMM_MFMA=16: +3 ... +5 % in isolated back-to-back launches, nothing in the model,
// profiles/r3_i_gemm_mfma16.md), and `k_gemm_w4`, the one-wave-per-SIMD 256 x 256 tile with all 256 accumulators in AGPRs (round 2,
// WF_GEMM_KERNEL=w4: a better K loop, the same 1200-1300 TFLOP/s, bench 0.234 vs 0.242).  Both are measured negatives documented in
// DESIGN.md section 4; their source is in the history (last present at 57ab8ba).
// ------------------------------------------------------------------------------------------------------------------

template <int EPI, int NI>
static void launch_pp(GemmArgs a, hipStream_t s) {
  using G = PPGeom<NI>;
  a.mt = ceil_div(a.M, PM);
  a.nt = ceil_div(a.N, G::PNT);
  const int nsuper = ((a.mt + 3) / 4) * ((a.nt + 3) / 4);
  const int grid = ((nsuper + 7) / 8) * 8 * 16;
  if (a.f16) {  // fp16 operands: the epilogues the VAE uses (the others are rejected by wf_gemm_f16)
    if constexpr (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_F32_ACC) hipLaunchKernelGGL((k_gemm_pp<EPI, NI, true>), dim3(grid, a.batch), dim3(PT), G::LDS, s, a);
  } else
    hipLaunchKernelGGL((k_gemm_pp<EPI, NI>), dim3(grid, a.batch), dim3(PT), G::LDS, s, a);
}

// 256- or 320-feature tiles: whichever leaves fewer idle workgroup slots in the last round of `n_cu` concurrent workgroups
// (ties go to the wider tile).  The result does not depend on the choice: every output element is the same sequence of MFMA
// accumulations over K in both.
static int pp_wide(int M, int N) {
  if (N % 320 != 0) return 0;
#ifdef WF_GEMM_LAB_TILE  // lab builds only: WF_GEMM_TILE=256|320 read per call, so that one process can launch both tile widths (tools/gemm_dephase.py, gemm_tiles.py)
  const char* e_ = getenv("WF_GEMM_TILE");
  const int force = e_ ? atoi(e_) : 0;
#else
  const int force = 0;
#endif
  if (force == 256) return 0;
  if (force == 320) return 1;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const long mt = ceil_div(M, PM);
  const long t256 = mt * ceil_div(N, 256), t320 = mt * (N / 320);
  const long r256 = (t256 + n_cu - 1) / n_cu, r320 = (t320 + n_cu - 1) / n_cu;
  // cost ~ rounds x tile width
  return r320 * 320 <= r256 * 256;
}

template <int EPI>
static void launch_pp_any(GemmArgs a, hipStream_t s) {
  if (pp_wide(a.M, a.N))
    launch_pp<EPI, 5>(a, s);
  else
    launch_pp<EPI, 2>(a, s);
}

}  // namespace

static int gemm_impl(const void* X, const void* W, const float* bias, void* out, const float* gate, int M, int N,
                     int K, int ldx, int ldw, int ldo, int epilogue, void* stream, int f16, float acc_scale = 1.0f) {
  WF_CHECK_ARG(acc_scale > 0.0f && acc_scale < 3.0e38f, "wf_gemm_f16: acc_scale must be a positive finite number (1 = none)");
  WF_CHECK_ARG(X && W && out, "wf_gemm_bf16: null pointer");
  WF_CHECK_ARG(!f16 || epilogue == EPI_BF16 || epilogue == EPI_F32 || epilogue == EPI_F32_ACC,
               "wf_gemm_f16: epilogue %d is not built for fp16 operands (0 = 16-bit out, 2 = f32, 4 = f32 accumulate)", epilogue);
  WF_CHECK_ARG(M > 0 && N > 0 && K > 0, "wf_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
  WF_CHECK_ARG(K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldw >= K && ldx >= K,
               "wf_gemm_bf16: K (%d), ldx (%d), ldw (%d) must be multiples of 8 with ld >= K", K, ldx, ldw);
  WF_CHECK_ARG(N % 4 == 0 && ldo % 4 == 0, "wf_gemm_bf16: N (%d) and ldo (%d) must be multiples of 4", N, ldo);
  WF_CHECK_ARG((((uintptr_t)X | (uintptr_t)W | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)gate) & 15) == 0,
               "wf_gemm_bf16: pointers must be 16-byte aligned");
  GemmArgs a;
  a.X = (const uint16_t*)X;
  a.W = (const uint16_t*)W;
  a.bias = bias;
  a.out = out;
  a.gate = gate;
  a.M = M;
  a.N = N;
  a.K = K;
  a.ldx = ldx;
  a.ldw = ldw;
  a.ldo = ldo;
  a.mt = ceil_div(M, BM);
  a.nt = ceil_div(N, BN);
  a.bsx = a.bsw = a.bso = 0;
  a.f16 = f16;
  a.batch = 1;
  a.acc_scale = acc_scale;
  const int nsuper = ((a.mt + 7) / 8) * ((a.nt + 7) / 8);
  const int grid = ((nsuper + 7) / 8) * 8 * 64;
  const size_t lds = 4 * TILE_BYTES;
  hipStream_t s = (hipStream_t)stream;
  // large problems with a whole number of 64-wide K tiles take the 256x256 ping-pong kernel
  // INVARIANT relied on by k_gemm_pp: its epilogues load bias / gate / old-value quads UNCONDITIONALLY at
  // clamped addresses `min(n, N - 4)` (a guard per load would put every load in its own basic block), which needs N >= 4 and N % 4 == 0.
  // `big` guarantees both (N >= 256, N % 4 == 0); a future relaxation of this gate must keep them.
  // ... and the saddr LDS-DMA of k_gemm_pp addresses a row's bytes by a 32-bit offset from X / W
  const bool big = K % PK == 0 && M >= 1024 && N >= 256 && N % 4 == 0 && (long)M * N >= (1L << 22) &&
                   (size_t)M * ldx * 2 < (1ull << 32) && (size_t)N * ldw * 2 < (1ull << 32);
  if (big) {
    switch (epilogue) {
      case EPI_BF16: launch_pp_any<EPI_BF16>(a, s); break;
      case EPI_BF16_GELU: launch_pp_any<EPI_BF16_GELU>(a, s); break;
      case EPI_F32: launch_pp_any<EPI_F32>(a, s); break;
      case EPI_RESID: launch_pp_any<EPI_RESID>(a, s); break;
      case EPI_F32_ACC: launch_pp_any<EPI_F32_ACC>(a, s); break;
      default: WF_CHECK_ARG(false, "wf_gemm_bf16: unknown epilogue %d", epilogue);
    }
    WF_LAUNCH_CHECK("wf_gemm_bf16");
    return WF_OK;
  }
  if (f16) {
    switch (epilogue) {
      case EPI_BF16: hipLaunchKernelGGL((k_gemm<EPI_BF16, true>), dim3(grid), dim3(NTHREADS), lds, s, a); break;
      case EPI_F32: hipLaunchKernelGGL((k_gemm<EPI_F32, true>), dim3(grid), dim3(NTHREADS), lds, s, a); break;
      default: hipLaunchKernelGGL((k_gemm<EPI_F32_ACC, true>), dim3(grid), dim3(NTHREADS), lds, s, a); break;
    }
    WF_LAUNCH_CHECK("wf_gemm_f16");
    return WF_OK;
  }
  switch (epilogue) {
    case EPI_BF16: hipLaunchKernelGGL(k_gemm<EPI_BF16>, dim3(grid), dim3(NTHREADS), lds, s, a); break;
    case EPI_BF16_GELU: hipLaunchKernelGGL(k_gemm<EPI_BF16_GELU>, dim3(grid), dim3(NTHREADS), lds, s, a); break;
    case EPI_F32: hipLaunchKernelGGL(k_gemm<EPI_F32>, dim3(grid), dim3(NTHREADS), lds, s, a); break;
    case EPI_RESID: hipLaunchKernelGGL(k_gemm<EPI_RESID>, dim3(grid), dim3(NTHREADS), lds, s, a); break;
    case EPI_F32_ACC: hipLaunchKernelGGL(k_gemm<EPI_F32_ACC>, dim3(grid), dim3(NTHREADS), lds, s, a); break;
    default: WF_CHECK_ARG(false, "wf_gemm_bf16: unknown epilogue %d", epilogue);
  }
  WF_LAUNCH_CHECK("wf_gemm_bf16");
  return WF_OK;
}

extern "C" int wf_gemm_bf16(const void* X, const void* W, const float* bias, void* out, const float* gate, int M, int N,
                            int K, int ldx, int ldw, int ldo, int epilogue, void* stream) {
  return gemm_impl(X, W, bias, out, gate, M, N, K, ldx, ldw, ldo, epilogue, stream, 0);
}

// The same GEMM on fp16 operands (X, W fp16; epilogue 0 writes fp16): v_mfma_f32_32x32x16_f16, fp32 accumulation.  Epilogues 0 / 2 / 4.
extern "C" int wf_gemm_f16(const void* X, const void* W, const float* bias, void* out, int M, int N, int K, int ldx, int ldw, int ldo,
                           int epilogue, float acc_scale, void* stream) {
  return gemm_impl(X, W, bias, out, nullptr, M, N, K, ldx, ldw, ldo, epilogue, stream, 1, acc_scale);
}

#ifdef WF_GEMM_TIMING
extern "C" int wf_debug_gemm_cycles(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gemm_cycles), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_cycles), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// `batch` independent small GEMMs in ONE launch (blockIdx.y = problem): out_b[M,N] = X_b[M,K] . W_b[N,K]^T, problem b at X + b*bsx,
// W + b*bsw, out + b*bso (elements).  The 128 x 128 register-staged kernel; epilogue EPI_BF16 or EPI_F32, no bias / gate.
static int gemm_batched_impl(const void* X, const void* W, void* out, int batch, int M, int N, int K, int ldx, int ldw, int ldo, int64_t bsx,
                             int64_t bsw, int64_t bso, int epilogue, void* stream, int f16) {
  WF_CHECK_ARG(X && W && out, "wf_gemm_bf16_batched: null pointer");
  WF_CHECK_ARG(batch > 0 && batch <= 65535 && M > 0 && N > 0 && K > 0, "wf_gemm_bf16_batched: empty problem batch=%d M=%d N=%d K=%d", batch, M, N, K);
  WF_CHECK_ARG(K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldw >= K && ldx >= K && bsx % 8 == 0 && bsw % 8 == 0,
               "wf_gemm_bf16_batched: K, ldx, ldw, bsx, bsw must be multiples of 8 with ld >= K");
  WF_CHECK_ARG(N % 4 == 0 && ldo % 4 == 0 && bso % 4 == 0, "wf_gemm_bf16_batched: N, ldo, bso must be multiples of 4");
  WF_CHECK_ARG((((uintptr_t)X | (uintptr_t)W | (uintptr_t)out) & 15) == 0, "wf_gemm_bf16_batched: pointers must be 16-byte aligned");
  WF_CHECK_ARG(epilogue == EPI_BF16 || epilogue == EPI_F32, "wf_gemm_bf16_batched: epilogue must be 0 (16-bit out) or 2 (f32)");
  GemmArgs a;
  a.X = (const uint16_t*)X;
  a.W = (const uint16_t*)W;
  a.bias = nullptr;
  a.out = out;
  a.gate = nullptr;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldo = ldo;
  a.mt = ceil_div(M, BM);
  a.nt = ceil_div(N, BN);
  a.bsx = bsx; a.bsw = bsw; a.bso = bso;
  a.f16 = f16;
  a.batch = batch;
  a.acc_scale = 1.0f;
  hipStream_t s = (hipStream_t)stream;
  // problems that are large TOGETHER (the VAE mid-block's P . V products: 6240 x 384 x 18 720 per frame, 21 frames) take the ping-pong kernel
  // with the batch index on gridDim.y; the gate is gemm_impl's with the batch counted in
  const bool big = K % PK == 0 && M >= 1024 && N >= 256 && N % 4 == 0 && (long)M * N * batch >= (1L << 22) &&
                   (size_t)M * ldx * 2 < (1ull << 32) && (size_t)N * ldw * 2 < (1ull << 32);
  if (big) {
    if (epilogue == EPI_BF16)
      launch_pp_any<EPI_BF16>(a, s);
    else
      launch_pp_any<EPI_F32>(a, s);
    WF_LAUNCH_CHECK("wf_gemm_bf16_batched");
    return WF_OK;
  }
  const int nsuper = ((a.mt + 7) / 8) * ((a.nt + 7) / 8);
  const int grid = ((nsuper + 7) / 8) * 8 * 64;
  const dim3 g(grid, batch), blk(NTHREADS);
  if (f16) {
    if (epilogue == EPI_BF16)
      hipLaunchKernelGGL((k_gemm<EPI_BF16, true>), g, blk, 4 * TILE_BYTES, s, a);
    else
      hipLaunchKernelGGL((k_gemm<EPI_F32, true>), g, blk, 4 * TILE_BYTES, s, a);
  } else {
    if (epilogue == EPI_BF16)
      hipLaunchKernelGGL(k_gemm<EPI_BF16>, g, blk, 4 * TILE_BYTES, s, a);
    else
      hipLaunchKernelGGL(k_gemm<EPI_F32>, g, blk, 4 * TILE_BYTES, s, a);
  }
  WF_LAUNCH_CHECK("wf_gemm_bf16_batched");
  return WF_OK;
}

extern "C" int wf_gemm_bf16_batched(const void* X, const void* W, void* out, int batch, int M, int N, int K, int ldx, int ldw, int ldo,
                                    int64_t bsx, int64_t bsw, int64_t bso, int epilogue, void* stream) {
  return gemm_batched_impl(X, W, out, batch, M, N, K, ldx, ldw, ldo, bsx, bsw, bso, epilogue, stream, 0);
}

// wf_gemm_bf16_batched on fp16 operands (epilogue 0 writes fp16): the per-frame P.V products of the VAE mid-block attention in its fp16
// three-term mode, all frames in one launch (each is 147 workgroups of the 128 x 128 kernel: alone it fills a fraction of the chip)
extern "C" int wf_gemm_f16_batched(const void* X, const void* W, void* out, int batch, int M, int N, int K, int ldx, int ldw, int ldo,
                                   int64_t bsx, int64_t bsw, int64_t bso, int epilogue, void* stream) {
  return gemm_batched_impl(X, W, out, batch, M, N, K, ldx, ldw, ldo, bsx, bsw, bso, epilogue, stream, 1);
}
