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
// One kernel, k_attn_w4: one wave per SIMD, 4 waves x 64 query rows, MFMA and softmax hand-placed in ONE instruction stream (description
// at the kernel).  Its formulation:
//   * "swapped" QK^T: S^T[key][q] = mfma(A = K tile, B = Q^T), so a lane owns ONE query column and 32 of the 64 scores of
//     a KV tile: the row max / row sum are in-lane reductions plus one exchange with lane^32 (v_permlane32_swap);
//   * the K rows are fed to the MFMA with index bits 2 and 3 swapped, which makes the scores a lane holds in registers
//     8m..8m+7 exactly the 8 consecutive keys the P^T B-operand of the P.V MFMA needs: P never leaves registers and
//     needs no cross-lane movement (cvt to bf16 only);
//   * O^T[d][q] += mfma(A = V^T tile, B = P^T): the lane ends up with quads of consecutive head-dim values of its query
//     row -> 8-byte stores;
//   * K and V^T tiles live in a ring of 32 KiB LDS buffers filled by LDS-DMA (global_load_lds, no staging VGPRs, no
//     ds_write pass); the XOR swizzle of the LDS image -- 16-byte chunk c of row r at c ^ (r & 15) (K, 256-B rows) resp.
//     c ^ ((r >> 1) & 7) (V^T, 128-B rows), conflict-free for the ds_read_b128 lane groups (SQ_LDS_BANK_CONFLICT = 0) -- is
//     applied to the per-lane SOURCE address because the DMA destination is lane-linear;
//   * QK^T and P.V MFMAs are interleaved: an MFMA that accumulates into the result of one issued < ~4 issue slots earlier stalls
//     (QK^T alone, two score accumulators: 850-1050 cycles per 16 MFMAs instead of 512);
//   * online softmax in fp32 with exp2 and the 1/sqrt(d)*log2(e) scale folded into one FMA; the rescale of O is deferred
//     until a row maximum grew by more than 2^8 (the same m is used for P and for the
//     row sum, so the result is exact for any reference m);
//   * XCD-aware grid: workgroup b runs on XCD b % 8; all query blocks of a head are given to one XCD so its 32 CUs share
//     that head's K / V^T stream through their L2.
#include "common.h"
#include "mfma.h"

#include <climits>
#include <cstdlib>
#include <type_traits>
#include <utility>

using namespace wf;

namespace {

constexpr int D = 128;
constexpr int QB = 256;  // query rows per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int K_TILE_BYTES = KB * D * 2;  // 16 KiB
constexpr int V_TILE_BYTES = D * KB * 2;  // 16 KiB

struct AttnArgs {
  const uint16_t* Q;
  const uint16_t* K;
  const uint16_t* Vt;
  uint16_t* O;
  int H, Lq, Lkp, kv_len, ldo;
  int seg_len;  // keys per K/V segment (= Lkp, or the per-rank shard length when K/V were all-gathered: [P][H][seg_len][128])
  // bytes from a head's first tile in one segment to the same head's first tile in the next one, in K and in V^T alike: H * seg_len * 256
  // for the dense [P][H][seg_len][128] form; larger when each source's K, V^T (and norm bounds) travel in ONE packed slot of an exchange
  // buffer (parallel.KVExchange: [P][K | V^T | bounds])
  size_t seg_stride;
  int n_qblk;
  float scale_log2;  // softmax_scale * log2(e)
  int accumulate;    // O += result (second cross-attention, model.py:227)
  int prio_mode;     // experiment: 1 = static s_setprio 1 for waves 4-7, 2 = for waves 0-3
  // split-KV (k_attn_w4 only): blockIdx.y = split s works on KV tiles [s * tiles_per_split, ...) and leaves un-normalised partials
  int nsplit, tiles_per_split;
  // KV-tile window of this launch (wf_attn_fwd_part: the segments that have ARRIVED so far): split s starts at absolute tile
  // t_begin0 + s * tiles_per_split, no tile >= t_end is touched, and the partials land in slot part0 + s.  Whole-sweep launches: 0, INT_MAX, 0.
  int t_begin0, t_end, part0;
  // optional SECOND window of a part launch (the keys on the far side of a hole -- the rank's own segment, walked earlier without a wait):
  // splits s >= n_first work on tiles [t_begin2 + (s - n_first) * tiles_per_split2, ...) below t_end2.  Launches with one window: n_first = INT_MAX
  int n_first, t_begin2, t_end2, tiles_per_split2;
  // PART launches (k_attn_w4_part) read the four fields above differently: the launch walks the JOINED sequence [t_begin0, t_end) ++
  // [t_begin2, t_end2) in splits of tiles_per_split; n_first = tiles in front of the hole (INT_MAX: no hole), tiles_per_split2 = the joined
  // length.  merge_n > 0 (a single-split LAST launch of a sweep): fold the partial slots 0 .. merge_n - 1 into this launch's own result
  // and write the normalised bf16 rows to O instead of leaving a partial.
  int merge_n = 0;
  float* o_part;   // [nsplit][Lq][H*128] f32
  float* ml_part;  // [nsplit][H][Lq][2] f32: reference max m (raw score units), row sum l
  // block-sparse attention (KIND 3): per (head, 256-row query group) a list of 128-key blocks to visit, entry = block * 4 + flags,
  // flags bit 0 / 1 = the block is selected by the first / second 128-row query block of the group
  const int* bsa_list;  // [H][n_qblk][bsa_max]
  const int* bsa_cnt;   // [H][n_qblk]
  int bsa_max;
  int bsa_shift;        // flag bits per entry = query blocks per workgroup: 2 (128-token blocks, 2 tiles per entry) or 4 (64-token, 1 tile)
  // KIND 4: per-head max over the rows of |k|^2 and of |q|^2 (wf_head_max_norm2; Q pre-scaled), kmax_n / qmax_n vectors of H floats each
  // (one per rank shard of an all-gathered K), or NULL
  const float* kmax2;
  const float* qmax2;
  int kmax_n, qmax_n;
  int kmax_stride;  // floats between two kmax2 vectors (H when they are contiguous; the packed exchange slots carry one vector each)
  // KIND 5 (two-context cross-attention, model.py:202-229): the key sequence is [n1 tiles of context 1 (kv_len1 valid keys) | the tiles of
  // context 2 (kv_len valid keys)]; the softmax is taken over each context separately and the two results are summed
  int kv_len1, n1;
  // test hook (wf_attn_debug_body_counter): device uint32[2], [0] += workgroups that ran the tracked body, [1] += the un-tracked one; NULL = off
  unsigned int* dbg_body;
};

#ifndef WF_ATTN_DMA_PLACE
#define WF_ATTN_DMA_PLACE 0  // lab only (tools/attn_lab.py): 9 = no LDS-DMA pieces in the tile loop (wrong results; what the pieces cost)
#endif
#ifndef WF_ATTN_ABLATE
#define WF_ATTN_ABLATE 0  // lab only (wrong results): 1 no softmax VALU, 2 no drain / barrier, 4 no LDS fragment reads in the loop
#endif
#ifdef WF_ATTN_TIMING
// Per-phase cycle accounting (debug builds only: python -m worldforge_amd.build with WF_EXTRA_HIPCC_FLAGS=-DWF_ATTN_TIMING).
__device__ unsigned long long g_attn_cycles[32];  // k_attn_w4: B*8 + (gap >> 3) for the 8-gap groups of even (B=0) / odd tiles, 16 commit, 17 drain, 18 barrier, 19 tiles
#endif

constexpr int MAX_MERGE = 11;  // earlier partial slots a merging part launch can fold in (MAX_PARTS - 1)
__device__ __forceinline__ float ws_at(const float (&w)[MAX_MERGE], int s) {  // w[s] for a run-time (wave-uniform) s without spilling the array
  float r = w[0];
#pragma unroll
  for (int i = 1; i < MAX_MERGE; ++i) r = s == i ? w[i] : r;
  return r;
}

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// KIND (k_attn_w4): 0 / 1 only name the launch for profilers (0 = long K/V: self-attention in its in-kernel-scale form, 1 = short K/V
// <= 1024 keys: stand-alone cross-attention launches); 3 = block-sparse attention of the LongCat refine pass
// (block_sparse_attention/bsa_interface.py:538-560 + flash_attn_bsa_varlen_mask.py:236-285: queries and keys in 3D-block order, a workgroup
// walks the UNION of its query blocks' selected key-block lists and masks the blocks a query block did not select with -inf; the running
// max starts finite so that a leading run of masked blocks cannot give inf - inf); 4 = pre-scaled Q (the DiT self-attention); 5 = the
// fused two-context cross-attention.  (The first design, an 8-wave two-per-SIMD ping-pong kernel `k_attn`, was kept behind
// WF_ATTN_KERNEL=w8 through round 2 and removed in round 3: 1100-1160 TFLOP/s against this kernel's 1260-1430)

// ======================================================================================================================================
// k_attn_w4: one wave per SIMD (4 waves = 256 query rows per workgroup, 64 rows = two 32-row q-blocks a / b per wave, the whole
// 512-entry register file).  With two waves per SIMD (k_attn) every VALU instruction of the softmax wave costs the MFMA wave issue
// time (measured: 32 MFMAs take 1300-1500 cycles instead of 1024 beside a partner's softmax).  Here ONE instruction stream carries
// both: per KV tile two regions, each 32 MFMAs of one q-block with the ~160 VALU instructions of the OTHER q-block's online softmax
// placed in the MFMA issue shadows (an MFMA occupies the matrix pipe for 32 cycles = ~8 issue slots):
//     R1(t):  softmax_a(t)   beside   P.V_b(t-1) , QK^T_b(t)        (their MFMAs interleaved: see phase_m of k_attn)
//     R2(t):  softmax_b(t)   beside   P.V_a(t)   , QK^T_a(t+1)
// so every dependency (scores -> softmax -> P -> P.V) has a full region of slack, the score registers are single-buffered
// (S_a is consumed while S_b is produced and vice versa), and there is ONE workgroup barrier per tile (the LDS ring).
// LDS-DMA: 4 waves x (4 K + 4 V^T) pieces per tile, issued at the head of R1(t) for tile t+2, drained before the tile's barrier.
// ======================================================================================================================================
constexpr int NT4 = 256;

// one rounding each, whatever -ffp-contract says (HIP's __fmul_rn / __fadd_rn are plain operators and may still be fused into an fma)
__device__ __forceinline__ float mul_rn(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float add_rn(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// NOMAX (KIND 4 only, chosen per workgroup by the kernel below): no running-max tracking after tile 0 -- the reference max of every row
// stays the first tile's for the whole sweep.
// PART (round 6): the instantiation wf_attn_fwd_part launches.  (i) Its tile window may have a HOLE -- tiles [t_begin0, t_end) and
// [t_begin2, t_end2) are walked by the SAME workgroups as one sequence (the hole is the rank's own segment of the exchange buffer, walked
// earlier without a wait): one partial slot and one prologue / epilogue per peer chunk where round 5 used a separate set of workgroups and
// a separate slot per side of the hole.  (ii) The LAST part launch of a sweep merges the earlier launches' partial slots in its epilogue
// and writes the normalised bf16 result (merge_n earlier slots; the arithmetic and its order are k_attn_merge's) instead of storing its own
// partial for a separate merge pass.  Everything PART adds is `if constexpr`: the whole-sweep instantiations are unchanged instruction for
// instruction.
template <int KIND, bool NOMAX, bool PART = false>
__device__ __forceinline__ void attn_w4_body(const AttnArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BUF_BYTES = K_TILE_BYTES + V_TILE_BYTES;
  constexpr int NBUF = 5;  // ring of 5 tile buffers = the whole 160 KiB: lets the workgroup barrier run every SECOND tile
  // The LDS of this kernel is ONE ring, nothing is overlaid on it (no epilogue staging: O leaves through registers); what the
  // hand-placed LDS-DMA pieces and the barrier-every-second-tile schedule rely on (guarded, not just commented):
  static_assert(NBUF * BUF_BYTES == 160 * 1024, "the ring is the whole LDS of a CU and the launch asks for exactly that (attn_launch: lds_w4)");
  static_assert((NT4 / 64) * 4 * 1024 == K_TILE_BYTES && (NT4 / 64) * 4 * 1024 == V_TILE_BYTES,
                "4 waves x (4 K + 4 V^T) pieces of 1 KiB (64 lanes x 16 B) cover a K tile and a V^T tile exactly: no piece may spill into the next ring slot");
  static_assert(K_TILE_BYTES % 1024 == 0 && BUF_BYTES % 16 == 0, "piece granularity / ds_read_b128 alignment of the slots");
  // tiles are staged 3 ahead of the one being consumed and the workgroup barrier runs once per TWO tiles: while a wave still reads tile t
  // (and its K(t+1) head), the fastest wave may already stage tile t + 3 + 1 -- the slot being written must not be one of the slots
  // t .. t + 1 still being read: needs at least 2 (being read) + 3 (staged ahead) slots
  static_assert(NBUF >= 2 + 3, "ring too short for the 3-ahead staging with one barrier per two tiles");

  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int hslot = j / a.n_qblk;
  const int head = hslot * 8 + xcd;
  const int qblk = j % a.n_qblk;
  if (head >= a.H) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int q_row[2];
  bf16x8 qf[2][8];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    q_row[x] = qblk * QB + wid * 64 + x * 32 + l31;
    const uint16_t* qp = a.Q + ((size_t)head * a.Lq + min(q_row[x], a.Lq - 1)) * D;
#pragma unroll
    for (int st = 0; st < 8; ++st) qf[x][st] = as_bf16x8(*reinterpret_cast<const u32x4*>(qp + 16 * st + 8 * hi));
  }

  const int wu = __builtin_amdgcn_readfirstlane(wid);
  int ksrc[4], vsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wu * 4 + i;
    const int kr = 4 * piece + (lane >> 4), ks = lane & 15;
    ksrc[i] = kr * D + ((ks ^ (kr & 15)) << 3);
    const int vr = 8 * piece + (lane >> 3), vs = lane & 7;
    vsrc[i] = vr * KB + ((vs ^ ((vr >> 1) & 7)) << 3);
  }
  const int tiles_per_seg = a.seg_len / KB;
  int krow_off[2], krow_sw[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    int kr = kb * 32 + swap23(l31);
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

  // Register homes are chosen by hand through inline-asm MFMAs (hipcc selects ONE accumulator form per function, and with the
  // AGPR form it shuttles every accumulator the VALU touches through v_accvgpr copies: ~600 per tile):
  //   O (128 regs)      AGPR  accumulated by "+a" MFMAs, touched by the VALU only in the rare rescale block and the epilogue
  //   Q (64 regs)       AGPR  MFMA B operand
  //   S (2 x 64 regs)   VGPR  "+v" MFMAs; double-buffered: the scores of tile t+1 are produced while those of tile t are consumed
  //   P (32 regs)       VGPR  written by the softmax just in time, MFMA B operand
  f32x16 o[2][4];
  f32x16 sb[2][2][2];  // [buffer][q-block][key block]
  bf16x8 pf[2][4];
  // KIND 3 (block-sparse): the running max starts finite so that leading masked blocks cannot give inf - inf
  float m_run[2] = {KIND == 3 ? -1e30f : -INFINITY, KIND == 3 ? -1e30f : -INFINITY}, l_run[2] = {0.f, 0.f}, mcq[2] = {0.f, 0.f};
  // KIND 4 ("prescaled"): Q arrives multiplied by softmax_scale * log2(e) (wf_rmsnorm_heads out_scale: applied before the producer's
  // single bf16 rounding, so it costs no precision), i.e. the MFMA result is already in the exp2 domain, and the score accumulators are
  // INITIALISED with -m (the C operand of each block's first MFMA is minit = 16 registers holding -m of the lane's query): the
  // registers hold t = s - m directly and the 64 v_fma of softmax stage A disappear from every tile.  m moves only in the rare rescale.
  constexpr bool PS = KIND == 4;
  f32x16 minit[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][db][r] = 0.f;
      asm volatile("" : "+a"(o[x][db]));  // O enters the loop in AGPRs
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[x][i] = as_bf16x8(u32x4{0u, 0u, 0u, 0u});
  }
  const float c = PS ? 1.0f : a.scale_log2;
  const int ntiles_all = (KIND == 5 ? a.n1 : 0) + (a.kv_len + KB - 1) / KB;
  const int split = a.part0 + blockIdx.y;                                // partial slot of this split
  // PART: the launch's tiles are the JOINED sequence [t_begin0, t_end) ++ [t_begin2, t_end2) (n_first = its length up to the hole, INT_MAX
  // without one; tiles_per_split2 = its whole length); split s takes tiles_per_split of them from relative tile s * tiles_per_split on.
  // hole_at: relative tile (within this split) in front of which the sweep hops over the hole; hole_tiles: what it skips.
  int t_begin, ntiles_, hole_at = INT_MAX, hole_tiles = 0;
  if constexpr (PART) {
    const int rel0 = (int)blockIdx.y * a.tiles_per_split;
    const bool after = rel0 >= a.n_first;
    t_begin = after ? a.t_begin2 + (rel0 - a.n_first) : a.t_begin0 + rel0;
    ntiles_ = min(a.tiles_per_split, a.tiles_per_split2 - rel0);
    if (!after && rel0 + ntiles_ > a.n_first) {
      hole_at = a.n_first - rel0;
      hole_tiles = a.t_begin2 - a.t_end;
    }
  } else {
    t_begin = a.t_begin0 + (int)blockIdx.y * a.tiles_per_split;  // first KV tile of this split (absolute)
    ntiles_ = min(a.tiles_per_split, min(ntiles_all, a.t_end) - t_begin);  // tiles of this split (>= 1 by construction of the grid)
  }
  // block-sparse variant: per-workgroup list of PHYSICAL key blocks (entry = block * 2^g + flags, g = bsa_shift query blocks per
  // workgroup = 2 waves each for 128-token blocks / 1 wave each for 64-token blocks), tpe tiles per entry
  const int* bsa = nullptr;
  const int tpe = (KIND == 3 && a.bsa_shift == 2) ? 2 : 1;
  const int mine = a.bsa_shift == 2 ? (wu >> 1) : wu;  // this wave's query block within the workgroup
  if constexpr (KIND == 3) {
    bsa = a.bsa_list + ((size_t)head * a.n_qblk + qblk) * a.bsa_max;
    ntiles_ = tpe * a.bsa_cnt[head * a.n_qblk + qblk];
    if (ntiles_ == 0) {  // empty selection for every query block of the group: zeros (flash_attn_bsa_varlen_mask.py:242-244)
#pragma unroll
      for (int x = 0; x < 2; ++x)
        if (q_row[x] < a.Lq) {
          uint16_t* op = a.O + (size_t)q_row[x] * a.ldo + head * D;
          for (int d = hi * 64; d < hi * 64 + 64; d += 8) *reinterpret_cast<u32x4*>(op + d) = u32x4{0u, 0u, 0u, 0u};
        }
      return;
    }
  }
  const int ntiles = ntiles_;
  const bool ragged = KIND != 3 && KIND != 5 && (a.kv_len & (KB - 1)) != 0;
  const int rag_t = ragged ? ntiles_all - 2 - t_begin - hole_tiles : -1;  // the tile in which the scores of the ragged last tile are produced (one scalar compare per tile; relative to the split: minus the tiles of a hole hopped over)
  const bool rag1 = KIND == 5 && (a.kv_len1 & (KB - 1)) != 0, rag2 = KIND == 5 && (a.kv_len & (KB - 1)) != 0;  // ragged last tile of either context

  // Element offset / LDS slot of the tile being staged (wave-uniform), advanced one tile at a time: a handful of scalar
  // instructions, no division.  Past the last tile the pieces are still issued (no branch in the MFMA stream): they re-read the last
  // tile into the next ring slots, whose tiles (ntiles-5 ... ntiles-3) are dead by then.
  const size_t tile_bytes = (size_t)(KB * D * 2);
  const size_t seg_jump = a.seg_stride - (size_t)tiles_per_seg * tile_bytes;  // to the next K/V segment of the same head (all-gathered shards)
  const size_t hole_bytes = (size_t)(hole_tiles / tiles_per_seg) * a.seg_stride;  // PART: the whole segments a hop skips
  const int seg0 = t_begin / tiles_per_seg, in0 = t_begin - seg0 * tiles_per_seg;
  size_t st_off = (size_t)seg0 * a.seg_stride + ((size_t)head * tiles_per_seg + in0) * tile_bytes;  // byte offset of the staged tile in K and in V^T
  int st_tile = 0, st_left = tiles_per_seg - in0;  // tile number (in the split), tiles left in its segment
  uint32_t st_base = wu * 4096;                    // LDS byte offset of this wave's pieces in the ring slot of the staged tile
  int e_stage = 0, e_mask = 0;                     // KIND 3: list entry of the tile being staged / of the tile whose scores come next
  // KIND 3: the list entries are read ONE SELECTION AHEAD (e_pref: the un-waited load of the entry the next stage_next will need) and the
  // entry of the tile whose scores come next is remembered from when that tile was staged (e_q: entries of tiles t+1, t+2, t+3) -- a
  // load that is used right away is waited for with vmcnt(0), which also waits for every LDS-DMA piece in flight: two such waits per
  // tile were ~a quarter of the block-sparse loop (tools/isa_audit.py: "loads waited singly")
  int e_pref = 0, e_q[3] = {0, 0, 0};
  const int n_entries = KIND == 3 ? ntiles / tpe : 1;
  auto entry_index = [&](int tile) { return min(tpe == 2 ? (tile >> 1) : tile, n_entries - 1); };
  if constexpr (KIND == 3) {
    e_stage = __builtin_amdgcn_readfirstlane(bsa[0]);
    st_off = (size_t)(e_stage >> a.bsa_shift) * tpe * tile_bytes;
    e_q[0] = e_q[1] = e_q[2] = e_stage;
    e_pref = bsa[entry_index(1)];
  }
  auto stage_next = [&]() {  // select tile st_tile + 1.  Branch-free on purpose: the same bookkeeping with a rarely-taken branch for the
    ++st_tile;               // segment / end-of-split cases (3 scalar instructions + s_cbranch in the common case) measured 22 cycles per tile SLOWER
    const bool live = st_tile < ntiles;
    if constexpr (KIND == 3) {
      if (live) {
        e_stage = __builtin_amdgcn_readfirstlane(e_pref);  // = bsa[entry_index(st_tile)], loaded one selection ago
        st_off = ((size_t)(e_stage >> a.bsa_shift) * tpe + (st_tile & (tpe - 1))) * tile_bytes;
      }
      e_pref = bsa[entry_index(st_tile + 1)];  // unconditional (clamped): consumed by the next selection
      e_q[0] = e_q[1];
      e_q[1] = e_q[2];
      e_q[2] = e_stage;
      st_base = st_base + BUF_BYTES >= (uint32_t)(NBUF * BUF_BYTES) ? st_base + BUF_BYTES - NBUF * BUF_BYTES : st_base + BUF_BYTES;
      return;
    }
    const bool wrap = --st_left == 0;
    if constexpr (PART) {  // the hole starts and ends on segment boundaries: the hop is a longer segment jump (scalar, branch-free like the rest)
      const bool hop = st_tile == hole_at;
      st_off += live ? (wrap ? tile_bytes + seg_jump + (hop ? hole_bytes : 0) : tile_bytes) : 0;
    } else {
      st_off += live ? (wrap ? tile_bytes + seg_jump : tile_bytes) : 0;
    }
    st_left = wrap ? tiles_per_seg : st_left;
    st_base = st_base + BUF_BYTES >= (uint32_t)(NBUF * BUF_BYTES) ? st_base + BUF_BYTES - NBUF * BUF_BYTES : st_base + BUF_BYTES;
  };
  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_offset(smem));
  uint32_t kbyte[4], vbyte[4];  // per-lane byte offsets of the pieces inside a tile (constant): uniform base + 32-bit VGPR offset is
#pragma unroll                   // the saddr form of global_load_lds -- no 64-bit address arithmetic on the VALU per piece
  for (int i = 0; i < 4; ++i) {
    kbyte[i] = (uint32_t)ksrc[i] * 2u;
    vbyte[i] = (uint32_t)vsrc[i] * 2u;
  }
  auto stage_piece = [&](int i) {  // piece i (0..7) of this wave's share of the selected tile: 4 K then 4 V^T pieces
    const unsigned char* kb_ = reinterpret_cast<const unsigned char*>(a.K) + st_off;
    const unsigned char* vb_ = reinterpret_cast<const unsigned char*>(a.Vt) + st_off;
    if (i < 4)
      glds16_saddr(kb_, kbyte[i], smem_base + st_base + i * 1024);
    else
      glds16_saddr(vb_, vbyte[i - 4], smem_base + st_base + K_TILE_BYTES + (i - 4) * 1024);
  };
  auto stage_piece_c = [&](auto IC) {  // the same for a compile-time piece number (main loop): M0 = slot base + immediate in one s_add_u32
    constexpr int i = decltype(IC)::value;
    const unsigned char* kb_ = reinterpret_cast<const unsigned char*>(a.K) + st_off;
    const unsigned char* vb_ = reinterpret_cast<const unsigned char*>(a.Vt) + st_off;
    if constexpr (i < 4)
      glds16_saddr_imm<i * 1024>(kb_, kbyte[i], smem_base + st_base);
    else
      glds16_saddr_imm<K_TILE_BYTES + (i - 4) * 1024>(vb_, vbyte[i - 4], smem_base + st_base);
  };
  auto kread = [&](const unsigned char* sKb, int i) {
    const int kb = i & 1, st = i >> 1;
    return *reinterpret_cast<const u32x4*>(sKb + krow_off[kb] + (((2 * st + hi) ^ krow_sw[kb]) << 4));
  };
  auto vread = [&](const unsigned char* sVb, int i) {
    const int db = i & 3, m4 = i >> 2;
    return *reinterpret_cast<const u32x4*>(sVb + vrow_off[db] + (((2 * m4 + hi) ^ vrow_sw[db]) << 4));
  };
  auto mfma_s = [&](f32x16& acc, u32x4 kf, const bf16x8& q, bool first, const f32x16* init = nullptr) {
    if (first && init)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(kf), "a"(q), "v"(*init));
    else if (first)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(kf), "a"(q));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(kf), "a"(q));
  };
  auto mfma_o = [&](f32x16& acc, u32x4 vf, const bf16x8& pfr) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(vf), "v"(pfr));
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto drain_dma = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // commit of the running max of q-block X given the row max `mloc` of the next tile's scores (already exchanged between the lane
  // halves); rare path: some row max grew -> rescale O_X, l_X.  O_X lives in AGPRs and must never be a VALU operand in the hot
  // loop: the copies to VGPRs and back are pinned inside this block by the empty asm statements.
  auto set_minit = [&](auto XC) {
    constexpr int X = decltype(XC)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[X][r] = -m_run[X];
    asm volatile("" : "+v"(minit[X]));
  };
  // KIND 4: `rel` = max(row max of the NEW scores relative to m_run, 0), already exchanged between the lane halves; `nb` = the score
  // buffer that holds those new scores (they were computed against the old m and are re-based here)
  auto commit_ps = [&](auto XC, auto NBC, float rel) {
    constexpr int X = decltype(XC)::value, NB = decltype(NBC)::value;
    if (__any(rel > 8.0f)) {
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
      const float alpha = __builtin_amdgcn_exp2f(-rel);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        f32x16 tmp = o[X][db];
        asm volatile("" : "+v"(tmp));
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[r] *= alpha;
        asm volatile("" : "+v"(tmp));
        o[X][db] = tmp;
        asm volatile("" : "+a"(o[X][db]));
      }
      l_run[X] *= alpha;
      m_run[X] += rel;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sb[NB][X][kb][r] -= rel;
      set_minit(XC);
    }
  };
  auto commit = [&](auto XC, float m_new) {
    constexpr int X = decltype(XC)::value;
    // Deferred rescale: the reference max m_run only has to bound the scores well enough for exp2 not to overflow; it is the SAME m
    // for P and for the row sum, so the result is exact for any m.  O_X / l_X are rescaled only when some row max of the wave grew
    // by more than 2^8 in the exp2 domain (p <= 256 otherwise) -- almost never after the first tiles, instead of ~1 tile in 4.
    if (__any(c * (m_new - m_run[X]) > 8.0f)) {
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");  // asm MFMA write -> v_accvgpr_read: 18 wait states
      const float alpha = __builtin_amdgcn_exp2f(c * (m_run[X] - m_new));
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        f32x16 tmp = o[X][db];
        asm volatile("" : "+v"(tmp));
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[r] *= alpha;
        asm volatile("" : "+v"(tmp));
        o[X][db] = tmp;
        asm volatile("" : "+a"(o[X][db]));
      }
      l_run[X] *= alpha;
      m_run[X] = m_new;
    }
    mcq[X] = c * m_run[X];
  };
  // row max of the 32 scores per lane of q-block X in buffer B, exchanged between the lane halves (un-sliced form: prologue only)
  auto rowmax_now = [&](auto BC, auto XC) {
    constexpr int B = decltype(BC)::value, X = decltype(XC)::value;
    float m = sb[B][X][0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sb[B][X][kb][r]);
    const unsigned mu = __float_as_uint(m);
    auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
    return fmaxf(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])), m_run[X]);
  };
  auto mask_ragged = [&](auto BC, int t) {  // scores of keys >= kv_len -> -inf (last tile only)
    constexpr int B = decltype(BC)::value;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = (t_begin + t + (PART && t >= hole_at ? hole_tiles : 0)) * KB + 32 * kb + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= a.kv_len) sb[B][x][kb][r] = -INFINITY;
        }
  };

  auto mask_tail = [&](auto BC, int tile_in_ctx, int len) {  // KIND 5: scores of keys >= len of a context's last tile -> -inf
    constexpr int B = decltype(BC)::value;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = tile_in_ctx * KB + 32 * kb + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= len) sb[B][x][kb][r] = -INFINITY;
        }
  };

  auto mask_unselected = [&](auto BC, int entry) {  // KIND 3: the key block of these scores is not in this wave's query block's list
    constexpr int B = decltype(BC)::value;
    if (!((entry >> mine) & 1)) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sb[B][x][kb][r] = -INFINITY;
    }
  };

  // ---- one KV tile t (scores of tile t in buffer B, produced earlier): 64 MFMA gaps -------------------------------------------------
  //   gaps  0..31  QK^T(t+1) -> buffer 1-B, q-blocks a / b alternating, ONE K fragment read per two MFMAs
  //   gaps 32..63  P.V(t), q-blocks alternating, one V^T fragment read per two MFMAs
  //   VALU, <= ~5 issue slots per gap:
  //     gaps 0..51   the 64 (q-block, element) pairs of softmax(t) in P-fragment order (f0a f0b f1a f1b f2a ...), each through a
  //                  3-stage pipeline one gap apart (t = c s - c m | p = exp2 t | row sum, bf16 pack) so that nothing waits on the
  //                  instruction before it; P fragment f is complete before gap 32 + 8 f, where its first MFMA sits
  //     gaps 52..63  row max of the NEW scores (v_max3 chains); the commit (+ rare rescale) follows the last gap
  //   LDS-DMA: the 8 pieces of tile t+3 in gaps 2, 6, ..., 30;  fragment rings are refilled across the phase / tile seams.
#ifndef WF_ATTN_PF4
#define WF_ATTN_PF4 4  // fragment ring depth of the tile loop (lab: 2 / 8; must divide 16)
#endif
  constexpr int PF4 = WF_ATTN_PF4;
  u32x4 ring[PF4];
#if defined(WF_ATTN_TIMING) && WF_ATTN_TIMING != 2
  unsigned long long tacc[20] = {};
#define W4MARK(k) do { const unsigned long long n__ = __builtin_readcyclecounter(); tacc[k] += n__ - tlast; tlast = n__; } while (0)
#else
#define W4MARK(k) do { } while (0)
#endif
  auto tile = [&](auto BC, int t, int slot0) {  // slot0 = ring slot of tile t
    constexpr int B = decltype(BC)::value;
    // past the last tile the next ring slots hold copies of the last tile (stage_next stops advancing, the pieces are still issued):
    // the scores computed from them are finite and nobody reads them -- no end-of-sequence selects in the tile's scalar preamble
    const int slot1 = slot0 + 1 == NBUF ? 0 : slot0 + 1;
    const int slot2 = slot1 + 1 == NBUF ? 0 : slot1 + 1;
    const unsigned char* sK1 = smem + slot1 * BUF_BYTES;                 // K(t+1): read now
    const unsigned char* sV0 = smem + slot0 * BUF_BYTES + K_TILE_BYTES;  // V^T(t)
    const unsigned char* sK2 = smem + slot2 * BUF_BYTES;                 // K(t+2): head of the next tile's ring
    float tq_[64], pq[64];
    float ls[2][4];
    uint32_t pk[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    float mx[2][4];
    float mn[2] = {0.f, 0.f};
#if defined(WF_ATTN_TIMING) && WF_ATTN_TIMING != 2
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_sched_barrier(0);
    for_const<64>([&](auto GC) {
      auto& sb_ = sb;
      auto& pf_ = pf;
      auto& o_ = o;
      auto& ring_ = ring;
      auto& tq__ = tq_;
      auto& pq_ = pq;
      auto& ls_ = ls;
      auto& pk_ = pk;
      auto& mx_ = mx;
      auto& mcq_ = mcq;
      constexpr int g = decltype(GC)::value;
      constexpr int q = g & 1;
      if constexpr (g < 32) {
        constexpr int i = g >> 1;  // K fragment i: key block i & 1, k-step i >> 1
        mfma_s(sb_[1 - B][q][i & 1], ring_[i % PF4], qf[q][i >> 1], i < 2, PS ? &minit[q] : nullptr);
        __builtin_amdgcn_sched_barrier(0);  // the gap's VALU must not be hoisted above its MFMA
        if constexpr (q == 1 && !(WF_ATTN_ABLATE & 4)) {  // fragment i consumed by both q-blocks: refill its slot
          if constexpr (i + PF4 < 16)
            ring_[i % PF4] = kread(sK1, i + PF4);
          else
            ring_[i % PF4] = vread(sV0, i + PF4 - 16);
        }
      } else {
        constexpr int i = (g - 32) >> 1;  // V^T fragment i: output block i & 3, key step i >> 2
        mfma_o(o_[q][i & 3], ring_[i % PF4], pf_[q][i >> 2]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q == 1 && !(WF_ATTN_ABLATE & 4)) {
          if constexpr (i + PF4 < 16)
            ring_[i % PF4] = vread(sV0, i + PF4);
          else if constexpr (B == 0)  // an odd tile's successor was staged after the last barrier: its ring is filled behind the next one
            ring_[i % PF4] = kread(sK2, i + PF4 - 16);
        }
      }
      // ---- softmax(t) slices: pair k = 0..63 -> fragment f = k >> 4, q-block (k >> 3) & 1, element e = 8 f + (k & 7).
      // Order inside a gap: A (reads only scores), C, B -- so that neither the first VALU after an MFMA nor any VALU directly after a
      // v_exp consumes that v_exp's result (each would cost a wait state).
      for_const<64>([&](auto KC) {
        (void)&sb_, (void)&pf_, (void)&tq__, (void)&pq_, (void)&ls_, (void)&pk_, (void)&mcq_;
        constexpr int k = decltype(KC)::value;
        constexpr int f = k >> 4, x = (k >> 3) & 1, e = 8 * f + (k & 7);
        if constexpr (!(WF_ATTN_ABLATE & 1) && !PS && (50 * k) / 64 == g) {  // stage A (KIND 4: the score register already holds s - m)
          tq__[k] = c * sb_[B][x][e >> 4][e & 15] - mcq_[x];
          asm volatile("" : "+v"(tq__[k]));
        }
      });
      for_const<64>([&](auto KC) {
        (void)&sb_, (void)&pf_, (void)&tq__, (void)&pq_, (void)&ls_, (void)&pk_, (void)&mcq_;
        constexpr int k = decltype(KC)::value;
        constexpr int f = k >> 4, x = (k >> 3) & 1, e = 8 * f + (k & 7);
        if constexpr (!(WF_ATTN_ABLATE & 1) && (50 * k) / 64 + 2 == g) {  // stage C
          if constexpr (e < 4) {
            ls_[x][e & 3] = pq_[k];  // first element of each partial row sum
          } else {
            ls_[x][e & 3] += pq_[k];
            asm volatile("" : "+v"(ls_[x][e & 3]));
          }
          if constexpr ((e & 1) != 0) {
            pk_[x][(e >> 1) & 3] = pack_bf16x2(pq_[k - 1], pq_[k]);
            asm volatile("" : "+v"(pk_[x][(e >> 1) & 3]));
            if constexpr ((e & 7) == 7) {
              u32x4 v4 = {pk_[x][0], pk_[x][1], pk_[x][2], pk_[x][3]};
              pf_[x][f] = as_bf16x8(v4);
            }
          }
        }
      });
      for_const<64>([&](auto KC) {
        (void)&sb_, (void)&pf_, (void)&tq__, (void)&pq_, (void)&ls_, (void)&pk_, (void)&mcq_;
        constexpr int k = decltype(KC)::value;
        if constexpr (!(WF_ATTN_ABLATE & 1) && (50 * k) / 64 + 1 == g) {  // stage B
          if constexpr (PS) {
            constexpr int f = k >> 4, x = (k >> 3) & 1, e = 8 * f + (k & 7);
            pq_[k] = __builtin_amdgcn_exp2f(sb_[B][x][e >> 4][e & 15]);
          } else {
            pq_[k] = __builtin_amdgcn_exp2f(tq__[k]);
          }
          asm volatile("" : "+v"(pq_[k]));
        }
      });
      // ---- row max of the new scores: 2 q-blocks x 4 chains x 4 v_max3 steps in gaps 52..59 (4 steps per gap); tree, lane-half
      // exchange and running max in gaps 60..63, so that only the (rare) rescale decision is left after the last MFMA ----
      if constexpr (!NOMAX && g >= 52 && g < 60) {
        for_const<4>([&](auto JC) {
          (void)&sb_, (void)&mx_;
          constexpr int k = 4 * (g - 52) + decltype(JC)::value;  // 0..31
          constexpr int x = k >> 4, ch = k & 3, st = (k >> 2) & 3;
          constexpr int kb = ch >> 1, r0 = (ch & 1) + 4 * st;
          if constexpr (st == 0) {
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx_[x][ch]) : "v"(sb_[1 - B][x][kb][r0]), "v"(sb_[1 - B][x][kb][r0 + 2]));
          } else {
            asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx_[x][ch]) : "v"(sb_[1 - B][x][kb][r0]), "v"(sb_[1 - B][x][kb][r0 + 2]));
          }
        });
      }
      if constexpr (!NOMAX && (g == 60 || g == 61)) {
        constexpr int x = g - 60;
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(mn[x]) : "v"(mx_[x][0]), "v"(mx_[x][1]), "v"(mx_[x][2]));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(mn[x]) : "v"(mx_[x][3]));
      }
      if constexpr (!NOMAX && (g == 62 || g == 63)) {
        constexpr int x = g - 62;
        const unsigned mu = __float_as_uint(mn[x]);
        auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
        if constexpr (PS) {
          const float zero = 0.0f;
          asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(mn[x]) : "v"(__uint_as_float(sw[0])), "v"(__uint_as_float(sw[1])), "v"(zero));
        } else {
          asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(mn[x]) : "v"(__uint_as_float(sw[0])), "v"(__uint_as_float(sw[1])), "v"(m_run[x]));
        }
      }
      if constexpr (!(WF_ATTN_ABLATE & 1) && (g == 54 || g == 56)) {  // the row sums of this tile are complete (last pack at gap 51)
        constexpr int x = (g - 54) >> 1;
        // one v_add_f32 each: left to itself the compiler SLP-packs the fold into v_pk_add_f32 (+ an s_nop), dearer beside MFMAs
        l_run[x] = add_rn(l_run[x], add_rn(add_rn(ls_[x][0], ls_[x][1]), add_rn(ls_[x][2], ls_[x][3])));
        asm volatile("" : "+v"(l_run[x]));
      }
#ifdef WF_ATTN_TIMING
      if constexpr ((g & 7) == 7) W4MARK(B * 8 + (g >> 3));
#endif
      if constexpr (g == 0) {
        stage_next();  // tile t + 3 (or trash): scalar bookkeeping, in the shadow of the first MFMA
        if constexpr (KIND == 3) e_mask = e_q[0];  // the entry of tile t + 1 (its scores are produced in this tile and masked at gap 51)
      }
#if WF_ATTN_DMA_PLACE == 0
      if constexpr (g >= 2 && g < 32 && (g & 3) == 2) stage_piece_c(std::integral_constant<int, ((g - 2) >> 2)>{});
#elif WF_ATTN_DMA_PLACE == 9  // ablation (results wrong): no pieces in the loop
#endif
      if constexpr (g == 51) {
        // the new scores were written by asm MFMAs (last one at gap 31): XDL write -> VALU read hazard is long covered; the ragged
        // mask of the last tile must be in place before its row max
        // (every rare block of the tile loop is marked unlikely: its common case must be the fall-through, see the loop below)
        if (__builtin_expect(t == rag_t, 0)) mask_ragged(std::integral_constant<int, 1 - B>{}, t + 1);
        if constexpr (KIND == 3) mask_unselected(std::integral_constant<int, 1 - B>{}, e_mask);
        if constexpr (KIND == 5) {
          if (__builtin_expect(rag1 && t + 1 == a.n1 - 1, 0)) mask_tail(std::integral_constant<int, 1 - B>{}, t + 1, a.kv_len1);
          if (__builtin_expect(rag2 && t + 1 == ntiles - 1, 0)) mask_tail(std::integral_constant<int, 1 - B>{}, t + 1 - a.n1, a.kv_len);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // one test for both q-blocks in the hot path; the (rare) commits re-test per q-block
    if constexpr (NOMAX) {
    } else if constexpr (PS) {
      if (__builtin_expect(t + 1 < ntiles && __any(fmaxf(mn[0], mn[1]) > 8.0f), 0)) {
        commit_ps(std::integral_constant<int, 0>{}, std::integral_constant<int, 1 - B>{}, mn[0]);
        commit_ps(std::integral_constant<int, 1>{}, std::integral_constant<int, 1 - B>{}, mn[1]);
      }
    } else if (__builtin_expect(t + 1 < ntiles && __any(c * fmaxf(mn[0] - m_run[0], mn[1] - m_run[1]) > 8.0f), 0)) {
      commit(std::integral_constant<int, 0>{}, mn[0]);
      commit(std::integral_constant<int, 1>{}, mn[1]);
    }
    W4MARK(16);
    // One barrier per TWO tiles (after the odd ones).  WAR: the pieces of tile t+3 overwrite the slot of tile t-2, whose last reader
    // ran before the barrier that ended tile t-1 or t-2.  RAW: tile T is staged during tile T-3 and first read for the ring of tile
    // T-1, i.e. at the tail of tile T-2 -- behind the barrier of tile T-3 when that is odd; when T-3 is even, tile T-2 is odd and
    // fills the ring after its own barrier instead (every wave drains its own pieces in front of each barrier).
    if constexpr (B == 1) {
      if constexpr (!(WF_ATTN_ABLATE & 2)) drain_dma();
      W4MARK(17);
      if constexpr (!(WF_ATTN_ABLATE & 2)) bar();
      W4MARK(18);
#pragma unroll
      for (int i = 0; i < ((WF_ATTN_ABLATE & 4) ? 0 : PF4); ++i) ring[i] = kread(sK2, i);  // K(t+2) was staged during tile t-1: readable only now
    }
  };
  // KIND 5: the seam between the two contexts, run after the last tile of context 1 (its P.V is complete; the scores of context 2's first
  // tile are already in buffer NB).  Context 1's result is normalised and rounded to bf16 exactly as a stand-alone launch would store it
  // (64 packed registers instead of a round trip through HBM), then O, l and the reference max start afresh -- m from the new scores
  // alone, as the prologue of a stand-alone launch takes it: the fused result is bit-identical to "context 1, then context 2 with accumulate".
  uint32_t stash[2][4][8];
  auto seam = [&](auto NBC) {
    constexpr int NB = decltype(NBC)::value;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");  // asm MFMA write -> v_accvgpr_read
    for_const<2>([&](auto XC) {
      constexpr int X = decltype(XC)::value;
      float l = l_run[X];
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        f32x16 tmp = o[X][db];
        asm volatile("" : "+v"(tmp));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          stash[X][db][2 * g] = pack_bf16x2(mul_rn(tmp[4 * g + 0], inv), mul_rn(tmp[4 * g + 1], inv));
          stash[X][db][2 * g + 1] = pack_bf16x2(mul_rn(tmp[4 * g + 2], inv), mul_rn(tmp[4 * g + 3], inv));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[r] = 0.f;
        asm volatile("" : "+v"(tmp));
        o[X][db] = tmp;
        asm volatile("" : "+a"(o[X][db]));
      }
      l_run[X] = 0.f;
      float m = sb[NB][X][0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sb[NB][X][kb][r]);
      const unsigned mu = __float_as_uint(m);
      auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      m_run[X] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      mcq[X] = c * m_run[X];
    });
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  // prologue: tiles 0, 1, 2 staged by everyone; scores of tile 0 (un-overlapped), their max committed; ring <- head of K(1)
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece(i);
  stage_next();
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece(i);
  stage_next();
#pragma unroll
  for (int i = 0; i < 8; ++i) stage_piece(i);
  drain_dma();
  bar();
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const u32x4 kf = kread(smem, i);
      mfma_s(sb[0][0][i & 1], kf, qf[0][i >> 1], i < 2);
      mfma_s(sb[0][1][i & 1], kf, qf[1][i >> 1], i < 2);
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
    if (ragged && t_begin == ntiles_all - 1) mask_ragged(B0{}, 0);
    if constexpr (KIND == 5) {
      if (rag1 && a.n1 == 1) mask_tail(B0{}, 0, a.kv_len1);
    }
    if constexpr (KIND == 3) mask_unselected(B0{}, __builtin_amdgcn_readfirstlane(bsa[0]));
    const float m0 = rowmax_now(B0{}, std::integral_constant<int, 0>{});
    const float m1 = rowmax_now(B0{}, std::integral_constant<int, 1>{});
    if constexpr (PS) {  // O = l = 0: nothing to rescale; the scores of tile 0 were accumulated from 0 and are re-based to the first max
      m_run[0] = m0;
      m_run[1] = m1;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sb[0][x][kb][r] -= m_run[x];
      set_minit(std::integral_constant<int, 0>{});
      set_minit(std::integral_constant<int, 1>{});
    } else {
      commit(std::integral_constant<int, 0>{}, m0);
      commit(std::integral_constant<int, 1>{}, m1);
    }
    const unsigned char* sK1 = smem + (ntiles > 1 ? 1 : 0) * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < PF4; ++i) ring[i] = kread(sK1, i);
  }
#if defined(WF_ATTN_TIMING) && WF_ATTN_TIMING == 2  // light mode: two s_memtime per workgroup, [20] = cycles of the whole tile loop, [19] = tiles
  const unsigned long long t_loop0 = __builtin_readcyclecounter();
#endif
  {
    int slot = 0;
    int t = 0;
    // a taken branch costs a lone wave ~130 cycles of instruction fetch (lab: the ragged-tile test alone was 5 % of the loop while its
    // common case was the taken direction): the hot body's back edge is taken once per FOUR tiles
    if constexpr (NOMAX) {
      for (; t + 3 < ntiles; t += 4) {
        tile(B0{}, t, slot);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        tile(B1{}, t + 1, slot);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        tile(B0{}, t + 2, slot);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        tile(B1{}, t + 3, slot);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
      }
    }
    for (; t < ntiles; t += 2) {
      tile(B0{}, t, slot);
      if constexpr (KIND == 5) {
        if (__builtin_expect(t == a.n1 - 1, 0)) seam(B1{});
      }
      slot = slot + 1 == NBUF ? 0 : slot + 1;
      if (t + 1 < ntiles) {
        tile(B1{}, t + 1, slot);
        if constexpr (KIND == 5) {
          if (__builtin_expect(t + 1 == a.n1 - 1, 0)) seam(B0{});
        }
      }
      slot = slot + 1 == NBUF ? 0 : slot + 1;
    }
  }
#if defined(WF_ATTN_TIMING) && WF_ATTN_TIMING == 2
  {
    const unsigned long long t_loop1 = __builtin_readcyclecounter();
    if (lane == 0 && wid == 0) {
      atomicAdd(&g_attn_cycles[20], t_loop1 - t_loop0);
      atomicAdd(&g_attn_cycles[19], (unsigned long long)ntiles);
    }
  }
#elif defined(WF_ATTN_TIMING)
  if (lane == 0 && wid == 0) {
    for (int i = 0; i < 19; ++i) atomicAdd(&g_attn_cycles[i], tacc[i]);
    atomicAdd(&g_attn_cycles[19], (unsigned long long)ntiles);
  }
#endif
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");

#pragma unroll
  for (int x = 0; x < 2; ++x) {
    float l = l_run[x];
    l += __shfl_xor(l, 32, 64);
    float inv = 1.0f / l;
    if constexpr (KIND == 3) inv = l > 0.f ? inv : 0.f;  // empty selection of this query block: zeros
    bool merged = false;
    if constexpr (PART) {
      if (a.merge_n > 0) {
        // the flash combine of k_attn_merge with this launch's own (m, l, O) still in registers: M = max over the slots' reference maxima
        // and the own one, every term weighted by 2^(m_s - M) (pre-scaled Q: exp2 domain), then the normalised store below
        merged = true;
        const size_t qr = (size_t)min(q_row[x], a.Lq - 1);
        float ms[MAX_MERGE], ws[MAX_MERGE];
        float M = m_run[x];
#pragma unroll
        for (int s = 0; s < MAX_MERGE; ++s) {
          ms[s] = s < a.merge_n ? a.ml_part[(((size_t)s * a.H + head) * a.Lq + qr) * 2] : -INFINITY;
          M = fmaxf(M, ms[s]);
        }
        const float w_own = __builtin_amdgcn_exp2f(m_run[x] - M);
        float lt = 0.f;
#pragma unroll
        for (int s = 0; s < MAX_MERGE; ++s) {
          ws[s] = s < a.merge_n ? __builtin_amdgcn_exp2f(ms[s] - M) : 0.f;
          if (s < a.merge_n) lt += ws[s] * a.ml_part[(((size_t)s * a.H + head) * a.Lq + qr) * 2 + 1];
        }
        l = lt + w_own * l;
        inv = 1.0f / l;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          f32x16 ov = o[x][db];
          asm volatile("" : "+v"(ov));
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < a.merge_n; ++s) {
              const f32x4 pv = *reinterpret_cast<const f32x4*>(a.o_part + ((size_t)s * a.Lq + qr) * (size_t)(a.H * D) + head * D + db * 32 + 8 * g + 4 * hi);
#pragma unroll
              for (int k = 0; k < 4; ++k) acc[k] += ws_at(ws, s) * pv[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) ov[4 * g + k] = acc[k] + w_own * ov[4 * g + k];
          }
          o[x][db] = ov;
        }
      }
    }
    if (a.nsplit > 1 && !merged) {  // un-normalised partial result of this KV split: O (f32), reference max, row sum -> k_attn_merge
      if (q_row[x] < a.Lq) {
        float* op = a.o_part + ((size_t)split * a.Lq + q_row[x]) * (size_t)(a.H * D) + head * D;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          f32x16 ov = o[x][db];
          asm volatile("" : "+v"(ov));
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 q4 = {ov[4 * g + 0], ov[4 * g + 1], ov[4 * g + 2], ov[4 * g + 3]};
            *reinterpret_cast<f32x4*>(op + db * 32 + 8 * g + 4 * hi) = q4;
          }
        }
        if (hi == 0) {
          float* mp = a.ml_part + (((size_t)split * a.H + head) * a.Lq + q_row[x]) * 2;
          mp[0] = m_run[x];
          mp[1] = l;
        }
      }
      continue;
    }
    // bf16 store as 8 x dwordx4 per lane and q-block instead of 16 x dwordx2 (a workgroup's store tail is issue-bound): a lane holds
    // d = 8 g + 4 hi + (0..3) of each 32-wide block; one v_permlane32_swap per dword hands the lower half-wave the partner's group of the
    // even g and the upper half-wave the partner's group of the odd g, so every lane owns 8 consecutive d = 16 j + 8 hi + (0..7).  Same
    // values, same roundings: only the storing lane changes.
    const bool live = q_row[x] < a.Lq;  // the two half-waves hold the same query rows
    uint16_t* op = a.O + (size_t)q_row[x] * a.ldo + head * D;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      f32x16 ov = o[x][db];
      asm volatile("" : "+v"(ov));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint32_t pk[2][2];  // [g & 1][dword]
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * j + gg;
          // the normalisation and the accumulation are two roundings (never contracted into one fma), so that the fused two-context
          // kernel (KIND 5) and the two-launch form (accumulate) produce the same bits
          float v0 = mul_rn(ov[4 * g + 0], inv), v1 = mul_rn(ov[4 * g + 1], inv), v2 = mul_rn(ov[4 * g + 2], inv),
                v3 = mul_rn(ov[4 * g + 3], inv);
          if constexpr (KIND == 5) {  // + context 1's stored (bf16) result
            const uint32_t o0 = stash[x][db][2 * g], o1 = stash[x][db][2 * g + 1];
            v0 = add_rn(v0, __uint_as_float(o0 << 16));
            v1 = add_rn(v1, __uint_as_float(o0 & 0xffff0000u));
            v2 = add_rn(v2, __uint_as_float(o1 << 16));
            v3 = add_rn(v3, __uint_as_float(o1 & 0xffff0000u));
          } else if (a.accumulate) {
            u32x2 old = {0u, 0u};
            if (live) old = *reinterpret_cast<const u32x2*>(op + db * 32 + 8 * g + 4 * hi);
            v0 = add_rn(v0, __uint_as_float(old[0] << 16));
            v1 = add_rn(v1, __uint_as_float(old[0] & 0xffff0000u));
            v2 = add_rn(v2, __uint_as_float(old[1] << 16));
            v3 = add_rn(v3, __uint_as_float(old[1] & 0xffff0000u));
          }
          pk[gg][0] = pack_bf16x2(v0, v1);
          pk[gg][1] = pack_bf16x2(v2, v3);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        if (live) {
          u32x4 st = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(op + db * 32 + 16 * j + 8 * hi) = st;
        }
      }
    }
  }
}

// KIND 4 with kmax2 / qmax2: Cauchy-Schwarz bounds every score of the head by B = max|q| max|k| (Q is pre-scaled: exp2 units), and the
// reference max a row takes from its first tile is >= -B, so s - m <= 2 B for every later score.  With 2 B <= 100 nothing can overflow
// exp2(s - m) or the fp32 row sums (P <= 2^100, l <= 2^117) and, the SAME m being used for P and l, the result is exact for that m: the
// workgroup runs the body WITHOUT running-max tracking (38 VALU + the rescale test per tile).  Otherwise (no bounds given, large norms,
// NaNs) it runs the tracked body.  The choice is made once, before anything else, and is uniform over the workgroup.
__device__ __forceinline__ bool attn_untracked_ok(const AttnArgs& a) {
  bool fast = false;
  if (a.kmax2 && a.qmax2) {
    const int b = blockIdx.x;
    const int head = ((b >> 3) / a.n_qblk) * 8 + (b & 7);
    if (head < a.H) {
      float kn2 = 0.f, qn2 = 0.f;
      for (int i = 0; i < a.kmax_n; ++i) kn2 = fmaxf(kn2, a.kmax2[(size_t)i * a.kmax_stride + head]);
      for (int i = 0; i < a.qmax_n; ++i) qn2 = fmaxf(qn2, a.qmax2[i * a.H + head]);
      fast = qn2 * kn2 <= 2500.0f;  // B <= 50.  Non-finite rows: k_head_max_norm2 reports +inf for a row holding a NaN or an inf, and
                                    // inf * x is inf (or NaN for x = 0), for which `<=` is false -> the tracked body
    }
  }
  if (a.dbg_body && threadIdx.x == 0) atomicAdd(a.dbg_body + (fast ? 1 : 0), 1u);
  return fast;
}

template <int KIND>
__global__ __launch_bounds__(NT4, 1) void k_attn_w4(AttnArgs a) {
  if constexpr (KIND == 4) {
    const bool fast = attn_untracked_ok(a);
    if (__builtin_amdgcn_readfirstlane((int)fast))
      attn_w4_body<4, true>(a);
    else
      attn_w4_body<4, false>(a);
  } else {
    attn_w4_body<KIND, false>(a);
  }
}

// The part launches of an own-first sweep (wf_attn_fwd_part): KIND 4 with the joined tile windows and the merging epilogue (PART).
__global__ __launch_bounds__(NT4, 1) void k_attn_w4_part(AttnArgs a) {
  const bool fast = attn_untracked_ok(a);
  if (__builtin_amdgcn_readfirstlane((int)fast))
    attn_w4_body<4, true, true>(a);
  else
    attn_w4_body<4, false, true>(a);
}

// Merge of the KV splits of k_attn_w4:  O = sum_s O_s 2^(c (m_s - M)) / sum_s l_s 2^(c (m_s - M)),  M = max_s m_s  (exact flash combine).
// One thread per (row, head, 4 consecutive head-dim values).
__global__ void k_attn_merge(AttnArgs a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)a.Lq * a.H * (D / 4);
  if (i >= n) return;
  const int d4 = (int)(i % (D / 4));
  const int h = (int)((i / (D / 4)) % a.H);
  const size_t q = i / ((size_t)(D / 4) * a.H);
  float M = -INFINITY;
  for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, a.ml_part[(((size_t)s * a.H + h) * a.Lq + q) * 2]);
  float l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.nsplit; ++s) {
    const float* ml = a.ml_part + (((size_t)s * a.H + h) * a.Lq + q) * 2;
    const float w = __builtin_amdgcn_exp2f(a.scale_log2 * (ml[0] - M));
    l += w * ml[1];
    const f32x4 ov = *reinterpret_cast<const f32x4*>(a.o_part + ((size_t)s * a.Lq + q) * (size_t)(a.H * D) + h * D + 4 * d4);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += w * ov[k];
  }
  const float inv = 1.0f / l;
  uint16_t* op = a.O + q * a.ldo + h * D + 4 * d4;
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = acc[k] * inv;
  if (a.accumulate) {
    const u32x2 old = *reinterpret_cast<const u32x2*>(op);
    v[0] += __uint_as_float(old[0] << 16);
    v[1] += __uint_as_float(old[0] & 0xffff0000u);
    v[2] += __uint_as_float(old[1] << 16);
    v[3] += __uint_as_float(old[1] & 0xffff0000u);
  }
  u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  *reinterpret_cast<u32x2*>(op) = pk;
}

}  // namespace

static unsigned int* g_dbg_body = nullptr;  // wf_attn_debug_body_counter

// part_index >= 0: a PART launch (wf_attn_fwd_part) -- KV tiles [part_t0, part_t1) only, un-normalised partials into slot part_index of the
// `nsplit`-slot workspace, no merge (wf_attn_merge follows once every slot is filled)
struct PartWindow {  // a part launch (wf_attn_fwd_part): KV tiles [t0, t1) (+ optionally, behind a hole, [t0b, t1b)) walked as ONE sequence in
  int index = -1, t0 = 0, t1 = 0, inner = 1, t0b = 0, t1b = 0;  // `inner` splits; merge: fold slots 0 .. index - 1 in and write O
  bool merge = false;
};
constexpr int MAX_PARTS = 12;

static int attn_launch(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int kv_len, int seg_len, int ldo,
                       float softmax_scale, int accumulate, int nsplit, void* workspace, const float* kmax2, int kmax_n, const float* qmax2,
                       int qmax_n, void* stream, const char* who, size_t seg_stride = 0, int kmax_stride = 0, PartWindow pw = PartWindow()) {
  const int part_index = pw.index, part_t0 = pw.t0, part_t1 = pw.t1, part_inner = pw.inner;
  WF_CHECK_ARG(Q && K && Vt && O, "%s: null pointer", who);
  WF_CHECK_ARG((!kmax2 || (kmax_n >= 1 && kmax_n <= 64)) && (!qmax2 || (qmax_n >= 1 && qmax_n <= 64)), "%s: kmax_n / qmax_n must be 1..64", who);
  WF_CHECK_ARG(H > 0 && Lq > 0 && kv_len > 0, "%s: empty problem", who);
  WF_CHECK_ARG(Lkp % KB == 0 && kv_len <= Lkp, "%s: Lkp (%d) must be a multiple of 64 and >= kv_len (%d)", who, Lkp, kv_len);
  WF_CHECK_ARG(seg_len > 0 && seg_len % KB == 0 && Lkp % seg_len == 0, "%s: seg_len (%d) must be a multiple of 64 dividing Lkp", who,
               seg_len);
  const size_t dense_stride = (size_t)H * seg_len * D * 2;
  WF_CHECK_ARG(seg_stride == 0 || (seg_stride >= dense_stride && seg_stride % 16 == 0),
               "%s: seg_stride_bytes (%zu) must be 0 (dense) or a multiple of 16 >= H * seg_len * 256 = %zu", who, seg_stride, dense_stride);
  WF_CHECK_ARG(kmax_stride == 0 || kmax_stride >= H, "%s: kmax_stride (%d) must be 0 (contiguous) or >= H", who, kmax_stride);
  WF_CHECK_ARG(ldo % 8 == 0 && ldo >= H * D, "%s: ldo %d must be a multiple of 8 (16-byte stores) and >= H * 128", who, ldo);
  WF_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O) & 15) == 0, "%s: 16-byte alignment", who);
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
  a.seg_stride = seg_stride ? seg_stride : dense_stride;
  a.kmax_stride = kmax_stride ? kmax_stride : H;
  a.ldo = ldo;
  a.n_qblk = ceil_div(Lq, QB);
  const bool prescaled = softmax_scale == 0.0f;  // Q already carries softmax_scale * log2(e) (wf_rmsnorm_heads out_scale)
  a.scale_log2 = prescaled ? 1.0f : softmax_scale * 1.4426950408889634f;
  a.accumulate = accumulate;
  a.prio_mode = 0;
  const int ntiles = ceil_div(kv_len, KB);
  a.nsplit = 1;
  a.tiles_per_split = ntiles;
  a.t_begin0 = 0;
  a.t_end = INT_MAX;
  a.part0 = 0;
  a.n_first = INT_MAX;
  a.t_begin2 = a.t_end2 = a.tiles_per_split2 = 0;
  a.o_part = nullptr;
  a.ml_part = nullptr;
  a.bsa_list = nullptr;
  a.bsa_cnt = nullptr;
  a.bsa_max = 0;
  a.bsa_shift = 2;
  a.kmax2 = kmax2;
  a.qmax2 = qmax2;
  a.kmax_n = kmax2 ? kmax_n : 0;
  a.qmax_n = qmax2 ? qmax_n : 0;
  a.dbg_body = g_dbg_body;
  a.kv_len1 = 0;
  a.n1 = 0;
  const int hslots = (H + 7) / 8;
  const int grid = hslots * a.n_qblk * 8;
  const size_t lds_w4 = 5 * (K_TILE_BYTES + V_TILE_BYTES);  // ring of 5 (160 KiB: the whole LDS of a CU)
  int grid_y = 1;
  if (part_index >= 0) {
    WF_CHECK_ARG(softmax_scale == 0.0f, "%s: part launches are built for the pre-scaled-Q form only (softmax_scale = 0)", who);
    WF_CHECK_ARG(part_t0 >= 0 && part_t0 < part_t1 && part_t0 < ntiles, "%s: empty tile window [%d, %d) of %d tiles", who, part_t0, part_t1, ntiles);
    const int t1 = part_t1 < ntiles ? part_t1 : ntiles;
    int joined = t1 - part_t0, t1b = 0;
    a.n_first = INT_MAX;
    if (pw.t1b > pw.t0b) {  // second window: the tiles beyond a hole (the own segment of the exchange buffer, walked by an earlier launch)
      const int tps_seg = seg_len / KB;
      WF_CHECK_ARG(pw.t0b >= t1 && pw.t0b < ntiles, "%s: second tile window [%d, %d) must lie behind the first [%d, %d) and inside the %d tiles", who,
                   pw.t0b, pw.t1b, part_t0, t1, ntiles);
      WF_CHECK_ARG(t1 == part_t1 && t1 % tps_seg == 0 && pw.t0b % tps_seg == 0 && pw.t0b > t1,
                   "%s: the hole [%d, %d) between the two windows must be whole segments of %d tiles", who, t1, pw.t0b, tps_seg);
      t1b = pw.t1b < ntiles ? pw.t1b : ntiles;
      a.n_first = t1 - part_t0;
      a.t_begin2 = pw.t0b;
      a.t_end2 = t1b;
      joined += t1b - pw.t0b;
    }
    // inner splits of the joined sequence (blockIdx.y): whole rounds of workgroups for short query shards, as wf_attn_fwd_split does for the
    // whole sweep
    const int tps = ceil_div(joined, part_inner < 1 ? 1 : part_inner);
    grid_y = ceil_div(joined, tps);  // splits that actually get tiles
    a.tiles_per_split2 = joined;
    if (pw.merge) {
      WF_CHECK_ARG(grid_y == 1 && part_index >= 1 && part_index <= MAX_MERGE && part_index == nsplit - 1,
                   "%s: a merging part launch is the LAST of its sweep and has one split (slot %d of %d, %d split(s))", who, part_index, nsplit, grid_y);
      a.merge_n = part_index;
    }
    WF_CHECK_ARG(nsplit >= 2 && nsplit <= MAX_PARTS && part_index + grid_y <= nsplit, "%s: slots %d..%d of %d (2..%d)", who, part_index,
                 part_index + grid_y - 1, nsplit, MAX_PARTS);
    WF_CHECK_ARG(workspace && (((uintptr_t)workspace) & 15) == 0, "%s: needs a 16-byte aligned workspace", who);
    a.nsplit = nsplit;  // > 1: the kernel leaves un-normalised partials
    a.t_begin0 = part_t0;
    a.t_end = t1;
    a.tiles_per_split = tps;
    a.part0 = part_index;
    a.o_part = (float*)workspace;
    a.ml_part = a.o_part + (size_t)nsplit * Lq * H * D;
  } else if (nsplit > 1) {
    WF_CHECK_ARG(workspace && (((uintptr_t)workspace) & 15) == 0, "%s: nsplit > 1 needs a 16-byte aligned workspace", who);
    int tps = ceil_div(ntiles, nsplit);
    const int ns = ceil_div(ntiles, tps);  // splits that actually get tiles
    a.nsplit = ns;
    a.tiles_per_split = tps;
    a.o_part = (float*)workspace;
    a.ml_part = a.o_part + (size_t)ns * Lq * H * D;
    grid_y = ns;
  }
  if (part_index >= 0)
    hipLaunchKernelGGL(k_attn_w4_part, dim3(grid, grid_y), dim3(NT4), lds_w4, (hipStream_t)stream, a);
  else if (prescaled)
    hipLaunchKernelGGL(k_attn_w4<4>, dim3(grid, grid_y), dim3(NT4), lds_w4, (hipStream_t)stream, a);
#ifdef WF_ATTN_LAB  // lab builds (tools/attn_lab.py) instantiate the timed kernel only: 5x shorter compile
  else
    WF_CHECK_ARG(false, "%s: lab build, pre-scaled self-attention only", who);
#else
  else if (Lkp > 1024)
    hipLaunchKernelGGL(k_attn_w4<0>, dim3(grid, grid_y), dim3(NT4), lds_w4, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_attn_w4<1>, dim3(grid, grid_y), dim3(NT4), lds_w4, (hipStream_t)stream, a);
#endif
  if (a.nsplit > 1 && part_index < 0) {
    const size_t n = (size_t)Lq * H * (D / 4);
    hipLaunchKernelGGL(k_attn_merge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  }
  WF_LAUNCH_CHECK(who);
  return WF_OK;
}

extern "C" int wf_attn_fwd(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int kv_len, int seg_len,
                           size_t seg_stride_bytes, int ldo, float softmax_scale, int accumulate, const float* kmax2, int kmax_n,
                           int kmax_stride, const float* qmax2, int qmax_n, void* stream) {
  return attn_launch(Q, K, Vt, O, H, Lq, Lkp, kv_len, seg_len, ldo, softmax_scale, accumulate, 1, nullptr, kmax2, kmax_n, qmax2, qmax_n,
                     stream, "wf_attn_fwd", seg_stride_bytes, kmax_stride);
}

// max over the rows of each head of |x|^2 (X bf16 [H][Lp][128], rows >= L ignored) -> out[h] (atomic max on the float bits: norms are
// >= 0, so the unsigned order is the float order; out must be zeroed by the caller)
namespace {
__global__ __launch_bounds__(256) void k_head_max_norm2(const uint16_t* __restrict__ X, int L, int Lp, int rows_per_block, float* __restrict__ out) {
  const int head = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(L, r0 + rows_per_block);
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;  // 16 lanes x 16 bytes = one 256-byte row
  float best = 0.f;
  // four rows per 16-lane group and trip: the loads are issued together, the four shuffle reductions interleave (a pure streaming pass:
  // 8 workgroups per head of 512 rows with one dependent load per trip ran at 1 TB/s)
  for (int r = r0 + grp; r < r1; r += 64) {
    u32x4 w4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 16 * u;
      w4[u] = rr < r1 ? *reinterpret_cast<const u32x4*>(X + ((size_t)head * Lp + rr) * D + sub * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(w4[u][e] << 16), hi = __uint_as_float(w4[u][e] & 0xffff0000u);
        sq[u] += lo * lo + hi * hi;
      }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) sq[u] += __shfl_xor(sq[u], o, 64);
    best = fmaxf(fmaxf(best, fmaxf(sq[0], sq[1])), fmaxf(sq[2], sq[3]));
    // fmaxf drops NaNs: a row holding a NaN (or an inf) must still push the attention kernel onto its tracked body -> report +inf
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (!(sq[u] <= 3.0e38f)) best = INFINITY;
  }
  // one atomic per workgroup at most, and only while the running maximum still grows: atomics on the 40 per-head words serialise in the
  // L2 (four per workgroup x 256 workgroups per head cost 100 us of a 137 us pass at 32 760 rows)
  __shared__ float wmax[4];
  best = wave_max(best);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    unsigned int* slot = reinterpret_cast<unsigned int*>(out + head);
    if (__float_as_uint(best) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(best));
  }
}
}  // namespace

extern "C" int wf_head_max_norm2(const void* X, int H, int L, int Lp, float* out, void* stream) {
  WF_CHECK_ARG(X && out, "wf_head_max_norm2: null pointer");
  WF_CHECK_ARG(H > 0 && L > 0 && Lp >= L, "wf_head_max_norm2: bad sizes H=%d L=%d Lp=%d", H, L, Lp);
  const int rows_per_block = 128;
  hipLaunchKernelGGL(k_head_max_norm2, dim3((L + rows_per_block - 1) / rows_per_block, H), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)X, L, Lp, rows_per_block, out);
  WF_LAUNCH_CHECK("wf_head_max_norm2");
  return WF_OK;
}

extern "C" int wf_attn_bsa_fwd(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int seg_len, int ldo,
                               float softmax_scale, const int* group_lists, const int* group_counts, int max_entries, int block,
                               void* stream) {
  WF_CHECK_ARG(Q && K && Vt && O && group_lists && group_counts, "wf_attn_bsa_fwd: null pointer");
  WF_CHECK_ARG(H > 0 && Lq > 0 && Lkp > 0 && max_entries > 0, "wf_attn_bsa_fwd: empty problem");
  WF_CHECK_ARG(block == 128 || block == 64, "wf_attn_bsa_fwd: block must be 128 or 64 tokens, got %d", block);
  WF_CHECK_ARG(Lq % block == 0 && Lkp % block == 0, "wf_attn_bsa_fwd: Lq (%d) and Lkp (%d) must be whole %d-token blocks", Lq, Lkp, block);
  WF_CHECK_ARG(seg_len > 0 && seg_len % block == 0 && Lkp % seg_len == 0, "wf_attn_bsa_fwd: seg_len (%d) must be whole blocks dividing Lkp",
               seg_len);
  WF_CHECK_ARG(ldo % 8 == 0 && ldo >= H * D && (((uintptr_t)O) & 15) == 0, "wf_attn_bsa_fwd: ldo %d must be a multiple of 8 (16-byte stores) and >= H * 128", ldo);
  WF_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O) & 15) == 0, "wf_attn_bsa_fwd: 16-byte alignment");
  AttnArgs a;
  a.Q = (const uint16_t*)Q;
  a.K = (const uint16_t*)K;
  a.Vt = (const uint16_t*)Vt;
  a.O = (uint16_t*)O;
  a.H = H;
  a.Lq = Lq;
  a.Lkp = Lkp;
  a.kv_len = Lkp;
  a.seg_len = seg_len;
  a.ldo = ldo;
  a.n_qblk = ceil_div(Lq, QB);
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  a.accumulate = 0;
  a.prio_mode = 0;
  a.nsplit = 1;
  a.tiles_per_split = 0;
  a.seg_stride = (size_t)a.H * a.seg_len * D * 2;
  a.kmax_stride = a.H;
  a.n_first = INT_MAX;
  a.t_begin2 = a.t_end2 = a.tiles_per_split2 = 0;
  a.t_begin0 = 0;
  a.t_end = INT_MAX;
  a.part0 = 0;
  a.o_part = nullptr;
  a.ml_part = nullptr;
  a.bsa_list = group_lists;
  a.bsa_cnt = group_counts;
  a.bsa_max = max_entries;
  a.bsa_shift = block == 128 ? 2 : 4;
  a.tiles_per_split = 0;
  a.kmax2 = a.qmax2 = nullptr;
  a.kmax_n = a.qmax_n = 0;
  a.dbg_body = nullptr;
  a.kv_len1 = 0;
  a.n1 = 0;
  const int grid = ((H + 7) / 8) * a.n_qblk * 8;
#ifdef WF_ATTN_LAB
  WF_CHECK_ARG(false, "%s: lab build, pre-scaled self-attention only", "wf_attn_bsa_fwd");
#else
  hipLaunchKernelGGL(k_attn_w4<3>, dim3(grid, 1), dim3(NT4), 5 * (K_TILE_BYTES + V_TILE_BYTES), (hipStream_t)stream, a);
#endif
  WF_LAUNCH_CHECK("wf_attn_bsa_fwd");
  return WF_OK;
}

extern "C" int wf_attn_cross2_fwd(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lk1p, int kv_len1, int Lk2p,
                                  int kv_len2, int ldo, float softmax_scale, void* stream) {
  WF_CHECK_ARG(Q && K && Vt && O, "wf_attn_cross2_fwd: null pointer");
  WF_CHECK_ARG(H > 0 && Lq > 0, "wf_attn_cross2_fwd: empty problem");
  WF_CHECK_ARG(Lk1p > 0 && Lk1p % KB == 0 && kv_len1 > Lk1p - KB && kv_len1 <= Lk1p,
               "wf_attn_cross2_fwd: context 1 must fill its %d padded rows up to the last 64-key tile (kv_len1 = %d)", Lk1p, kv_len1);
  WF_CHECK_ARG(Lk2p > 0 && Lk2p % KB == 0 && kv_len2 > Lk2p - KB && kv_len2 <= Lk2p,
               "wf_attn_cross2_fwd: context 2 must fill its %d padded rows up to the last 64-key tile (kv_len2 = %d)", Lk2p, kv_len2);
  WF_CHECK_ARG(ldo % 8 == 0 && ldo >= H * D && (((uintptr_t)O) & 15) == 0, "wf_attn_cross2_fwd: ldo %d must be a multiple of 8 (16-byte stores) and >= H * 128", ldo);
  WF_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O) & 15) == 0, "wf_attn_cross2_fwd: 16-byte alignment");
  WF_CHECK_ARG(softmax_scale > 0.0f, "wf_attn_cross2_fwd: softmax_scale must be positive (the scale is applied inside the kernel)");
  AttnArgs a;
  a.Q = (const uint16_t*)Q;
  a.K = (const uint16_t*)K;
  a.Vt = (const uint16_t*)Vt;
  a.O = (uint16_t*)O;
  a.H = H;
  a.Lq = Lq;
  a.Lkp = Lk1p + Lk2p;
  a.kv_len = kv_len2;
  a.seg_len = a.Lkp;
  a.ldo = ldo;
  a.n_qblk = ceil_div(Lq, QB);
  a.scale_log2 = softmax_scale * 1.4426950408889634f;
  a.accumulate = 0;
  a.prio_mode = 0;
  a.nsplit = 1;
  a.tiles_per_split = a.Lkp / KB;
  a.seg_stride = (size_t)a.H * a.seg_len * D * 2;
  a.kmax_stride = a.H;
  a.n_first = INT_MAX;
  a.t_begin2 = a.t_end2 = a.tiles_per_split2 = 0;
  a.t_begin0 = 0;
  a.t_end = INT_MAX;
  a.part0 = 0;
  a.o_part = nullptr;
  a.ml_part = nullptr;
  a.bsa_list = nullptr;
  a.bsa_cnt = nullptr;
  a.bsa_max = 0;
  a.bsa_shift = 2;
  a.kmax2 = a.qmax2 = nullptr;
  a.kmax_n = a.qmax_n = 0;
  a.dbg_body = nullptr;
  a.kv_len1 = kv_len1;
  a.n1 = Lk1p / KB;
  const int grid = ((H + 7) / 8) * a.n_qblk * 8;
#ifdef WF_ATTN_LAB
  WF_CHECK_ARG(false, "%s: lab build, pre-scaled self-attention only", "wf_attn_cross2_fwd");
#else
  hipLaunchKernelGGL(k_attn_w4<5>, dim3(grid, 1), dim3(NT4), 5 * (K_TILE_BYTES + V_TILE_BYTES), (hipStream_t)stream, a);
#endif
  WF_LAUNCH_CHECK("wf_attn_cross2_fwd");
  return WF_OK;
}

// The KV sweep of wf_attn_fwd in PARTS: the key segments of a sequence-parallel layer arrive one source rank after the other, and a part
// launch walks only the tiles [t_begin, t_end) that are there already (pre-scaled Q form) and leaves un-normalised partials (O f32, reference
// max, row sum) in slot `part` of an `nparts`-slot workspace (wf_attn_split_workspace_bytes(H, Lq, nparts)); wf_attn_merge combines the
// slots exactly (the flash combine of wf_attn_fwd_split) once every slot has been written.  The rank's own shard needs no wait at all, so a
// forward WITHOUT a second CFG branch to hide under still overlaps the exchange with attention itself.  Round 6: a second window behind a
// hole is walked by the SAME workgroups (one slot per launch and split), and the last launch of a sweep can fold the earlier slots in itself
// (O_merge) -- see attn_w4_body's PART notes.
extern "C" int wf_attn_fwd_part(const void* Q, const void* K, const void* Vt, int H, int Lq, int Lkp, int kv_len, int seg_len,
                                size_t seg_stride_bytes, int t_begin, int t_end, int t_begin2, int t_end2, int inner_splits, int part, int nparts,
                                void* workspace, void* O_merge, int ldo, const float* kmax2, int kmax_n, int kmax_stride, const float* qmax2,
                                int qmax_n, void* stream) {
  PartWindow pw;
  pw.index = part;
  pw.t0 = t_begin;
  pw.t1 = t_end;
  pw.inner = inner_splits;
  pw.t0b = t_begin2;
  pw.t1b = t_end2;
  pw.merge = O_merge != nullptr;
  WF_CHECK_ARG(part >= 0, "wf_attn_fwd_part: part index %d", part);
  // O is written by a merging (last) part launch only; the others hand attn_launch's pointer checks the workspace
  return attn_launch(Q, K, Vt, O_merge ? O_merge : workspace, H, Lq, Lkp, kv_len, seg_len, O_merge ? ldo : H * D, 0.0f, 0, nparts, workspace, kmax2,
                     kmax_n, qmax2, qmax_n, stream, "wf_attn_fwd_part", seg_stride_bytes, kmax_stride, pw);
}

extern "C" int wf_attn_merge(void* O, int H, int Lq, int ldo, int accumulate, int nparts, const void* workspace, void* stream) {
  WF_CHECK_ARG(O && workspace, "wf_attn_merge: null pointer");
  WF_CHECK_ARG(H > 0 && Lq > 0 && nparts >= 2 && nparts <= MAX_PARTS, "wf_attn_merge: H=%d Lq=%d nparts=%d (2..%d)", H, Lq, nparts, MAX_PARTS);
  WF_CHECK_ARG(ldo % 8 == 0 && ldo >= H * D && ((((uintptr_t)O) | ((uintptr_t)workspace)) & 15) == 0, "wf_attn_merge: ldo %d / alignment", ldo);
  AttnArgs a = {};
  a.O = (uint16_t*)O;
  a.H = H;
  a.Lq = Lq;
  a.ldo = ldo;
  a.accumulate = accumulate;
  a.scale_log2 = 1.0f;  // pre-scaled Q: the partial maxima are in the exp2 domain
  a.nsplit = nparts;
  a.o_part = (float*)const_cast<void*>(workspace);
  a.ml_part = a.o_part + (size_t)nparts * Lq * H * D;
  const size_t n = (size_t)Lq * H * (D / 4);
  hipLaunchKernelGGL(k_attn_merge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  WF_LAUNCH_CHECK("wf_attn_merge");
  return WF_OK;
}

extern "C" int wf_attn_debug_body_counter(void* counters2) {
  g_dbg_body = (unsigned int*)counters2;
  return WF_OK;
}

extern "C" size_t wf_attn_split_workspace_bytes(int H, int Lq, int nsplit) {
  if (H <= 0 || Lq <= 0 || nsplit <= 1) return 0;
  return ((size_t)nsplit * Lq * H * D + (size_t)nsplit * H * Lq * 2) * sizeof(float);
}

extern "C" int wf_attn_fwd_split(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int kv_len, int seg_len,
                                 size_t seg_stride_bytes, int ldo, float softmax_scale, int accumulate, int nsplit, void* workspace,
                                 const float* kmax2, int kmax_n, int kmax_stride, const float* qmax2, int qmax_n, void* stream) {
  WF_CHECK_ARG(nsplit >= 1 && nsplit <= 8, "wf_attn_fwd_split: nsplit %d out of range 1..8", nsplit);
  return attn_launch(Q, K, Vt, O, H, Lq, Lkp, kv_len, seg_len, ldo, softmax_scale, accumulate, nsplit, workspace, kmax2, kmax_n, qmax2,
                     qmax_n, stream, "wf_attn_fwd_split", seg_stride_bytes, kmax_stride);
}

#ifdef WF_ATTN_TIMING
extern "C" int wf_debug_attn_cycles(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_attn_cycles), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_cycles), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
