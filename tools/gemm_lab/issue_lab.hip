// Issue-slot lab (round 4): how many OTHER instructions does a lone wave (one per SIMD) issue for free beside a dense stream of bf16 MFMAs,
// for the two shapes v_mfma_f32_32x32x16_bf16 (32 pipe cycles each) and v_mfma_f32_16x16x32_bf16 (16 pipe cycles each)?
// The attention kernel k_attn_w4 is instruction-issue bound (profiles/r3_q_attn_lab.md): per 32 pipe cycles it wants ~2.5 VALU + 0.5
// ds_read_b128 + 0.125 LDS-DMA pieces + scalar bookkeeping beside the MFMA.  If the shorter MFMA frees more issue time per pipe cycle, the
// 16x16x32 form of that kernel can win; if every MFMA instruction blocks the wave's issue for a fixed time, it loses.  Measured in shader
// cycles (s_memtime), so the answer does not depend on the clock the power limit allows.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_lab/issue_lab.hip -o tools/gemm_lab/issue_lab     (built here, runs on the GPU box)
//   ./issue_lab            prints a table: shape x (VALU per 32 pipe cycles) x (ds_read_b128 per 32 pipe cycles) -> cycles per 32-cycle unit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;

// One "unit" = 32 pipe cycles = one 32x32x16 or two 16x16x32.  NV VALU instructions (alternating v_exp_f32 / v_add_f32 on 8 independent
// registers: no dependency closer than 8 instructions) and NL2 half-ds_reads (NL2 = 1 -> one ds_read_b128 every second unit) per unit.
template <int SHAPE, int NV, int NL2, bool TRANS>
__global__ __launch_bounds__(256, 1) void k_issue(const u32x4* __restrict__ src, float* __restrict__ sink, unsigned long long* cyc, int iters) {
  __shared__ u32x4 lds[2048];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += blockDim.x) lds[i] = src[i];
  __syncthreads();
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) a[i] = lds[(i * 64 + lane) & 2047];
  for (int i = 0; i < 4; ++i) b[i] = lds[(1024 + i * 64 + lane) & 2047];
  f32x16 acc[8];
  f32x4 acc16[32];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 32; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (SHAPE == 32)
    for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[i]));
  else
    for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(acc16[i]));
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i) * 1e-3f;
  u32x4 frag[4];
  for (int i = 0; i < 4; ++i) frag[i] = a[i];
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds + lane * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    int vcount = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {  // 16 units per iteration
      if constexpr (SHAPE == 32) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u & 7]) : "v"(frag[u & 3]), "v"(b[(u >> 2) & 3]));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int r = (u * NV + v) & 7;
          if (TRANS && (v & 1) == 0)
            asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
          else
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[r]) : "v"(x[(r + 4) & 7]));
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[(2 * u + h) & 31]) : "v"(frag[u & 3]), "v"(b[(u >> 2) & 3]));
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            if ((v & 1) != h) continue;  // the unit's VALU instructions split between its two MFMAs
            const int r = (u * NV + v) & 7;
            if (TRANS && (v & 1) == 0)
              asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
            else
              asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[r]) : "v"(x[(r + 4) & 7]));
          }
        }
      }
      if constexpr (NL2 > 0) {
        if ((u * NL2) % 2 == 0 || NL2 >= 2) {
          const int n = NL2 >= 2 ? NL2 / 2 : 1;
#pragma unroll
          for (int k = 0; k < n; ++k) {
            // the fragment this unit just consumed is refilled for the unit four later (address constant per lane: the read still has to issue and return)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[(u + 4 - k) & 3]) : "v"(lbase), "n"(((u * 2 + k) & 31) * 1024));
          }
        }
      }
      (void)vcount;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][lane & 15] + x[i];
  for (int i = 0; i < 32; ++i) s += acc16[i][lane & 3];
  for (int i = 0; i < 4; ++i) s += __uint_as_float(frag[i][0]);
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}


// The attention kernel's own mix per PAIR of units (64 pipe cycles): 2 v_exp_f32, the row-sum adds (MODE 0: 2 v_add_f32, MODE 1: ONE
// v_pk_add_f32 on a register pair, MODE 2: none), 1 v_cvt_pk_bf16_f32, 1 ds_read_b128, beside two 32x32x16 MFMAs.
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_softmax_mix(const u32x4* __restrict__ src, float* __restrict__ sink, unsigned long long* cyc, int iters) {
  __shared__ u32x4 lds[2048];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += blockDim.x) lds[i] = src[i];
  __syncthreads();
  u32x4 b[4];
  for (int i = 0; i < 4; ++i) b[i] = lds[(1024 + i * 64 + lane) & 2047];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[i]));
  f32x2 x[8], ls[2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)(lane + i) * -1e-3f, (float)(lane - i) * -1e-3f};
  u32x4 frag[4];
  for (int i = 0; i < 4; ++i) frag[i] = lds[(i * 64 + lane) & 2047];
  uint32_t pk = 0;
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds + lane * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      f32x2 p;
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u & 7]) : "v"(frag[u & 3]), "v"(b[(u >> 2) & 3]));
      asm volatile("v_exp_f32 %0, %1" : "=v"(p[0]) : "v"(x[(u >> 1) & 7][0]));
      asm volatile("v_exp_f32 %0, %1" : "=v"(p[1]) : "v"(x[(u >> 1) & 7][1]));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[u & 3]) : "v"(lbase), "n"((u & 31) * 1024));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[(u + 1) & 7]) : "v"(frag[(u + 1) & 3]), "v"(b[(u >> 2) & 3]));
      if constexpr (MODE == 0) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[(u >> 1) & 1][0]) : "v"(p[0]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[(u >> 1) & 1][1]) : "v"(p[1]));
      } else if constexpr (MODE == 1) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ls[(u >> 1) & 1]) : "v"(p));
      }
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(p[0]), "v"(p[1]));
      asm volatile("" :: "v"(pk));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = ls[0][0] + ls[0][1] + ls[1][0] + ls[1][1] + __uint_as_float(pk);
  for (int i = 0; i < 8; ++i) s += acc[i][lane & 15] + x[i][0];
  for (int i = 0; i < 4; ++i) s += __uint_as_float(frag[i][0]);
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
template <int MODE>
static double run_mix(const u32x4* d, float* sink, unsigned long long* cyc) {
  const int iters = 4000;
  hipLaunchKernelGGL((k_softmax_mix<MODE>), dim3(256), dim3(256), 0, 0, d, sink, cyc, 200);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k_softmax_mix<MODE>), dim3(256), dim3(256), 0, 0, d, sink, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  return (double)c / ((double)iters * 16);
}

template <int SHAPE, int NV, int NL2, bool TRANS>
static double run(const u32x4* d, float* sink, unsigned long long* cyc) {
  const int iters = 4000;
  hipLaunchKernelGGL((k_issue<SHAPE, NV, NL2, TRANS>), dim3(256), dim3(256), 0, 0, d, sink, cyc, 200);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k_issue<SHAPE, NV, NL2, TRANS>), dim3(256), dim3(256), 0, 0, d, sink, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  return (double)c / ((double)iters * 16);
}

template <int SHAPE, int NL2, bool TRANS>
static void row(const u32x4* d, float* sink, unsigned long long* cyc) {
  printf("| %dx%d | %s | %.1f |", SHAPE, SHAPE, TRANS ? "exp+add" : "add", NL2 * 0.5);
  printf(" %.1f |", run<SHAPE, 0, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |", run<SHAPE, 1, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |", run<SHAPE, 2, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |", run<SHAPE, 3, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |", run<SHAPE, 4, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |", run<SHAPE, 5, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |", run<SHAPE, 6, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |", run<SHAPE, 8, NL2, TRANS>(d, sink, cyc));
  printf(" %.1f |\n", run<SHAPE, 10, NL2, TRANS>(d, sink, cyc));
}

int main() {
  std::vector<uint32_t> h(2048 * 4, 0x3f803f80u);  // bf16 1.0 pairs
  u32x4* d; float* sink; unsigned long long* cyc;
  hipMalloc(&d, h.size() * 4); hipMalloc(&sink, 64); hipMalloc(&cyc, 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("shader cycles per 32 matrix-pipe cycles (one 32x32x16 or two 16x16x32 MFMAs), lone wave per SIMD, vs other instructions per unit\n");
  printf("| MFMA | VALU kind | ds_read_b128 per unit | 0 VALU | 1 | 2 | 3 | 4 | 5 | 6 | 8 | 10 |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n");
  row<32, 0, false>(d, sink, cyc);
  row<16, 0, false>(d, sink, cyc);
  row<32, 0, true>(d, sink, cyc);
  row<16, 0, true>(d, sink, cyc);
  row<32, 1, true>(d, sink, cyc);
  row<16, 1, true>(d, sink, cyc);
  row<32, 2, true>(d, sink, cyc);
  row<16, 2, true>(d, sink, cyc);
  printf("\nattention mix per 64 pipe cycles (2 MFMA 32x32x16 + 2 v_exp + row-sum adds + 1 v_cvt_pk_bf16 + 1 ds_read_b128), cycles per 32-cycle unit:\n");
  printf("| 2 x v_add_f32 | 1 x v_pk_add_f32 | no adds |\n|---|---|---|\n| %.1f | %.1f | %.1f |\n", run_mix<0>(d, sink, cyc), run_mix<1>(d, sink, cyc), run_mix<2>(d, sink, cyc));
  return 0;
}
