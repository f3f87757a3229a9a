// MFMA energy lab (VERDICT r2 "Next" #5): what does the chip sustain on bf16 MFMA streams when NOTHING but the matrix pipe (and,
// optionally, LDS fragment reads) is busy, on N(0,1) operands vs zeros?  If a register-only stream of v_mfma_f32_32x32x16_bf16 on random
// data already sits at the clock the DiT GEMMs run at, no re-tiling of the GEMM can buy throughput; if it clocks much higher, the gap is
// the energy of the operand traffic and is worth attacking.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_lab/mfma_lab.hip -o tools/gemm_lab/mfma_lab      (built here, runs on the GPU box)
//   ./mfma_lab <variant> <data: 0 zeros | 1 N(0,1)> [seconds]
// variants: 0 = 32x32x16, operands in registers (16 accumulator tiles = 256 AGPR-class registers, 4 waves per CU, one per SIMD)
//           1 = 16x16x32, operands in registers (same flops per iteration)
//           2 = 32x32x16 + 8 ds_read_b128 per 16 MFMAs (the 0.5 fragment reads per MFMA of a 128 x 128 wave tile)
//           3 = 32x32x16 + 12 ds_read_b128 per 16 MFMAs (0.75: the 256-wide ping-pong tile)
//           4 = 32x32x16, TWO waves per SIMD (8 waves per CU, 8 accumulator tiles each), registers only
// Prints one line: variant, data, TFLOP/s, effective clock (shader cycles of one wave / wall time), MFMA issue efficiency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include <chrono>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int VAR>
__global__ __launch_bounds__(VAR == 4 ? 512 : 256, 1) void k_lab(const u32x4* __restrict__ src, float* __restrict__ sink, unsigned long long* cyc,
                                                                  int iters) {
  __shared__ u32x4 lds[4096];  // 64 KiB of fragments
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = src[(blockIdx.x * 4096 + i) % (1 << 16)];
  __syncthreads();
  constexpr int NA = VAR == 4 ? 2 : 4, NB = 4;
  u32x4 a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = lds[(i * 64 + lane) & 4095];
  for (int i = 0; i < NB; ++i) b[i] = lds[(1024 + i * 64 + lane) & 4095];
  f32x16 acc[NA][NB];
  constexpr int N16 = 48;  // 16x16x32 accumulator tiles (192 registers)
  f32x4 acc16[VAR == 1 ? N16 : 1];
  for (int i = 0; i < NA; ++i)
    for (int j = 0; j < NB; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (VAR == 1)
    for (int i = 0; i < N16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < NA; ++i)
    for (int j = 0; j < NB; ++j) asm volatile("" : "+a"(acc[i][j]));   // accumulators live in AGPRs for the whole loop (inline-asm MFMAs)
  if constexpr (VAR == 1)
    for (int i = 0; i < N16; ++i) asm volatile("" : "+a"(acc16[i]));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (VAR == 1) {
      // 48 x (16x16x32): 48 * 2*16*16*32 flop = 1.5 x what 16 x (32x32x16) do (counted as 24 units of 2*32*32*16 in run())
#pragma unroll
      for (int i = 0; i < N16; ++i)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
          if constexpr (VAR == 2 || VAR == 3) {
            constexpr int RD = VAR == 2 ? 8 : 12;
            const int idx = i * NB + j;
            if (idx < RD) {  // refresh one fragment from LDS (address varies with the iteration so the read cannot be hoisted)
              const u32x4 v = lds[(it * 64 + idx * 256 + lane) & 4095];
              if (idx & 1) a[(idx >> 1) & 3] = v; else b[(idx >> 1) & 3] = v;
            }
          }
        }
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NA; ++i)
    for (int j = 0; j < NB; ++j) s += acc[i][j][lane & 15];
  if constexpr (VAR == 1)
    for (int i = 0; i < N16; ++i) s += acc16[i][lane & 3];
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int VAR>
static void run(int data, double seconds) {
  const int n = 1 << 16;
  std::vector<uint32_t> h((size_t)n * 4);
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& w : h) {
    if (!data) { w = 0; continue; }
    uint32_t lo, hi;
    float x = nd(rng), y = nd(rng);
    memcpy(&lo, &x, 4); memcpy(&hi, &y, 4);
    w = (lo >> 16) | (hi & 0xffff0000u);  // two bf16 (truncated) N(0,1) values
  }
  u32x4* d; float* sink; unsigned long long* cyc;
  hipMalloc(&d, h.size() * 4); hipMalloc(&sink, 64); hipMalloc(&cyc, 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int blocks = 256, threads = VAR == 4 ? 512 : 256;
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_lab<VAR>, dim3(blocks), dim3(threads), 0, 0, d, sink, cyc, 2000);  // warm-up
  hipDeviceSynchronize();
  double total_ms = 0, flop = 0; unsigned long long cycles = 0; double last_ms = 0;
  const auto tstart = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count() < seconds) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_lab<VAR>, dim3(blocks), dim3(threads), 0, 0, d, sink, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms; last_ms = ms;
    flop += (double)blocks * (threads / 64) * iters * (VAR == 4 ? 8.0 : (VAR == 1 ? 24.0 : 16.0)) * 2.0 * 32 * 32 * 16;
    hipMemcpy(&cycles, cyc, 8, hipMemcpyDeviceToHost);
  }
  const double mfma_per_wave = (double)iters * (VAR == 1 ? 48 : (VAR == 4 ? 8 : 16));
  const double cyc_per_mfma = (double)cycles / mfma_per_wave;
  printf("variant %d data %s: %.0f TFLOP/s, clock %.2f GHz, %.1f cycles per MFMA per wave (ideal %d), %.0f ms per launch\n", VAR,
         data ? "N(0,1)" : "zeros", flop / (total_ms * 1e-3) / 1e12, (double)cycles / (last_ms * 1e-3) / 1e9, cyc_per_mfma,
         VAR == 1 ? 16 : (VAR == 4 ? 64 : 32), last_ms);
  hipFree(d); hipFree(sink); hipFree(cyc);
}

int main(int argc, char** argv) {
  const int var = argc > 1 ? atoi(argv[1]) : 0, data = argc > 2 ? atoi(argv[2]) : 1;
  const double sec = argc > 3 ? atof(argv[3]) : 3.0;
  switch (var) {
    case 0: run<0>(data, sec); break;
    case 1: run<1>(data, sec); break;
    case 2: run<2>(data, sec); break;
    case 3: run<3>(data, sec); break;
    case 4: run<4>(data, sec); break;
    default: fprintf(stderr, "variant 0..4\n"); return 2;
  }
  return 0;
}
