// Box calibration (VERDICT r4 "next" #5): the boxes of the pool differ by up to 6 % in what they sustain on the same binary (socket power
// limit / silicon), more than most single changes of a round are worth.  bench.py therefore times, right before and right after its timed
// window, a FIXED register-only stream of v_mfma_f32_32x32x16_bf16 on N(0,1) operands (the round-3 energy lab's variant 0: one wave per SIMD,
// 16 accumulator tiles in AGPRs, 32.0 cycles per MFMA = perfect issue, nothing but the matrix pipe busy) and prints the rate as
// `box_calib_tflops` next to a `value_normalised` -- a diagnostic for comparing lines across boxes, never the contract value.
#include "common.h"
#include "mfma.h"

using namespace wf;

namespace {

__global__ __launch_bounds__(256, 1) void k_calib_mfma(const u32x4* __restrict__ src, float* __restrict__ sink, int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  u32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(blockIdx.x * 1024 + i * 64 + lane) & 65535];
    b[i] = src[(blockIdx.x * 1024 + 512 + i * 64 + lane) & 65535];
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      asm volatile("" : "+a"(acc[i][j]));  // accumulators live in AGPRs for the whole loop
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][lane & 15];
  if (s == 12345.678f) sink[0] = s;  // never true for the data the host provides; keeps the loop alive
}

// A stream-ordered delay: one wave spinning on the constant-rate wall clock (100 MHz) with s_sleep between polls.  parallel.LoopbackComm
// uses it to stand in for the TRANSFER TIME of a collective when one GPU plays one rank of N under a bandwidth model (bench.py
// --as-rank-of N --emulate-comm ...): the local copies that serve the collective cost a few hundred microseconds, a real exchange over
// xGMI takes bytes / rate.  Like an RCCL copy kernel it needs a CU slot to start and holds one while it runs.
__global__ void k_delay(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace

extern "C" int wf_delay_us(double us, void* stream) {
  WF_CHECK_ARG(us >= 0.0 && us < 5.0e6, "wf_delay_us: %f microseconds out of range (0 .. 5 s)", us);
  if (us <= 0.0) return WF_OK;
  hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)(us * 100.0));
  WF_LAUNCH_CHECK("wf_delay_us");
  return WF_OK;
}

// One launch of the calibration stream: 256 workgroups x 4 waves x iters x 16 MFMAs of 2 * 32 * 32 * 16 flop.  src: 1 MiB of bf16 operand
// data in device memory (the host fills it with N(0,1) values: the rate depends on the data -- zeros run 35 % faster), sink: >= 4 bytes.
// Asynchronous on `stream`; the caller times it with events.  Returns the flop count of the launch through *flop (host pointer, may be NULL).
extern "C" int wf_calib_mfma(const void* src, float* sink, int iters, double* flop, void* stream) {
  WF_CHECK_ARG(src && sink && iters > 0, "wf_calib_mfma: bad arguments");
  WF_CHECK_ARG((((uintptr_t)src) & 15) == 0, "wf_calib_mfma: 16-byte alignment");
  hipLaunchKernelGGL(k_calib_mfma, dim3(256), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, sink, iters);
  WF_LAUNCH_CHECK("wf_calib_mfma");
  if (flop) *flop = 256.0 * 4.0 * (double)iters * 16.0 * 2.0 * 32.0 * 32.0 * 16.0;
  return WF_OK;
}
