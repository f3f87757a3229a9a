// Shared device/host helpers for libwf_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/wf_hip.h"

namespace wf {

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define WF_CHECK_ARG(cond, ...)                  \
  do {                                           \
    if (!(cond)) {                               \
      wf::set_error(__VA_ARGS__);                \
      return WF_EINVAL;                          \
    }                                            \
  } while (0)

#define WF_LAUNCH_CHECK(name)                                  \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return wf::check_hip(e__, name);    \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch's .to(bfloat16)) ---
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// value-level rounding through bf16 (what a torch bf16 elementwise op does to its result)
__device__ __forceinline__ float rbf(float f) { return bf16_to_f32(f32_to_bf16(f)); }
template <bool RB>
__device__ __forceinline__ float rnd(float f) {
  if constexpr (RB) return rbf(f);
  return f;
}

// ---- dtype-erased scalar access for the small latent-shaped tensors --------
struct TView {
  void* p;
  int dt;  // WF_F32 / WF_BF16
};
__device__ __forceinline__ float tload(const TView& t, size_t i) {
  if (t.dt == WF_BF16) return bf16_to_f32(((const uint16_t*)t.p)[i]);
  return ((const float*)t.p)[i];
}
__device__ __forceinline__ void tstore(const TView& t, size_t i, float v) {
  if (t.dt == WF_BF16)
    ((uint16_t*)t.p)[i] = f32_to_bf16(v);
  else
    ((float*)t.p)[i] = v;
}

// ---- wave (64 lanes) / block reductions -------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int grid_for(size_t n, int block, int max_blocks = 2048) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace wf
