// MFMA fragment types and helpers for gfx950 (CDNA4).  64-lane wavefronts, v_mfma_f32_32x32x16_bf16.
//
// Operand maps of __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c) -- D[i][n] = sum_k A[i][k] * B[k][n] + C[i][n]:
//   A: lane l holds A[i = l & 31][k = 8*(l >> 5) + j], j = 0..7          (8 bf16 = 4 VGPRs)
//   B: lane l holds B[k = 8*(l >> 5) + j][n = l & 31], j = 0..7
//   C/D: lane l, register r (0..15): column n = l & 31, row i = (r & 3) + 8*(r >> 2) + 4*(l >> 5)
// So a lane owns ONE column and, per register quad, 4 consecutive rows: with A = the operand whose index should end up
// contiguous in memory (output features / head-dim) and B = the token operand, every lane writes 4 consecutive
// outputs of one token row (8-B bf16 or 16-B f32 stores).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
  union {
    u32x4 u;
    bf16x8 b;
  } x;
  x.u = v;
  return x.b;
}

// pack two f32 -> two bf16 (round-to-nearest-even) in one dword: lo in bits 0..15
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
  a += 0x7fffu + ((a >> 16) & 1u);
  b += 0x7fffu + ((b >> 16) & 1u);
  return (a >> 16) | (b & 0xffff0000u);
}

}  // namespace wf
