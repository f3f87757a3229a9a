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

#include <type_traits>
#include <utility>

namespace wf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// The same instruction shape on fp16 operands (v_mfma_f32_32x32x16_f16, same rate): the VAE's "fp16x3" / "fp16" operand formats
// (11-bit significands; round 4).  Kernels that serve both formats take the element type as a template flag and keep their operand
// fragments as raw u32x4.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16>
__device__ __forceinline__ f32x16 mfma32t(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (F16) {
    union { u32x4 u; f16x8 h; } xa, xb;
    xa.u = a;
    xb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(xa.h, xb.h, c, 0, 0, 0);
  } else {
    union { u32x4 u; bf16x8 h; } xa, xb;
    xa.u = a;
    xb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa.h, xb.h, c, 0, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
  union {
    u32x4 u;
    bf16x8 b;
  } x;
  x.u = v;
  return x.b;
}

// pack two f32 -> two bf16 (round-to-nearest-even) in one dword, lo in bits 0..15: one v_cvt_pk_bf16_f32 on gfx950
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  union {
    bf16x2_t b;
    uint32_t u;
  } x;
  x.b = __builtin_convertvector(v, bf16x2_t);
  return x.u;
}

// two f32 -> two fp16 (round-to-nearest-even) in one dword, lo in bits 0..15
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  union {
    f16x2_t h;
    uint32_t u;
  } x;
  x.h = __builtin_convertvector(v, f16x2_t);
  return x.u;
}
// 16-bit operand copy of an epilogue: fp16 when the launch's operands are fp16 (wave-uniform flag), bf16 otherwise
__device__ __forceinline__ uint32_t pack16x2(int f16, float lo, float hi) { return f16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// async global -> LDS copy, 16 bytes per lane; LDS destination = wave-uniform base + lane * 16 (so any swizzle goes on the
// per-lane SOURCE address), completion tracked by vmcnt
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
// The same LDS-DMA issued from inline asm, invisible to hipcc's s_waitcnt bookkeeping: with the builtin the compiler drains
// vmcnt(0) in front of the next ds_read (it cannot prove the DMA's LDS destination does not alias it), which serialises the
// copy with the MFMA phase it is supposed to hide under.  The caller owns the ordering: s_waitcnt vmcnt(N) + a workgroup barrier
// before anyone reads the destination.  M0 (the LDS base of the DMA) is compiler-reserved: saved / restored in the statement.
__device__ __forceinline__ void glds16_async(const void* gsrc, uint32_t lds_wave_base_byte) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_wave_base_byte)
      : "memory");
}
// glds16_async without the M0 save / restore: M0 is declared clobbered instead (DS instructions do not read M0 on gfx9+, and the
// compiler re-materialises it wherever it needs it) -- two scalar instructions fewer per piece in a hand-scheduled MFMA stream.
__device__ __forceinline__ void glds16_async_m0(const void* gsrc, uint32_t lds_wave_base_byte) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_wave_base_byte)
      : "memory", "m0");
}
// glds16_async_m0 with the piece's LDS offset as an immediate: M0 = wave-uniform base + IMM in ONE scalar instruction (the 16 per-piece
// destinations of k_conv_w4 were 16 scalars the compiler kept spilled in VGPR lanes: a v_readlane + s_add per piece)
template <int IMM>
__device__ __forceinline__ void glds16_async_m0_imm(const void* gsrc, uint32_t lds_base_byte) {
  asm volatile(
      "s_add_u32 m0, %1, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_base_byte), "n"(IMM)
      : "memory", "m0", "scc");
}
// LDS-DMA in the saddr form: wave-uniform 64-bit base in SGPRs + 32-bit per-lane byte offset -- no 64-bit VALU address arithmetic.
// Inline asm like glds16_async (the caller owns the ordering: s_waitcnt vmcnt + barrier before anyone reads the destination).
__device__ __forceinline__ void glds16_saddr(const void* base_uniform, uint32_t voff_bytes, uint32_t lds_wave_base_byte) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff_bytes), "s"(base_uniform), "s"(lds_wave_base_byte)
      : "memory", "m0");
}
// the same with the piece's LDS offset as an immediate: M0 = uniform ring-slot base + IMM in ONE scalar instruction
template <int IMM>
__device__ __forceinline__ void glds16_saddr_imm(const void* base_uniform, uint32_t voff_bytes, uint32_t lds_slot_base_byte) {
  asm volatile(
      "s_add_u32 m0, %2, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff_bytes), "s"(base_uniform), "s"(lds_slot_base_byte), "n"(IMM)
      : "memory", "m0", "scc");
}
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// compile-time loop: f(integral_constant<int, 0>), ..., f(integral_constant<int, N-1>).  Hand-placed instruction streams need every
// index to be a constant (a #pragma unroll that the optimizer declines turns register arrays into scratch).
template <class F, int... Is>
__device__ __forceinline__ void for_const_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void for_const(F&& f) {
  for_const_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

}  // namespace wf
