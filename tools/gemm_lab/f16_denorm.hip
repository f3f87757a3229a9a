// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 SUBNORMAL inputs (needed by the fp16 three-term operand split of the VAE: the lo
// term of a value below ~0.06 is an fp16 subnormal)?  A[i][k] = a (all entries), B[k][n] = b: D = 16 a b.  Prints D for a few (a, b).
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_lab/f16_denorm.hip -o tools/gemm_lab/f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float a, float b, float* out) {
  f16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = (_Float16)a; vb[i] = (_Float16)b; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)va[0]; out[2] = (float)vb[0]; }
}
int main() {
  float* d; hipMalloc(&d, 16);
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {6.0e-8f, 1.0f}, {1.0e-6f, 1.0e-6f}, {5.0e-7f, 1024.0f}, {6.1e-5f, 6.1e-5f}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("a = %.3e (as f16 %.6e)  b = %.3e (as f16 %.6e)  mfma D = %.6e  expected 16 a b = %.6e\n", c[0], h[1], c[1], h[2], h[0], 16.0 * h[1] * h[2]);
  }
  return 0;
}
