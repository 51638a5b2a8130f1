// Probe: sustained rate of v_mfma_f32_32x32x2_f32 (and 16x16x4) on gfx950, registers only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = __sinf(threadIdx.x * 12.9898f + blockIdx.x) * 43758.5453f; x = x - floorf(x) - 0.5f; float y = __cosf(threadIdx.x * 78.233f) * 0.7f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f4v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = __sinf(threadIdx.x * 12.9898f + blockIdx.x) * 43758.5453f; x = x - floorf(x) - 0.5f; float y = __cosf(threadIdx.x * 78.233f) * 0.7f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
  float* d;
  hipMalloc(&d, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 2048;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      if (which == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (which == 0 ? 4096.0 : 2048.0) * 4 * iters * 4.0 * blocks;  // per MFMA x 4 acc x iters x 4 waves x blocks
      if (rep) printf("%s: %.2f ms, %.1f TFLOP/s\n", which == 0 ? "mfma_f32_32x32x2" : "mfma_f32_16x16x4", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
