// Issue rate of v_pk_fma_f32 against v_fma_f32 (same flops): is packed fp32 worth restructuring a VALU-bound kernel?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  v2f acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = v2f{(float)threadIdx.x + j, 1.f + j};
  v2f va = {a, a + 1e-3f}, vb = {b, b - 1e-3f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (PK) acc[j] = __builtin_elementwise_fma(acc[j], va, vb);
        else {
          acc[j].x = fmaf(acc[j].x, va.x, vb.x);
          acc[j].y = fmaf(acc[j].y, va.y, vb.y);
        }
      }
  }
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pk = 0; pk < 2; ++pk) for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    if (pk) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, out, 0.999f, 0.5f, 4096);
    else hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, out, 0.999f, 0.5f, 4096);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 2048 * 256 * 4096.0 * 64;
    if (rep) printf("%s: %.3f ms, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, flops / ms / 1e9);
  }
  return 0;
}
