// Probe: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (run once on the GPU box; used to design lo_precond.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
  const int l = threadIdx.x;
  // hypothesis: a = A[i = l % 16][k = l / 16], b = B[k = l / 16][j = l % 16]
  const int i = l % 16, kk = l / 16;
  const double a = (kk == 1) ? (double)(i + 1) : 0.0;          // A[i][1] = i + 1
  const double b = (kk == 1) ? (double)(100 * (i + 1)) : 0.0;  // B[1][j] = 100 (j + 1)
  double4v c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double* d;
  hipMalloc(&d, 256 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok_hyp = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int v = (int)h[l * 4 + r];  // = (i+1) * 100 (j+1)
      const int j1 = v / 100, rem = v % 100;
      (void)rem;
      // decode i, j: v = 100 * (i+1) * (j+1): ambiguous in general; check the hypothesis D[i = 4*(l/16) + r][j = l%16]
      const int ei = 4 * (l / 16) + r, ej = l % 16;
      if (v != 100 * (ei + 1) * (ej + 1)) ok_hyp = 0;
      if (l < 20 || l > 60) printf("lane %d reg %d -> %d (j1 %d)\n", l, r, v, j1);
    }
  printf("hypothesis D[4*(l/16)+r][l%%16]: %s\n", ok_hyp ? "OK" : "WRONG");
  return 0;
}
