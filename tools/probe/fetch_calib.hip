// Probe: what does rocprofv3's FETCH_SIZE report for a known byte count, per access pattern?  (the
// x 2 correction of the guide is calibrated for 16-byte coalesced streams; k_cg_step_cols reads its [N, 17] vectors with
// 4-byte loads in the accumulator layout of v_mfma_f32_16x16x4_f32.)  Each kernel reads the SAME 544 MB buffer exactly
// once (8 Mi rows x 17 floats); FETCH_SIZE x 1024 / bytes is the factor to apply for that pattern.
//   k_calib_b128      16 bytes per lane, consecutive lanes consecutive (the guide's calibrated case)
//   k_calib_b32       4 bytes per lane, consecutive lanes consecutive
//   k_calib_acc17     4 bytes per lane in the accumulator layout over [rows, 17]: lane (n = l & 15, kk = l >> 4) reads
//                     rows 16 rb + 4 kk + i, column n of the first column tile and -- lanes with n == 0 -- column 16
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/probe/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc -- /tmp/fetch_calib
//   python tools/pmc_summary.py /tmp/fc k_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr size_t ROWS = (size_t)8 << 20;
constexpr int C = 17;
constexpr size_t FLOATS = ROWS * C;  // 142.6 M floats = 570 MB

__global__ __launch_bounds__(256) void k_calib_b128(const float4* __restrict__ p, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_b32(const float* __restrict__ p, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 12345.678f) sink[0] = acc;
}
// a wave owns 64 rows per step (4 blocks of 16 rows), as a workgroup of k_cg_step_cols owns 256
__global__ __launch_bounds__(256) void k_calib_acc17(const float* __restrict__ p, size_t rows, float* sink) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = l & 15, kk = l >> 4;
  float acc = 0.f;
  for (size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 64; r0 < rows; r0 += (size_t)gridDim.x * 256) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float v[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t row = r0 + 16 * rb + 4 * kk + i;
        v[i] = p[row * C + n];
        w[i] = (n == 0) ? p[row * C + 16] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += v[i] + w[i];
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  float* buf;
  float* sink;
  CK(hipMalloc(&buf, FLOATS * 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, FLOATS * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    for (int which = 0; which < 3; ++which) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_calib_b128, dim3(4096), dim3(256), 0, 0, (const float4*)buf, FLOATS / 4, sink);
      if (which == 1) hipLaunchKernelGGL(k_calib_b32, dim3(4096), dim3(256), 0, 0, buf, FLOATS, sink);
      if (which == 2) hipLaunchKernelGGL(k_calib_acc17, dim3(4096), dim3(256), 0, 0, buf, ROWS, sink);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s: %.1f MB read once in %.3f ms = %.0f GB/s\n", which == 0 ? "b128" : (which == 1 ? "b32" : "acc17"),
             FLOATS * 4 / 1e6, ms, FLOATS * 4 / ms / 1e6);
    }
  }
  return 0;
}
