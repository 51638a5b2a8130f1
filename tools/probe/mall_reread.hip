// Probe: what does a second read of a 1 MiB chunk cost when it follows the first by LAG MiB of other streamed data?
// (design question of round 5: may the R-space CG kernel drop a member's rows between its reduction pass and its
// x = D^-1 (xi b + C y) pass and read them again from the Infinity Cache, instead of holding them in VGPRs?)
// Persistent workgroups walk the chunks of a buffer in order (chunk c -> workgroups c * WPC .. c * WPC + WPC - 1 of a
// round); with REREAD a workgroup also reads chunk c - lag.  Prints GB/s of first-pass bytes for each lag.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_reread tools/probe/mall_reread.hip && /tmp/mall_reread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int TPB = 256;
constexpr size_t CHUNK = 1 << 20;              // bytes per chunk (one member's C)
constexpr int WPC = 8;                         // workgroups per chunk (the group of the CG kernel)
constexpr int PER = CHUNK / WPC / TPB / 16;    // float4 loads per thread per chunk slice (32)

__global__ __launch_bounds__(TPB) void k(const float4* __restrict__ buf, int nchunks, int lag, float* sink) {
  const int ngroups = gridDim.x / WPC;
  const int grp = blockIdx.x / WPC, wig = blockIdx.x % WPC;
  float acc = 0.f;
  for (int c = grp; c < nchunks; c += ngroups) {
    const float4* p = buf + ((size_t)c * CHUNK + (size_t)wig * (CHUNK / WPC)) / 16;
    float4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = p[i * TPB + threadIdx.x];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    if (lag > 0 && c - lag >= 0) {
      const float4* q = buf + ((size_t)(c - lag) * CHUNK + (size_t)wig * (CHUNK / WPC)) / 16;
#pragma unroll
      for (int i = 0; i < PER; ++i) v[i] = q[i * TPB + threadIdx.x];
#pragma unroll
      for (int i = 0; i < PER; ++i) acc += v[i].x * 0.5f + v[i].y + v[i].z + v[i].w;
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int nchunks = 2048;  // 2 GiB
  float4* buf;
  float* sink;
  CK(hipMalloc(&buf, (size_t)nchunks * CHUNK));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, (size_t)nchunks * CHUNK));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int lags[] = {0, 8, 32, 64, 128, 160, 200, 240, 512};
  for (int wgs_per_cu : {2, 4, 8}) {
    const int grid = 256 * wgs_per_cu;
    for (int lag : lags) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(TPB), 0, 0, buf, nchunks, lag, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double first = (double)nchunks * CHUNK;
      printf("wgs/cu %d  lag %4d MiB (%d groups in flight = %d MiB): %.3f ms  first-pass %.0f GB/s  total-read %.0f GB/s\n",
             wgs_per_cu, lag, grid / WPC, grid / WPC, best, first / best / 1e6,
             (lag ? (first + (double)(nchunks - lag) * CHUNK) : first) / best / 1e6);
    }
  }
  return 0;
}
