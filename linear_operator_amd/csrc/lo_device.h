// lo_device.h -- device-side helpers (wave64 / 256-thread workgroups, gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace lo {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Sum over all threads of a 256-thread block; result valid in every thread.  `red` >= 4 floats of LDS.
// Fixed order (wave butterfly, then waves 0..3) -> bitwise reproducible.
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Column-wise block reduction for the "thread t owns column t % c, row slot t / c" map:
// threads t < c*nrs participate; returns the sum over row slots in threads with slot == 0 (t < c).
// `red` = 256 floats of LDS.  Fixed tree order.
__device__ __forceinline__ float block_colsum(float v, int c, int nrs, float* red) {
  const int t = threadIdx.x;
  const int slot = t / c;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  int h = 1;
  while (h < nrs) h <<= 1;
  for (h >>= 1; h >= 1; h >>= 1) {
    if (slot < h && slot + h < nrs) red[t] += red[t + h * c];
    __syncthreads();
  }
  return red[t];
}

}  // namespace lo
