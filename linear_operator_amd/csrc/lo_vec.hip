// lo_vec.hip -- elementwise / reduction kernels over vectors [B, N, c] (c innermost).
// Thread map: a member's row range is a contiguous slab; thread t owns column t % c and row slot t / c
// (threads >= c * (256 / c) idle), so consecutive lanes touch consecutive addresses and each thread
// keeps ONE accumulator per reduction.  Partials are [B, S, c], summed in fixed order by the consumer.
#include <algorithm>

#include "lo_device.h"
#include <stdint.h>

#include "lo_internal.h"

namespace lo {

Split choose_split(int64_t B, int64_t N, int min_rows, int max_split) {
  // aim for >= ~1024 workgroups (4 per CU) but keep >= min_rows rows per workgroup; at most max_split slices (256: one
  // very tall member -- N = 1M rows -- used to get 64 workgroups = a quarter of the CUs, 0.7 TB/s)
  int64_t S = (1024 + B - 1) / B;
  int64_t maxS = std::max<int64_t>(1, N / std::max(min_rows, 4));
  S = std::max<int64_t>(1, std::min<int64_t>(S, maxS));
  // (every consumer sums the S partials of a member: beyond 64 slices only as far as a slice keeps >= 16 S rows)
  int64_t cap = 64;
  while (cap < max_split && (cap * 2) * (cap * 2) * 16 <= N) cap *= 2;
  S = std::min<int64_t>(S, std::min<int64_t>(cap, max_split));
  int64_t rows = (N + S - 1) / S;
  rows = (rows + 3) / 4 * 4;
  S = (N + rows - 1) / rows;
  Split sp;
  sp.S = (int)S;
  sp.rows = (int)rows;
  return sp;
}

__global__ __launch_bounds__(kThreads) void k_dot_part(const float* __restrict__ a, const float* __restrict__ bvec,
                                                        int c, float* __restrict__ part, int N, int rows_per,
                                                        const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float red[kThreads];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  float acc = 0.f;
  if (slot < nrs) {
    const size_t base = (size_t)b * N * c + col;
    for (int row = r0 + slot; row < r1; row += nrs) acc = fmaf(a[base + (size_t)row * c], bvec[base + (size_t)row * c], acc);
  }
  const float tot = block_colsum(slot < nrs ? acc : 0.f, c, nrs, red);
  if (threadIdx.x < c) part[((size_t)b * S + s) * c + col] = tot;
}

int vec_dot_part(const float* a, const float* b, int64_t c, float* part, int64_t B, int64_t N, Split sp, const int* stop,
                 hipStream_t st) {
  if (c < 1 || c > kMaxCols) return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("vec_dot_part", st);
  hipLaunchKernelGGL(k_dot_part, dim3(sp.S, (unsigned)B), dim3(kThreads), 0, st, a, b, (int)c, part, (int)N, sp.rows,
                     stop);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

__global__ __launch_bounds__(kThreads) void k_add_diag(const float* __restrict__ dd, int dd_mode,
                                                        const float* __restrict__ v, float* __restrict__ y, int c,
                                                        int N, int rows_per, const int* __restrict__ stop) {
  if (stop && *stop) return;
  const int s = blockIdx.x, b = blockIdx.y;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  if (slot >= nrs) return;
  const size_t base = (size_t)b * N * c + col;
  const float dc = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;
  for (int row = r0 + slot; row < r1; row += nrs) {
    const float dv = (dd_mode == LO_DIAG_FULL) ? dd[(size_t)b * N + row] : dc;
    y[base + (size_t)row * c] = fmaf(dv, v[base + (size_t)row * c], y[base + (size_t)row * c]);
  }
}

int vec_add_diag(const float* dd, int dd_mode, const float* v, float* y, int64_t c, int64_t B, int64_t N, Split sp,
                 const int* stop, hipStream_t st) {
  if (dd_mode == LO_DIAG_NONE) return LO_OK;
  if (c < 1 || c > kMaxCols) return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("vec_add_diag", st);
  hipLaunchKernelGGL(k_add_diag, dim3(sp.S, (unsigned)B), dim3(kThreads), 0, st, dd, dd_mode, v, y, (int)c, (int)N,
                     sp.rows, stop);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

__global__ __launch_bounds__(kThreads) void k_axpy1(float* __restrict__ y, const float* __restrict__ a, size_t n,
                                                     const int* __restrict__ stop) {
  if (stop && *stop) return;
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    float4 yv = reinterpret_cast<float4*>(y)[i];
    const float4 av = reinterpret_cast<const float4*>(a)[i];
    yv.x += av.x; yv.y += av.y; yv.z += av.z; yv.w += av.w;
    reinterpret_cast<float4*>(y)[i] = yv;
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) y[i] += a[i];
}

// Clears a small control region (control block + hand-off granules of a resident launch) with ONE kernel launch:
// hipMemsetAsync of such a span became five __amd_rocclr_fillBufferAligned launches per solve on ROCm 7
// (profiles/r04/kernel_stats_bench.csv of the first build: 8.5 us of fill kernels in front of every 340 us solve).
// p: 16-byte aligned; bytes: any (the tail is cleared bytewise).
__global__ __launch_bounds__(kThreads) void k_zero_span(uint4* __restrict__ p, size_t n16, unsigned char* __restrict__ tail,
                                                         int ntail) {
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

int zero_span(void* p, size_t bytes, hipStream_t st) {
  if (!bytes) return LO_OK;
  if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) {  // (never on an Arena allocation: the library path of last resort)
    LO_HIP_CHECK(hipMemsetAsync(p, 0, bytes, st));
    return LO_OK;
  }
  const size_t n16 = bytes / 16;
  const int ntail = (int)(bytes - 16 * n16);
  const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((n16 + kThreads - 1) / kThreads, 512));
  hipLaunchKernelGGL(k_zero_span, dim3(grid), dim3(kThreads), 0, st, reinterpret_cast<uint4*>(p), n16,
                     reinterpret_cast<unsigned char*>(p) + 16 * n16, ntail);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

// y += a: the accumulation step of a SumLinearOperator matvec (sum_linear_operator.py:47-51)
int vec_axpy1(float* y, const float* a, size_t n, const int* stop, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<size_t>((n / 4 + kThreads - 1) / kThreads + 1, 16384);
  LO_PROF_BEGIN("vec_axpy1", st);
  hipLaunchKernelGGL(k_axpy1, dim3(grid), dim3(kThreads), 0, st, y, a, n, stop);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // namespace lo
