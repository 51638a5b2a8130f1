// lo_dense.hip -- y = K v + d o v for a batch of dense symmetric operators K [B,N,N]
// (reference: AddedDiagLinearOperator._matmul added_diag_linear_operator.py:72-76 over
//  DenseLinearOperator._matmul dense_linear_operator.py:60-64).  HBM-bound: K is streamed once per
// column chunk, 16 B per lane; v (N*c floats) is re-read through L1/L2.
// VALU kernel: one wave owns 4 rows at a time (register tile 4 rows x CT columns), lanes stride over the
// K dimension; wave butterfly at the end of each row group.  Fused epilogue: + d o v, and the CG inner
// product partial sum_rows v o y.
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int kDenseRB = 4;  // rows per wave pass

template <int CT>
__global__ __launch_bounds__(kThreads) void k_dense_mv(const float* __restrict__ K, const float* __restrict__ dd,
                                                        int dd_mode, const float* __restrict__ v, int ldv, int c,
                                                        float* __restrict__ y, float* __restrict__ dot_part, int ldd,
                                                        int N, int rows_per_wg, const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float red[kThreads];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows_per_wave = rows_per_wg / 4;
  const int wr0 = s * rows_per_wg + wave * rows_per_wave;
  const int wr1 = min(N, wr0 + rows_per_wave);
  const float* Kb = K + (size_t)b * N * N;
  const float* vb = v + (size_t)b * N * ldv;
  float* yb = y + (size_t)b * N * ldv;
  const float dc = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;
  const int N4 = N & ~3;

  float dacc[CT];
#pragma unroll
  for (int k = 0; k < CT; ++k) dacc[k] = 0.f;

  for (int row = wr0; row < wr1; row += kDenseRB) {
    float acc[kDenseRB][CT];
#pragma unroll
    for (int u = 0; u < kDenseRB; ++u)
#pragma unroll
      for (int k = 0; k < CT; ++k) acc[u][k] = 0.f;
    const float* kr[kDenseRB];
#pragma unroll
    for (int u = 0; u < kDenseRB; ++u) kr[u] = Kb + (size_t)min(row + u, N - 1) * N;

    if ((N & 3) == 0) {
      for (int j = 4 * lane; j < N4; j += 256) {
        float4 a[kDenseRB];
#pragma unroll
        for (int u = 0; u < kDenseRB; ++u) a[u] = *reinterpret_cast<const float4*>(kr[u] + j);
        float vv[4][CT];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int k = 0; k < CT; ++k) vv[jj][k] = (k < c) ? vb[(size_t)(j + jj) * ldv + k] : 0.f;
#pragma unroll
        for (int u = 0; u < kDenseRB; ++u)
#pragma unroll
          for (int k = 0; k < CT; ++k) {
            float t = acc[u][k];
            t = fmaf(a[u].x, vv[0][k], t);
            t = fmaf(a[u].y, vv[1][k], t);
            t = fmaf(a[u].z, vv[2][k], t);
            t = fmaf(a[u].w, vv[3][k], t);
            acc[u][k] = t;
          }
      }
    } else {
      for (int j = lane; j < N; j += 64) {
#pragma unroll
        for (int u = 0; u < kDenseRB; ++u) {
          const float a = kr[u][j];
#pragma unroll
          for (int k = 0; k < CT; ++k) acc[u][k] = fmaf(a, (k < c) ? vb[(size_t)j * ldv + k] : 0.f, acc[u][k]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kDenseRB; ++u)
#pragma unroll
      for (int k = 0; k < CT; ++k) acc[u][k] = wave_sum(acc[u][k]);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < kDenseRB; ++u) {
        const int rr = row + u;
        if (rr < wr1) {
          const float dv = (dd_mode == LO_DIAG_FULL) ? dd[(size_t)b * N + rr] : dc;
#pragma unroll
          for (int k = 0; k < CT; ++k) {
            if (k < c) {
              const float vin = vb[(size_t)rr * ldv + k];
              const float yv = fmaf(dv, vin, acc[u][k]);
              yb[(size_t)rr * ldv + k] = yv;
              dacc[k] = fmaf(vin, yv, dacc[k]);
            }
          }
        }
      }
    }
  }
  if (dot_part) {
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      if (k < c) {
        const float tot = block_sum256(lane == 0 ? dacc[k] : 0.f, red);
        if (threadIdx.x == 0) dot_part[((size_t)b * S + s) * ldd + k] = tot;
      }
    }
  }
}

int dense_rows_per_wg(int64_t B, int64_t N) {
  // >= ~1024 workgroups when possible, 16..128 rows per workgroup (multiple of 16)
  int64_t rows = 128;
  while (rows > 16 && B * ((N + rows - 1) / rows) < 1024) rows /= 2;
  return (int)rows;
}

int dense_S_dot(int64_t B, int64_t N, int64_t c) {
  if (dense_mfma_ok(N, c)) return dense_mfma_tiles(N);
  const int rows = dense_rows_per_wg(B, N);
  return (int)((N + rows - 1) / rows);
}

int dense_matvec(const float* K, const float* d, int dd_mode, const float* v, float* y, float* dot_part, int64_t B,
                 int64_t N, int64_t c, int rows_per_wg, float* ypart, const int* stop, hipStream_t st) {
  if (c < 1) return LO_ERR_BADARG;
  if (dense_mfma_ok(N, c)) return dense_matvec_mfma(K, d, dd_mode, v, y, dot_part, B, N, c, ypart, stop, st);
  const int S = (int)((N + rows_per_wg - 1) / rows_per_wg);
  dim3 grid(S, (unsigned)B), block(kThreads);
  for (int64_t c0 = 0; c0 < c; c0 += 4) {
    const int cn = (int)std::min<int64_t>(4, c - c0);
    const float* vp = v + c0;
    float* yp = y + c0;
    float* dp = dot_part ? dot_part + c0 : nullptr;
#define LO_DM(CT)                                                                                              \
  hipLaunchKernelGGL((k_dense_mv<CT>), grid, block, 0, st, K, d, dd_mode, vp, (int)c, cn, yp, dp, (int)c, (int)N, \
                     rows_per_wg, stop)
    LO_PROF_BEGIN("dense_mv", st);
    if (cn == 1) LO_DM(1);
    else if (cn == 2) LO_DM(2);
    else LO_DM(4);
#undef LO_DM
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

}  // namespace lo
