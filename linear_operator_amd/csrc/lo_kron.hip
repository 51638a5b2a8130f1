// lo_kron.hip -- y = (K1 (x) K2) v  ==  vec(K1 V K2^T), V = v viewed as [n1, n2, c]
// (reference: module-level _matmul of operators/kronecker_product_linear_operator.py:34-45: per factor a
//  view [n_i, -1], a batched matmul, and a transposing reshape).  Two batched strided GEMMs, no
//  transposing copies:
//    W[i1,(j2,col)]  = sum_j1 K1[i1,j1] V[j1,(j2,col)]                (M=n1, K=n1, N=n2*c)
//    Y[i1,i2,col]    = sum_j2 W[i1,j2,col] K2[i2,j2]   per (b,col)    (M=n1, K=n2, N=n2)
// fp32 LDS-tiled GEMM, 64x64 tile, BK=16, 4x4 register micro-tile (round 1: VALU; the MFMA
// v_mfma_f32_32x32x2_f32 version is the next step for this compute-bound operator).
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

struct GemmArgs {
  const float* A;
  const float* Bm;
  float* C;
  int M, N, K;
  int64_t sa_r, sa_c, sb_r, sb_c, sc_r, sc_c;
  // batch z = zo * inner + zi
  int inner;
  int64_t a_zo, a_zi, b_zo, b_zi, c_zo, c_zi;
};

constexpr int BM = 64, BN = 64, BK = 16;

__global__ __launch_bounds__(kThreads) void k_gemm(GemmArgs g, const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  const int z = blockIdx.z;
  const int zo = z / g.inner, zi = z % g.inner;
  const float* A = g.A + zo * g.a_zo + zi * g.a_zi;
  const float* Bm = g.Bm + zo * g.b_zo + zi * g.b_zi;
  float* C = g.C + zo * g.c_zo + zi * g.c_zi;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;  // 16 x 16 threads, 4x4 each
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < g.K; k0 += BK) {
    // stage A tile [BM x BK] and B tile [BK x BN]
    for (int e = threadIdx.x; e < BM * BK; e += kThreads) {
      int mm, kk;
      if (g.sa_c == 1) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
      const int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < g.M && gk < g.K) ? A[gm * g.sa_r + gk * g.sa_c] : 0.f;
    }
    for (int e = threadIdx.x; e < BK * BN; e += kThreads) {
      int nn, kk;
      if (g.sb_c == 1) { nn = e % BN; kk = e / BN; } else { kk = e % BK; nn = e / BK; }
      const int gn = n0 + nn, gk = k0 + kk;
      Bs[kk][nn] = (gn < g.N && gk < g.K) ? Bm[gk * g.sb_r + gn * g.sb_c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(&As[kk][4 * ty]);
      const float4 bq = *reinterpret_cast<const float4*>(&Bs[kk][4 * tx]);
      const float av[4] = {a.x, a.y, a.z, a.w};
      const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + 4 * ty + i;
    if (gm >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + 4 * tx + j;
      if (gn < g.N) C[gm * g.sc_r + gn * g.sc_c] = acc[i][j];
    }
  }
}

static int launch_gemm(const GemmArgs& g, int nbatch, const int* stop, hipStream_t st) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, nbatch);
  LO_PROF_BEGIN("kron_gemm", st);
  hipLaunchKernelGGL(k_gemm, grid, dim3(kThreads), 0, st, g, stop);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

int kron_matvec(const float* K1, const float* K2, const float* v, float* tmp, float* y, int64_t B, int n1, int n2,
                int64_t c, const int* stop, hipStream_t st) {
  const int64_t N = (int64_t)n1 * n2;
  GemmArgs g1;
  g1.A = K1; g1.Bm = v; g1.C = tmp;
  g1.M = n1; g1.K = n1; g1.N = (int)(n2 * c);
  g1.sa_r = n1; g1.sa_c = 1; g1.sb_r = n2 * c; g1.sb_c = 1; g1.sc_r = n2 * c; g1.sc_c = 1;
  g1.inner = 1;
  g1.a_zo = (int64_t)n1 * n1; g1.a_zi = 0; g1.b_zo = N * c; g1.b_zi = 0; g1.c_zo = N * c; g1.c_zi = 0;
  int rc = launch_gemm(g1, (int)B, stop, st);
  if (rc) return rc;
  GemmArgs g2;
  g2.A = tmp; g2.Bm = K2; g2.C = y;
  g2.M = n1; g2.K = n2; g2.N = n2;
  g2.sa_r = n2 * c; g2.sa_c = c; g2.sb_r = 1; g2.sb_c = n2; g2.sc_r = n2 * c; g2.sc_c = c;
  g2.inner = (int)c;
  g2.a_zo = N * c; g2.a_zi = 1; g2.b_zo = (int64_t)n2 * n2; g2.b_zi = 0; g2.c_zo = N * c; g2.c_zi = 1;
  if (B * c > 65535) return LO_ERR_UNSUPPORTED;
  return launch_gemm(g2, (int)(B * c), stop, st);
}

}  // namespace lo
