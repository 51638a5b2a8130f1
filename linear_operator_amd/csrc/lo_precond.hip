// lo_precond.hip -- builds the cached form of the pivoted-Cholesky Woodbury preconditioner
// P = L L^T + D, restating AddedDiagLinearOperator._init_cache* (added_diag_linear_operator.py:144-184).
// The reference takes a thin QR of [L / sqrt(d); I_k] (or [L; sqrt(sigma) I_k]) and keeps
// Q <- Q[:N] / sqrt(d) and logdet = 2 sum log|R_ii| + sum log d_i  (resp. + (N-k) log sigma).
// Only Q Q^T and |R_ii| are ever used, so any factorisation R^T R = G of the k x k Gram
// matrix G = I + W^T W (W = L / sqrt(d)) gives the same operator:  Q = W R^{-1} / sqrt(d).
// Here: G, its Cholesky factor and the triangular inverse are computed in fp64 (k <= 32, tiny) and Q is
// rounded ONCE to fp32 -- at least as accurate as the reference's fp32 Householder QR, no rocSOLVER,
// two streaming passes over L.  For a constant diagonal the 1/sqrt(sigma) is folded into Q so that the
// apply is z = r o dinv - Q (Q^T r) in both cases.
// Output Q is written with the zero-padded row stride ldq = 4 * pow2 >= k that the skinny kernels read.
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int kPbRows = 32;   // rows staged per step
constexpr int kPbMaxK = 32;

// partial Gram matrices: gpart[b,s,k,k] (fp64) = sum_{rows in slice} w w^T ; logd_part[b,s] = sum log d
__global__ __launch_bounds__(kThreads) void k_pb_gram(const float* __restrict__ L, const float* __restrict__ dd,
                                                       int diag_mode, int N, int k, int rows_per,
                                                       double* __restrict__ gpart, double* __restrict__ logd_part) {
  __shared__ float w_s[kPbRows][kPbMaxK + 1];
  __shared__ float sc_s[kPbRows];
  __shared__ double redd[4];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const int npair = k * k;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};  // pairs threadIdx.x + 256*u
  double lacc = 0.0;
  for (int base = r0; base < r1; base += kPbRows) {
    const int nr = min(kPbRows, r1 - base);
    __syncthreads();
    if ((int)threadIdx.x < nr) {  // one fp64 rsqrt / log per ROW, not per element
      double sc = 1.0;
      if (diag_mode == LO_DIAG_FULL) {
        const double dv = (double)dd[(size_t)b * N + base + threadIdx.x];
        sc = 1.0 / sqrt(dv);
        lacc += log(dv);
      }
      sc_s[threadIdx.x] = (float)sc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nr * k; e += kThreads) {
      const int rr = e / k, a = e % k;
      w_s[rr][a] = L[((size_t)b * N + base + rr) * k + a] * sc_s[rr];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pr = threadIdx.x + kThreads * u;
      if (pr < npair) {
        const int a = pr / k, c2 = pr % k;
        double t = acc[u];
        for (int rr = 0; rr < nr; ++rr) t = fma((double)w_s[rr][a], (double)w_s[rr][c2], t);
        acc[u] = t;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int pr = threadIdx.x + kThreads * u;
    if (pr < npair) gpart[((size_t)b * S + s) * npair + pr] = acc[u];
  }
  // block sum of lacc (fp64)
  double v = wave_sum_d(lacc);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) logd_part[b * S + s] = (redd[0] + redd[1]) + (redd[2] + redd[3]);
}

// one wave per member: G = base + sum partials; Cholesky G = Lc Lc^T; M = Lc^{-1}; logdet
__global__ __launch_bounds__(64) void k_pb_chol(const double* __restrict__ gpart, const double* __restrict__ logd_part,
                                                 const float* __restrict__ dd, int diag_mode, int N, int k, int S,
                                                 double* __restrict__ Minv, float* __restrict__ logdet,
                                                 float* __restrict__ dinv_const) {
  __shared__ double G[kPbMaxK][kPbMaxK + 1];
  __shared__ double M[kPbMaxK][kPbMaxK + 1];
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const int npair = k * k;
  const double sigma = (diag_mode == LO_DIAG_CONST) ? (double)dd[b] : 1.0;
  for (int pr = lane; pr < npair; pr += 64) {
    double t = 0.0;
    for (int s = 0; s < S; ++s) t += gpart[((size_t)b * S + s) * npair + pr];
    const int a = pr / k, c2 = pr % k;
    if (a == c2) t += sigma;  // + I (non-constant: W already scaled)  or  + sigma I (constant)
    G[a][c2] = t;
  }
  __syncthreads();
  // right-looking Cholesky, lower factor stored in G's lower triangle
  for (int j = 0; j < k; ++j) {
    if (lane == 0) G[j][j] = sqrt(G[j][j]);
    __syncthreads();
    const double dj = G[j][j];
    for (int i = j + 1 + lane; i < k; i += 64) G[i][j] /= dj;
    __syncthreads();
    for (int e = lane; e < (k - j - 1) * (k - j - 1); e += 64) {
      const int i = j + 1 + e / (k - j - 1), c2 = j + 1 + e % (k - j - 1);
      if (c2 <= i) G[i][c2] -= G[i][j] * G[c2][j];
    }
    __syncthreads();
  }
  // M = Lc^{-1} (lower triangular): column by column forward substitution, one lane per column
  for (int col = lane; col < k; col += 64) {
    for (int i = 0; i < k; ++i) {
      double t = (i == col) ? 1.0 : 0.0;
      for (int a = col; a < i; ++a) t -= G[i][a] * M[a][col];
      M[i][col] = (i < col) ? 0.0 : t / G[i][i];
    }
  }
  __syncthreads();
  for (int pr = lane; pr < npair; pr += 64) Minv[(size_t)b * npair + pr] = M[pr / k][pr % k];
  if (lane == 0) {
    double ld = 0.0;
    for (int j = 0; j < k; ++j) ld += log(fabs(G[j][j]));
    ld *= 2.0;                                                         // 2 sum log|R_ii|          :168,:181
    if (diag_mode == LO_DIAG_CONST) {
      ld += (double)(N - k) * log(sigma);                              // + (n-k) log sigma        :171
      dinv_const[b] = (float)(1.0 / sigma);
    } else {
      double t = 0.0;
      for (int s = 0; s < S; ++s) t += logd_part[b * S + s];
      ld += t;                                                         // - sum log(1/d)           :183
    }
    logdet[b] = (float)ld;
  }
}

// Q[row, j] = scale_row * sum_{a<=j} M[j][a] w[a]
template <int KM>
__global__ __launch_bounds__(kThreads) void k_pb_q(const float* __restrict__ L, const float* __restrict__ dd,
                                                    int diag_mode, int N, int k, int ldq, int rows_per,
                                                    const double* __restrict__ Minv, float* __restrict__ Q,
                                                    float* __restrict__ dinv) {
  __shared__ double M[kPbMaxK][kPbMaxK + 1];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  for (int pr = threadIdx.x; pr < k * k; pr += kThreads) M[pr / k][pr % k] = Minv[(size_t)b * k * k + pr];
  __syncthreads();
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const double inv_sqrt_sigma = (diag_mode == LO_DIAG_CONST) ? 1.0 / sqrt((double)dd[b]) : 1.0;
  for (int row = r0 + threadIdx.x; row < r1; row += kThreads) {
    double w[KM];
    double sc = inv_sqrt_sigma;
    if (diag_mode == LO_DIAG_FULL) {
      const double dv = (double)dd[(size_t)b * N + row];
      sc = 1.0 / sqrt(dv);
      dinv[(size_t)b * N + row] = (float)(1.0 / dv);
    }
    const float* lr = L + ((size_t)b * N + row) * k;
#pragma unroll
    for (int a = 0; a < KM; ++a) w[a] = (a < k) ? (double)lr[a] * ((diag_mode == LO_DIAG_FULL) ? sc : 1.0) : 0.0;
    float* qr = Q + ((size_t)b * N + row) * ldq;
#pragma unroll
    for (int j = 0; j < KM; ++j) {
      if (j < k) {
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < KM; ++a)
          if (a <= j && a < k) t = fma(M[j][a], w[a], t);
        qr[j] = (float)(t * sc);
      }
    }
    for (int j = k; j < ldq; ++j) qr[j] = 0.f;
  }
}

static int padded_k(int k) {
  int rq = (k + 3) / 4, p = 1;
  while (p < rq) p <<= 1;
  return 4 * p;
}

}  // namespace lo

using namespace lo;

extern "C" {

size_t lo_precond_build_workspace_bytes(int64_t B, int64_t N, int32_t k) {
  Split sp = choose_split(B, N, 256);
  Arena ar(nullptr, 0);
  ar.take<double>((size_t)B * sp.S * k * k);
  ar.take<double>((size_t)B * sp.S);
  ar.take<double>((size_t)B * k * k);
  return ar.off + 1024;
}

// Q must hold [B, N, ldq] floats with ldq = 4 * pow2ceil(ceil(k/4)); dinv [B,N] (FULL) or [B] (CONST).
int lo_precond_build_f32(const float* L, const float* d, int32_t diag_mode, int64_t B, int64_t N, int32_t k, float* Q,
                         float* dinv, float* logdet_p, void* ws, size_t ws_bytes, void* stream) {
  if (!L || !d || !Q || !dinv || !logdet_p || !ws) return LO_ERR_BADARG;
  if (diag_mode != LO_DIAG_FULL && diag_mode != LO_DIAG_CONST) return LO_ERR_BADARG;
  if (k < 1 || k > kPbMaxK) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Split sp = choose_split(B, N, 256);
  Arena ar(ws, ws_bytes);
  double* gpart = ar.take<double>((size_t)B * sp.S * k * k);
  double* logd = ar.take<double>((size_t)B * sp.S);
  double* Minv = ar.take<double>((size_t)B * k * k);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const int ldq = padded_k(k);
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  LO_PROF_BEGIN("pb_gram", st);
  hipLaunchKernelGGL(k_pb_gram, grid, block, 0, st, L, d, diag_mode, (int)N, (int)k, sp.rows, gpart, logd);
  LO_PROF_END(st);
  hipLaunchKernelGGL(k_pb_chol, dim3((unsigned)B), dim3(64), 0, st, gpart, logd, d, diag_mode, (int)N, (int)k, sp.S,
                     Minv, logdet_p, dinv);
  LO_PROF_BEGIN("pb_q", st);
  if (k <= 16)
    hipLaunchKernelGGL((k_pb_q<16>), grid, block, 0, st, L, d, diag_mode, (int)N, (int)k, ldq, sp.rows, Minv, Q, dinv);
  else
    hipLaunchKernelGGL((k_pb_q<32>), grid, block, 0, st, L, d, diag_mode, (int)N, (int)k, ldq, sp.rows, Minv, Q, dinv);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // extern "C"
