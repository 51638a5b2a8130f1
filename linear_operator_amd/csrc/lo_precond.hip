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
#include <stdint.h>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int kPbRows = 32;   // rows staged per step
constexpr int kPbMaxK = 32;

// partial Gram matrices: gpart[b,s,k,k] (fp64) = sum_{rows in slice} w w^T ; logd_part[b,s] = sum log d
// L element (member b, row i, column a) = L[b * ls.member + i * ls.row + a * ls.col]: [B,N,k] (reference layout)
// or the [B, max_rank, N] rows the pivoted-Cholesky kernels write (no transposed copy needed)
struct LStride {
  int64_t member, row, col;
};

__global__ __launch_bounds__(kThreads) void k_pb_gram(const float* __restrict__ L, LStride ls,
                                                       const float* __restrict__ dd, int diag_mode, int N, int k,
                                                       int rows_per, double* __restrict__ gpart,
                                                       double* __restrict__ logd_part) {
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
    const float* Lb = L + (size_t)b * ls.member;
    if (ls.row == 1) {  // rows layout: consecutive threads take consecutive rows of one column
      for (int e = threadIdx.x; e < kPbRows * k; e += kThreads) {
        const int a = e / kPbRows, rr = e % kPbRows;
        if (rr < nr) w_s[rr][a] = Lb[(size_t)a * ls.col + base + rr] * sc_s[rr];
      }
    } else {
      for (int e = threadIdx.x; e < nr * k; e += kThreads) {
        const int rr = e / k, a = e % k;
        w_s[rr][a] = Lb[(size_t)(base + rr) * ls.row + (size_t)a * ls.col] * sc_s[rr];
      }
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
__global__ __launch_bounds__(kThreads) void k_pb_q(const float* __restrict__ L, LStride ls,
                                                    const float* __restrict__ dd, int diag_mode, int N, int k,
                                                    int ldq, int rows_per,
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
    const float* lr = L + (size_t)b * ls.member + (size_t)row * ls.row;
#pragma unroll
    for (int a = 0; a < KM; ++a)
      w[a] = (a < k) ? (double)lr[(size_t)a * ls.col] * ((diag_mode == LO_DIAG_FULL) ? sc : 1.0) : 0.0;
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

// ---- fp64 matrix-core versions for the rows layout (ls.row == 1) and k <= 16 -------------------------------
// v_mfma_f64_16x16x4_f64 (layout probed on gfx950, tools/probe/mfma_f64_layout.hip): lane l supplies
// A[i = l % 16][kk = l / 16] and B[kk = l / 16][j = l % 16]; result register r of lane l is D[4 r + l / 16][l % 16].
typedef double f64x4 __attribute__((ext_vector_type(4)));

// per-row quantities of the non-constant diagonal, once: sc = 1/sqrt(d) (fp64, rounded to fp32), dinv = 1/d,
// logd_part[b,s] = sum log d (fp64)
__global__ __launch_bounds__(kThreads) void k_pb_scale(const float* __restrict__ dd, int N, int rows_per,
                                                        float* __restrict__ sc, float* __restrict__ dinv,
                                                        double* __restrict__ logd_part) {
  __shared__ double redd[4];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  double lacc = 0.0;
  for (int row = r0 + threadIdx.x; row < r1; row += kThreads) {
    const double dv = (double)dd[(size_t)b * N + row];
    sc[(size_t)b * N + row] = (float)(1.0 / sqrt(dv));
    dinv[(size_t)b * N + row] = (float)(1.0 / dv);
    lacc += log(dv);
  }
  const double v = wave_sum_d(lacc);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) logd_part[b * S + s] = (redd[0] + redd[1]) + (redd[2] + redd[3]);
}

// G partial = W^T W over the slice, W = L o sc (fp32 product, as k_pb_gram), accumulated in fp64 on the matrix
// cores: one MFMA consumes 4 rows, A and B operand of a lane are the SAME value w[row][a = l % 16].
// A wave takes 32-row chunks: lane (a, kk) loads rows chunk + 8 kk .. + 7 of column a (two 16-byte loads).
__global__ __launch_bounds__(kThreads) void k_pb_gram_mfma(const float* __restrict__ L, LStride ls,
                                                            const float* __restrict__ sc, int N, int k, int rows_per,
                                                            double* __restrict__ gpart) {
  __shared__ double red[4][64][4];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int a = l & 15, kk = l >> 4;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float* col = L + (size_t)b * ls.member + (size_t)a * ls.col;
  const float* scb = sc ? sc + (size_t)b * N : nullptr;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int cb = r0 + 32 * wave; cb < r1; cb += 128) {
    const int rb = cb + 8 * kk;
    float w[8];
    if (cb + 32 <= r1) {  // wave-uniform: whole chunk inside the slice (rows are multiples of 4 -> aligned)
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
      if (a < k) {
        x0 = *reinterpret_cast<const float4*>(col + rb);
        x1 = *reinterpret_cast<const float4*>(col + rb + 4);
      }
      float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0;
      if (scb) {
        s0 = *reinterpret_cast<const float4*>(scb + rb);
        s1 = *reinterpret_cast<const float4*>(scb + rb + 4);
      }
      w[0] = x0.x * s0.x; w[1] = x0.y * s0.y; w[2] = x0.z * s0.z; w[3] = x0.w * s0.w;
      w[4] = x1.x * s1.x; w[5] = x1.y * s1.y; w[6] = x1.z * s1.z; w[7] = x1.w * s1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = rb + e;
        w[e] = (a < k && row < r1) ? col[row] * (scb ? scb[row] : 1.f) : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w[e], (double)w[e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][l][r] = acc[r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ga = 4 * r + kk, gc = a;  // D[4 r + l / 16][l % 16]
      if (ga < k && gc < k)
        gpart[((size_t)b * S + s) * k * k + ga * k + gc] = (red[0][l][r] + red[1][l][r]) + (red[2][l][r] + red[3][l][r]);
    }
  }
}

// Q tile = W M^T on the matrix cores: 64-row chunks per wave as 4 interleaved 16-row tiles (tile t holds the rows
// chunk + 4 i + t, so that lane (i, kk) fetches its 4 tiles' operands of column a = 4 s + kk with ONE 16-byte load).
__global__ __launch_bounds__(kThreads) void k_pb_q_mfma(const float* __restrict__ L, LStride ls,
                                                         const float* __restrict__ sc, const float* __restrict__ dd,
                                                         int diag_mode, int N, int k, int ldq, int rows_per,
                                                         const double* __restrict__ Minv, float* __restrict__ Q) {
  __shared__ __attribute__((aligned(16))) float q_tile[kThreads / 64][64 * 20];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int i = l & 15, kk = l >> 4;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  double mb[4];  // B operand of step s: M^T[4 s + kk][j = i] = M[i][4 s + kk]
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int a = 4 * st + kk;
    mb[st] = (i < k && a < k) ? Minv[(size_t)b * k * k + i * k + a] : 0.0;
  }
  const float* Lb = L + (size_t)b * ls.member;
  const float* scb = sc ? sc + (size_t)b * N : nullptr;
  const double cs = (diag_mode == LO_DIAG_CONST) ? 1.0 / sqrt((double)dd[b]) : 1.0;
  float* Qb = Q + (size_t)b * N * ldq;
  for (int cb = r0 + 64 * wave; cb < r1; cb += 256) {
    const bool full = cb + 64 <= r1;
    const int rb = cb + 4 * i;
    float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (scb) {
      if (full) sv = *reinterpret_cast<const float4*>(scb + rb);
      else {
        sv.x = rb < r1 ? scb[rb] : 0.f; sv.y = rb + 1 < r1 ? scb[rb + 1] : 0.f;
        sv.z = rb + 2 < r1 ? scb[rb + 2] : 0.f; sv.w = rb + 3 < r1 ? scb[rb + 3] : 0.f;
      }
    }
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int a = 4 * st + kk;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a < k) {
        const float* cp = Lb + (size_t)a * ls.col + rb;
        if (full) x = *reinterpret_cast<const float4*>(cp);
        else {
          x.x = rb < r1 ? cp[0] : 0.f; x.y = rb + 1 < r1 ? cp[1] : 0.f;
          x.z = rb + 2 < r1 ? cp[2] : 0.f; x.w = rb + 3 < r1 ? cp[3] : 0.f;
        }
      }
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(x.x * sv.x), mb[st], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(x.y * sv.y), mb[st], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(x.z * sv.z), mb[st], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(x.w * sv.w), mb[st], acc[3], 0, 0, 0);
    }
    // acc[t][r] = Qtile_t[4 r + kk][j = i] -> row cb + 4 (4 r + kk) + t, column j
    if (full && ldq == 16) {
      // the wave's 64 rows x 16 columns are 4 KB contiguous in Q: through the wave's LDS tile (row stride 20 floats: the
      // four row groups kk of a store land in different banks, rows stay 16-byte aligned) and out as four 16-byte stores
      // per lane -- the direct stores below write 64-byte pieces of four rows per instruction (2.6 TB/s for the kernel)
      float* tile = q_tile[wave];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 16 * r + 4 * kk;
        const float4 s4 = scb ? *reinterpret_cast<const float4*>(scb + cb + lrow) : make_float4(1.f, 1.f, 1.f, 1.f);
        const double rs4[4] = {scb ? (double)s4.x : cs, scb ? (double)s4.y : cs, scb ? (double)s4.z : cs,
                               scb ? (double)s4.w : cs};
#pragma unroll
        for (int t = 0; t < 4; ++t) tile[(lrow + t) * 20 + i] = (float)(acc[t][r] * rs4[t]);
      }
      __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave execute in order; this pins the compiler's order)
      float4* dst = reinterpret_cast<float4*>(Qb + (size_t)cb * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = l + 64 * q;  // float4 index inside the 64 x 16 tile: row f / 4, columns 4 (f % 4) ..
        dst[f] = *reinterpret_cast<const float4*>(tile + (f >> 2) * 20 + 4 * (f & 3));
      }
      __builtin_amdgcn_wave_barrier();  // the next chunk reuses the tile
    } else if (i < ldq) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rowb = cb + 16 * r + 4 * kk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = rowb + t;
          if (row < r1) {
            const double rs = scb ? (double)scb[row] : cs;
            Qb[(size_t)row * ldq + i] = (float)(acc[t][r] * rs);
          }
        }
      }
    }
  }
}

// ---- the same for the reference layout L [B, N, k] (k contiguous), k <= 32: NA = number of 16-column halves --------
// Gram: lane (a, kk) feeds w[row = chunk + 4 e + kk][a] (and [a + 16]) to MFMA e: a wave instruction reads 4 rows x
// 64 contiguous bytes.  Blocks G00, G01, G11 are accumulated, G10 = G01^T is written on the way out.
template <int NA>
__global__ __launch_bounds__(kThreads) void k_pb_gram_mfma_nk(const float* __restrict__ L, LStride ls,
                                                               const float* __restrict__ sc, int N, int k,
                                                               int rows_per, double* __restrict__ gpart) {
  constexpr int NB = (NA == 2) ? 3 : 1;
  __shared__ double red[4][64][4 * NB];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int a = l & 15, kk = l >> 4;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float* Lb = L + (size_t)b * ls.member;
  const float* scb = sc ? sc + (size_t)b * N : nullptr;
  f64x4 acc00 = {0.0, 0.0, 0.0, 0.0}, acc01 = acc00, acc11 = acc00;
  for (int cb = r0 + 32 * wave; cb < r1; cb += 128) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = cb + 4 * e + kk;
      float w0 = 0.f, w1 = 0.f;
      if (row < r1) {
        const float sv = scb ? scb[row] : 1.f;
        const float* lr = Lb + (size_t)row * ls.row;
        if (a < k) w0 = lr[a] * sv;
        if (NA == 2 && a + 16 < k) w1 = lr[a + 16] * sv;
      }
      acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w0, (double)w0, acc00, 0, 0, 0);
      if (NA == 2) {
        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w0, (double)w1, acc01, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)w1, (double)w1, acc11, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave][l][r] = acc00[r];
    if (NA == 2) {
      red[wave][l][4 + r] = acc01[r];
      red[wave][l][8 + r] = acc11[r];
    }
  }
  __syncthreads();
  if (wave == 0) {
    double* gp = gpart + ((size_t)b * S + s) * k * k;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 4 * r + kk, gj = a;  // D[4 r + l / 16][l % 16]
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        const double v = (red[0][l][4 * blk + r] + red[1][l][4 * blk + r]) + (red[2][l][4 * blk + r] + red[3][l][4 * blk + r]);
        const int i = gi + (blk == 2 ? 16 : 0), j = gj + (blk >= 1 ? 16 : 0);
        if (i < k && j < k) {
          gp[i * k + j] = v;
          if (blk == 1) gp[j * k + i] = v;  // G10 = G01^T
        }
      }
    }
  }
}

// Q = (W M^T) o scale: lane (i, kk) holds w[row][a = KA kk + st] (KA = 4 NA consecutive columns: 16-byte loads) and
// B operand M[j][a]; 64-row chunks per wave as 4 tiles of 16 consecutive rows.
template <int NA>
__global__ __launch_bounds__(kThreads) void k_pb_q_mfma_nk(const float* __restrict__ L, LStride ls,
                                                            const float* __restrict__ sc, const float* __restrict__ dd,
                                                            int diag_mode, int N, int k, int ldq, int rows_per,
                                                            const double* __restrict__ Minv, float* __restrict__ Q) {
  constexpr int KA = 4 * NA;
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int i = l & 15, kk = l >> 4;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  double mb[KA][NA];
#pragma unroll
  for (int st = 0; st < KA; ++st)
#pragma unroll
    for (int hj = 0; hj < NA; ++hj) {
      const int a = KA * kk + st, j = i + 16 * hj;
      mb[st][hj] = (j < k && a < k) ? Minv[(size_t)b * k * k + j * k + a] : 0.0;
    }
  const float* Lb = L + (size_t)b * ls.member;
  const float* scb = sc ? sc + (size_t)b * N : nullptr;
  const double cs = (diag_mode == LO_DIAG_CONST) ? 1.0 / sqrt((double)dd[b]) : 1.0;
  float* Qb = Q + (size_t)b * N * ldq;
  const bool vec_ok = (ls.row % 4) == 0 && (ls.member % 4) == 0 && ((uintptr_t)L % 16) == 0 && k >= KA * 4;  // 16-byte rows
  for (int cb = r0 + 64 * wave; cb < r1; cb += 256) {
    f64x4 acc[4][NA];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int hj = 0; hj < NA; ++hj) acc[t][hj] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = cb + 16 * t + i;
      float x[KA];
#pragma unroll
      for (int st = 0; st < KA; ++st) x[st] = 0.f;
      float sv = 0.f;
      if (row < r1) {
        sv = scb ? scb[row] : 1.f;
        const float* lr = Lb + (size_t)row * ls.row + KA * kk;
        if (vec_ok) {
#pragma unroll
          for (int q = 0; q < NA; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(lr + 4 * q);
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int st = 0; st < KA; ++st)
            if (KA * kk + st < k) x[st] = lr[st];
        }
      }
#pragma unroll
      for (int st = 0; st < KA; ++st)
#pragma unroll
        for (int hj = 0; hj < NA; ++hj)
          acc[t][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(x[st] * sv), mb[st][hj], acc[t][hj], 0, 0, 0);
    }
    // acc[t][hj][r] = Qtile_t[4 r + kk][j = i + 16 hj]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = cb + 16 * t + 4 * r + kk;
        if (row < r1) {
          const double rs = scb ? (double)scb[row] : cs;
#pragma unroll
          for (int hj = 0; hj < NA; ++hj) {
            const int j = i + 16 * hj;
            if (j < ldq) Qb[(size_t)row * ldq + j] = (float)(acc[t][hj][r] * rs);
          }
        }
      }
  }
}

static int padded_k(int k) {
  int rq = (k + 3) / 4, p = 1;
  while (p < rq) p <<= 1;
  return 4 * p;
}

// Gram partials of W = C o sc for a ROOT C [B, N, R] (R <= 32, rows contiguous): the rows are staged through LDS with
// coalesced 16-byte loads (the per-lane 4-byte loads of k_pb_gram_mfma_nk reach 1.7 TB/s only), then fed to the fp64
// matrix cores as in k_pb_gram_mfma_nk.  gpart [B, S, R, R].
template <int NA>
__global__ __launch_bounds__(kThreads) void k_pb_gram_root(const float* __restrict__ C, const float* __restrict__ sc,
                                                            int N, int R, int rows_per, double* __restrict__ gpart) {
  constexpr int NB = (NA == 2) ? 3 : 1;
  constexpr int TR = 128;                 // rows per tile: 32 per wave
  constexpr int LD = 33;                  // tile row stride (conflict-free column reads)
  __shared__ float tile[2][TR * LD];
  __shared__ double red[4][64][4 * NB];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int a = l & 15, kk = l >> 4;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float* Cb = C + (size_t)b * N * R;
  const float* scb = sc ? sc + (size_t)b * N : nullptr;
  const int RQ = R >> 2;                  // 16-byte pieces per row (R % 4 == 0 on this path)
  // register prefetch: the next tile's 16-byte pieces (4 per thread at R = 32) are in flight while the matrix cores work
  // on the current tile, and go to LDS afterwards
  constexpr int PPT = TR * 8 / kThreads;  // pieces per thread at the widest root (R = 32)
  float4 pv[PPT];
  float ps[PPT];
  auto issue = [&](int base) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = i * kThreads + threadIdx.x;
      const int row = e / RQ, q = e % RQ;
      pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      ps[i] = 1.f;
      if (e < TR * RQ && base + row < r1) {
        pv[i] = *reinterpret_cast<const float4*>(Cb + (size_t)(base + row) * R + 4 * q);
        if (scb) ps[i] = scb[base + row];
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int e = i * kThreads + threadIdx.x;
      if (e < TR * RQ) {
        const int row = e / RQ, q = e % RQ;
        float* t = &tile[buf][row * LD + 4 * q];
        t[0] = pv[i].x * ps[i]; t[1] = pv[i].y * ps[i]; t[2] = pv[i].z * ps[i]; t[3] = pv[i].w * ps[i];
      }
    }
  };
  // fp32 matrix cores inside a tile (32 rows per wave: the partial of a 32-term sum carries the rounding of the final
  // fp32 result anyway), fp64 across tiles (8192 rows = 64 tiles per wave slice)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  double acc00[4] = {0.0, 0.0, 0.0, 0.0}, acc01[4] = {0.0, 0.0, 0.0, 0.0}, acc11[4] = {0.0, 0.0, 0.0, 0.0};
  int buf = 0;
  issue(r0);
  commit(0);
  __syncthreads();
  for (int base = r0; base < r1; base += TR) {
    const bool more = base + TR < r1;
    if (more) issue(base + TR);
    f32x4 t00 = {0.f, 0.f, 0.f, 0.f}, t01 = t00, t11 = t00;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = 32 * wave + 4 * e + kk;
      const float w0 = (a < R) ? tile[buf][row * LD + a] : 0.f;
      t00 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, w0, t00, 0, 0, 0);
      if (NA == 2) {
        const float w1 = (a + 16 < R) ? tile[buf][row * LD + a + 16] : 0.f;
        t01 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, w1, t01, 0, 0, 0);
        t11 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, w1, t11, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc00[r] += (double)t00[r];
      if (NA == 2) {
        acc01[r] += (double)t01[r];
        acc11[r] += (double)t11[r];
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave][l][r] = acc00[r];
    if (NA == 2) {
      red[wave][l][4 + r] = acc01[r];
      red[wave][l][8 + r] = acc11[r];
    }
  }
  __syncthreads();
  if (wave == 0) {
    double* gp = gpart + ((size_t)b * S + s) * R * R;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 4 * kk + r, gj = a;  // fp32 MFMA result layout: D[4 (l / 16) + r][l % 16]
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        const double v = (red[0][l][4 * blk + r] + red[1][l][4 * blk + r]) + (red[2][l][4 * blk + r] + red[3][l][4 * blk + r]);
        const int i = gi + (blk == 2 ? 16 : 0), j = gj + (blk >= 1 ? 16 : 0);
        if (i < R && j < R) {
          gp[i * R + j] = v;
          if (blk == 1) gp[j * R + i] = v;  // G10 = G01^T
        }
      }
    }
  }
}

// ---- root form (see lo_amd.h: lo_precond_desc.F / EF / E) ------------------------------------------------------------
// One wave per member, fp64 in LDS: E from the Gram partials of W = C / sqrt(d) (or C^T C / sigma), the recurrence for
// M on the pivot rows, G = I + M^T E M, its Cholesky factor, F = M G^-1 M^T = (M Lg^-T)(M Lg^-T)^T, EF, logdet.
// (256 threads: the kernel is a chain of global-load latencies -- pivots -> pivot rows of C and L, S partial Gram
// matrices -- and small fp64 loops; with one wave every lane walked 8 - 16 dependent loads one after the other: 74 us)
__global__ __launch_bounds__(kThreads) void k_pb_rootform(const double* __restrict__ gpart, const double* __restrict__ logd_part,
                                                     const float* __restrict__ C, int R, const float* __restrict__ dd,
                                                     int diag_mode, const float* __restrict__ L, LStride ls,
                                                     const long long* __restrict__ perm, int N, int k, int S, int ld,
                                                     float* __restrict__ F, float* __restrict__ EF,
                                                     float* __restrict__ Eo, float* __restrict__ logdet,
                                                     float* __restrict__ dinv_const,
                                                     const float* __restrict__ Cpiv = nullptr,
                                                     float* __restrict__ kappa = nullptr,
                                                     const double* __restrict__ gpart2 = nullptr,
                                                     double* __restrict__ RS = nullptr) {
  __shared__ double E[kPbMaxK][kPbMaxK + 1];
  __shared__ double M[kPbMaxK][kPbMaxK + 1];   // [R][k]
  __shared__ double T[kPbMaxK][kPbMaxK + 1];   // scratch: E M, then Y = M Lg^-T
  __shared__ double G[kPbMaxK][kPbMaxK + 1];
  __shared__ double Fm[kPbMaxK][kPbMaxK + 1];
  __shared__ long long piv[kPbMaxK];
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  constexpr int NT = kThreads;
  const double sigma = (diag_mode == LO_DIAG_CONST) ? (double)dd[b] : 1.0;
  if (lane < k) piv[lane] = perm[(size_t)b * N + lane];
  for (int pr = lane; pr < R * R; pr += NT) {
    double t = 0.0;
    for (int s = 0; s < S; ++s) t += gpart[((size_t)b * S + s) * R * R + pr];
    E[pr / R][pr % R] = t / sigma;  // (FULL: the rows were scaled by 1/sqrt(d); CONST: C^T C / sigma)
  }
  // M[:, j] = (C[pi_j, :]^T - sum_{i<j} M[:, i] L[pi_j, i]) / L[pi_j, j]: the pivot rows of C and the pivot entries of
  // L are fetched first (all loads in flight together: T <- C[pi_j, :], Fm <- L[pi_j, i]), the recurrence runs from LDS
  const float* Cb = C + (size_t)b * N * R;
  const float* Lb = L + (size_t)b * ls.member;
  __syncthreads();
  for (int pr = lane; pr < k * R; pr += NT) {
    const int j = pr / R, a = pr % R;
    // (Cpiv: the k pivot rows handed over explicitly, [B, R, R] -- the Kronecker root form has no tall C in memory)
    T[j][a] = Cpiv ? (double)Cpiv[((size_t)b * R + j) * R + a] : (double)Cb[(size_t)piv[j] * R + a];
  }
  for (int pr = lane; pr < k * k; pr += NT) {
    const int j = pr / k, i = pr % k;
    if (i <= j) Fm[j][i] = (double)Lb[(size_t)piv[j] * ls.row + (size_t)i * ls.col];
  }
  __syncthreads();
  if (lane < R) {  // (row `lane` of M depends on its own earlier entries only: no barrier inside the recurrence)
    for (int j = 0; j < k; ++j) {
      double col = T[j][lane];
      for (int i = 0; i < j; ++i) col -= M[lane][i] * Fm[j][i];
      M[lane][j] = col / Fm[j][j];
    }
  }
  __syncthreads();
  for (int pr = lane; pr < R * k; pr += NT) {  // T = E M  [R][k]
    const int a = pr / k, j = pr % k;
    double t = 0.0;
    for (int c2 = 0; c2 < R; ++c2) t += E[a][c2] * M[c2][j];
    T[a][j] = t;
  }
  __syncthreads();
  for (int pr = lane; pr < k * k; pr += NT) {  // G = I + M^T T
    const int i = pr / k, j = pr % k;
    double t = (i == j) ? 1.0 : 0.0;
    for (int a = 0; a < R; ++a) t += M[a][i] * T[a][j];
    G[i][j] = t;
  }
  __syncthreads();
  for (int j = 0; j < k; ++j) {  // right-looking Cholesky, lower factor in G's lower triangle
    if (lane == 0) G[j][j] = sqrt(G[j][j]);
    __syncthreads();
    const double dj = G[j][j];
    for (int i = j + 1 + lane; i < k; i += NT) G[i][j] /= dj;
    __syncthreads();
    for (int e = lane; e < (k - j - 1) * (k - j - 1); e += NT) {
      const int i = j + 1 + e / (k - j - 1), c2 = j + 1 + e % (k - j - 1);
      if (c2 <= i) G[i][c2] -= G[i][j] * G[c2][j];
    }
    __syncthreads();
  }
  // Y = M Lg^-T: row a of Y solves Lg y = M[a, :]^T (forward substitution), one lane per row
  if (lane < R) {
    for (int j = 0; j < k; ++j) {
      double t = M[lane][j];
      for (int i = 0; i < j; ++i) t -= G[j][i] * T[lane][i];
      T[lane][j] = t / G[j][j];
    }
  }
  __syncthreads();
  for (int pr = lane; pr < R * R; pr += NT) {  // F = Y Y^T
    const int a = pr / R, c2 = pr % R;
    double t = 0.0;
    for (int j = 0; j < k; ++j) t += T[a][j] * T[c2][j];
    Fm[a][c2] = t;
  }
  __syncthreads();
  float* Fb = F + (size_t)b * ld * ld;
  float* EFb = EF + (size_t)b * ld * ld;
  float* Eb = Eo + (size_t)b * ld * ld;
  for (int pr = lane; pr < ld * ld; pr += NT) {
    const int a = pr / ld, c2 = pr % ld;
    double f = 0.0, ef = 0.0, e = 0.0;
    if (a < R && c2 < R) {
      f = Fm[a][c2];
      e = E[a][c2];
      for (int q = 0; q < R; ++q) ef += E[a][q] * Fm[q][c2];
    }
    Fb[pr] = (float)f;
    EFb[pr] = (float)ef;
    Eb[pr] = (float)e;
  }
  if (RS) {  // R-space form (lo_precond_desc.RS): E | F E | E F E | G2 = C^T C | F | E F in fp64, [6][ld][ld], zero padded
    // (gpart2: partials of C^T C; a constant diagonal has one set of partials -- E = C^T C / sigma was formed above)
    __syncthreads();
    for (int pr = lane; pr < R * R; pr += NT) {  // T <- E F (Y is no longer needed)
      const int a = pr / R, c2 = pr % R;
      double ef = 0.0;
      for (int q = 0; q < R; ++q) ef += E[a][q] * Fm[q][c2];
      T[a][c2] = ef;
    }
    __syncthreads();
    double* Rb = RS + (size_t)b * 6 * ld * ld;
    for (int pr = lane; pr < ld * ld; pr += NT) {
      const int a = pr / ld, c2 = pr % ld;
      double f = 0.0, ef = 0.0, fe = 0.0, efe = 0.0, e = 0.0, g2 = 0.0;
      if (a < R && c2 < R) {
        f = Fm[a][c2];
        e = E[a][c2];
        ef = T[a][c2];
        fe = T[c2][a];  // F E = (E F)^T
        for (int q = 0; q < R; ++q) efe += T[a][q] * E[q][c2];
        for (int s = 0; s < S; ++s) g2 += gpart2[((size_t)b * S + s) * R * R + a * R + c2];
      }
      Rb[pr] = e;
      Rb[ld * ld + pr] = fe;
      Rb[2 * ld * ld + pr] = efe;
      Rb[3 * ld * ld + pr] = g2;
      Rb[4 * ld * ld + pr] = f;
      Rb[5 * ld * ld + pr] = ef;
    }
    __syncthreads();
  }
  if (kappa) {  // sqrt(sum_m (F E F)_mm E_mm): amplification of the fp32 rounding of C^T (r/d) in P^-1 r (lo_amd.h)
    if (lane < R) {
      double t = 0.0;
      for (int q = 0; q < R; ++q) {
        double ef = 0.0;
        for (int c2 = 0; c2 < R; ++c2) ef += E[q][c2] * Fm[c2][lane];
        t += Fm[lane][q] * ef;
      }
      T[lane][0] = t * E[lane][lane];
    }
    __syncthreads();
    if (lane == 0) {
      double t = 0.0;
      for (int m = 0; m < R; ++m) t += T[m][0];
      kappa[b] = (float)sqrt(fabs(t));
    }
  }
  if (lane == 0) {
    double ldt = 0.0;
    for (int j = 0; j < k; ++j) ldt += log(fabs(G[j][j]));
    ldt *= 2.0;
    if (diag_mode == LO_DIAG_CONST) {
      ldt += (double)N * log(sigma);
      dinv_const[b] = (float)(1.0 / sigma);
    } else {
      double t = 0.0;
      for (int s = 0; s < S; ++s) t += logd_part[b * S + s];
      ldt += t;
    }
    logdet[b] = (float)ldt;
  }
}


// ---- Kronecker root form (lo_precond_desc.kron_*) --------------------------------------------------------------------
// Row pi of K1 (x) K2 (what the pivoted Cholesky fetched, _pivoted_cholesky.py:81 through kronecker_product...py:34-45)
// is K1[pi / n2, :] (x) K2[pi % n2, :].  One workgroup per member gathers the k <= 16 pivot rows of both factors
// (transposed and zero padded to 16: a quad of lanes reads the 16 values of a row index as one 64-byte line), the
// Gram matrix KP^T KP = (A^T A) o (B^T B) in fp64 and the start of the recurrence of k_pb_rootform (unit vectors).
__global__ __launch_bounds__(kThreads) void k_pb_kron_gather(const float* __restrict__ K1, const float* __restrict__ K2,
                                                             int n1, int n2, const long long* __restrict__ perm, int N,
                                                             int k, float* __restrict__ ka, float* __restrict__ kb,
                                                             double* __restrict__ gpart, float* __restrict__ Cpiv) {
  __shared__ int p1[16], p2[16];
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  if (t < 16) {
    const long long pv = t < k ? perm[(size_t)b * N + t] : 0;
    p1[t] = (int)(pv / n2);
    p2[t] = (int)(pv % n2);
  }
  __syncthreads();
  const float* K1b = K1 + (size_t)b * n1 * n1;
  const float* K2b = K2 + (size_t)b * n2 * n2;
  for (int e = t; e < n1 * 16; e += kThreads) {
    const int i = e >> 4, m = e & 15;
    ka[(size_t)b * n1 * 16 + e] = m < k ? K1b[(size_t)p1[m] * n1 + i] : 0.f;
  }
  for (int e = t; e < n2 * 16; e += kThreads) {
    const int i = e >> 4, m = e & 15;
    kb[(size_t)b * n2 * 16 + e] = m < k ? K2b[(size_t)p2[m] * n2 + i] : 0.f;
  }
  {  // one pair (m, n) per thread (kThreads == 256)
    const int m = t >> 4, n = t & 15;
    double g1 = 0.0, g2 = 0.0;
    float cp = 0.f;
    if (m < k && n < k) {
      const float* ra = K1b + (size_t)p1[m] * n1;
      const float* rb = K1b + (size_t)p1[n] * n1;
      for (int i = 0; i < n1; ++i) g1 += (double)ra[i] * (double)rb[i];
      ra = K2b + (size_t)p2[m] * n2;
      rb = K2b + (size_t)p2[n] * n2;
      for (int i = 0; i < n2; ++i) g2 += (double)ra[i] * (double)rb[i];
      // column m of K IS column m of KP: the recurrence of k_pb_rootform starts from the unit vectors
      // (M[:, j] = (e_j - sum_{i<j} M[:, i] L[pi_j, i]) / L[pi_j, j], i.e. M = Lp^-T with Lp = L[pivots, :])
      cp = (m == n) ? 1.0f : 0.0f;
    }
    gpart[(size_t)b * 256 + t] = g1 * g2;
    Cpiv[(size_t)b * 256 + t] = cp;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide preconditioners, 32 < k <= 128 (settings.max_preconditioner_size beyond the register-resident algebra above).
// Same mathematics -- G = I + W^T W in fp64, G = T T^T, Q = W T^-T -- as three plain kernels:
//   k_pbw_gram : 32 x 32 tiles of the lower triangle of W^T W, fp64 accumulation of fp32 products, rows staged in LDS;
//   k_pbw_chol : one workgroup per member, T and X = T^-1 packed lower-triangular in LDS (2 x 66 KB at k = 128), logdet;
//   k_pbw_q    : Q = (W X^T) per 64-row tile, 4 x 8 register tile per thread, fp32.
constexpr int kPbWideMaxK = 128;
constexpr int kPbwRows = 64;  // rows staged per step

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower triangle, j <= i

// grid (S, tile pairs, B); tile pair p -> (ti >= tj).  gpart [B, S, k, k]: only the lower triangle is written.
__global__ __launch_bounds__(kThreads) void k_pbw_gram(const float* __restrict__ L, LStride ls,
                                                        const float* __restrict__ sc, int N, int k, int rows_per,
                                                        double* __restrict__ gpart) {
  __shared__ float wi[kPbwRows][33];
  __shared__ float wj[kPbwRows][33];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.z;
  int ti = 0, p = blockIdx.y;
  while (p > ti) { p -= ti + 1; ++ti; }
  const int tj = p;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const int ta = threadIdx.x >> 4, tb = threadIdx.x & 15;  // outputs (2 ta + {0,1}, 2 tb + {0,1}) of the tile
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  const float* Lb = L + (size_t)b * ls.member;
  for (int rc = r0; rc < r1; rc += kPbwRows) {
    __syncthreads();
    for (int e = threadIdx.x; e < kPbwRows * 32; e += kThreads) {
      int rr, cc;
      if (ls.row == 1) { rr = e % kPbwRows; cc = e / kPbwRows; } else { cc = e & 31; rr = e >> 5; }
      const int row = rc + rr;
      float vi = 0.f, vj = 0.f;
      if (row < r1) {
        const float f = sc ? sc[(size_t)b * N + row] : 1.0f;
        const int ci = 32 * ti + cc, cj = 32 * tj + cc;
        if (ci < k) vi = L[(size_t)b * ls.member + (size_t)row * ls.row + (size_t)ci * ls.col] * f;
        if (cj < k) vj = (ti == tj) ? vi : L[(size_t)b * ls.member + (size_t)row * ls.row + (size_t)cj * ls.col] * f;
      }
      wi[rr][cc] = vi;
      wj[rr][cc] = vj;
    }
    __syncthreads();
#pragma unroll 8
    for (int rr = 0; rr < kPbwRows; ++rr) {
      const double a0 = wi[rr][2 * ta], a1 = wi[rr][2 * ta + 1];
      const double b0 = wj[rr][2 * tb], b1 = wj[rr][2 * tb + 1];
      acc[0][0] = fma(a0, b0, acc[0][0]); acc[0][1] = fma(a0, b1, acc[0][1]);
      acc[1][0] = fma(a1, b0, acc[1][0]); acc[1][1] = fma(a1, b1, acc[1][1]);
    }
  }
  (void)Lb;
  double* g = gpart + ((size_t)b * S + s) * k * k;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int i = 32 * ti + 2 * ta + x, j = 32 * tj + 2 * tb + y;
      if (i < k && j <= i) g[(size_t)i * k + j] = acc[x][y];
    }
}

// One workgroup per member.  Dynamic LDS: T (packed lower) | X (packed lower), k (k + 1) doubles.
// Xt [B, k, k] fp32: Xt[c][a] = X[a][c] (c <= a), zero above -- the operand layout k_pbw_q reads.
__global__ __launch_bounds__(kThreads) void k_pbw_chol(const double* __restrict__ gpart, const double* __restrict__ logd_part,
                                                        const float* __restrict__ dd, int diag_mode, int N, int k, int S,
                                                        float* __restrict__ Xt, float* __restrict__ logdet,
                                                        float* __restrict__ dinv_const) {
  extern __shared__ double tx[];
  const int np = k * (k + 1) / 2;
  double* T = tx;
  double* X = tx + np;
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  const double sigma = (diag_mode == LO_DIAG_CONST) ? (double)dd[b] : 1.0;
  for (int e = t; e < np; e += kThreads) {
    int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while (tri(i + 1, 0) <= e) ++i;
    while (tri(i, 0) > e) --i;
    const int j = e - tri(i, 0);
    double v = 0.0;
    for (int s = 0; s < S; ++s) v += gpart[(((size_t)b * S + s) * k + i) * k + j];  // fixed order
    if (diag_mode == LO_DIAG_CONST) v /= sigma;
    T[e] = v + (i == j ? 1.0 : 0.0);
  }
  // right-looking Cholesky, column by column
  for (int j = 0; j < k; ++j) {
    __syncthreads();
    const double piv = sqrt(T[tri(j, j)]);
    __syncthreads();
    for (int i = j + t; i < k; i += kThreads) T[tri(i, j)] = (i == j) ? piv : T[tri(i, j)] / piv;
    __syncthreads();
    const int m = k - j - 1;  // trailing block rows j+1 .. k-1
    for (int e = t; e < m * (m + 1) / 2; e += kThreads) {
      int ii = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
      while (tri(ii + 1, 0) <= e) ++ii;
      while (tri(ii, 0) > e) --ii;
      const int ll = e - tri(ii, 0);
      const int i = j + 1 + ii, l = j + 1 + ll;
      T[tri(i, l)] = fma(-T[tri(i, j)], T[tri(l, j)], T[tri(i, l)]);
    }
  }
  __syncthreads();
  // X = T^-1, one column per thread (forward substitution down the column)
  for (int c = t; c < k; c += kThreads) {
    X[tri(c, c)] = 1.0 / T[tri(c, c)];
    for (int i = c + 1; i < k; ++i) {
      double sacc = 0.0;
      for (int l = c; l < i; ++l) sacc = fma(T[tri(i, l)], X[tri(l, c)], sacc);
      X[tri(i, c)] = -sacc / T[tri(i, i)];
    }
  }
  __syncthreads();
  for (int e = t; e < k * k; e += kThreads) {
    const int c = e / k, a = e % k;
    Xt[(size_t)b * k * k + e] = (c <= a) ? (float)X[tri(a, c)] : 0.f;
  }
  if (t == 0) {
    double ldt = 0.0;
    for (int j = 0; j < k; ++j) ldt += log(T[tri(j, j)]);
    ldt *= 2.0;
    if (diag_mode == LO_DIAG_CONST) {
      ldt += (double)N * log(sigma);
      dinv_const[b] = (float)(1.0 / sigma);
    } else {
      double acc = 0.0;
      for (int s = 0; s < S; ++s) acc += logd_part[b * S + s];
      ldt += acc;
    }
    logdet[b] = (float)ldt;
  }
}

// Q[row][a] = (1 / d_row) sum_c L[row][c] Xt[c][a].  grid (row tiles of 64, B); dynamic LDS: Xt (k x ldq)
// | L tile transposed (k x 64).
__global__ __launch_bounds__(kThreads) void k_pbw_q(const float* __restrict__ L, LStride ls, const float* __restrict__ sc,
                                                     const float* __restrict__ dd, int diag_mode, int N, int k, int ldq,
                                                     const float* __restrict__ Xt, float* __restrict__ Q) {
  extern __shared__ float qs[];
  float* xs = qs;                       // [k][ldq]
  float* lt = qs + (size_t)k * ldq;     // [k][64]
  const int64_t b = blockIdx.y;
  const int r0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  for (int e = t; e < k * ldq; e += kThreads) {
    const int c = e / ldq, a = e % ldq;
    xs[e] = (a < k) ? Xt[(size_t)b * k * k + (size_t)c * k + a] : 0.f;
  }
  for (int e = t; e < k * 64; e += kThreads) {
    int rr, cc;
    if (ls.row == 1) { rr = e & 63; cc = e >> 6; } else { cc = e % k; rr = e / k; }
    const int row = r0 + rr;
    lt[cc * 64 + rr] = (row < N) ? L[(size_t)b * ls.member + (size_t)row * ls.row + (size_t)cc * ls.col] : 0.f;
  }
  __syncthreads();
  const int tr = t >> 4, ta = t & 15;  // rows 4 tr .. + 3; columns a = 8 ta + 128 h .. + 7
  for (int a0 = 8 * ta; a0 < ldq; a0 += 128) {
    float acc[4][8];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) acc[x][y] = 0.f;
    for (int c = 0; c < k; ++c) {
      const float4 l4 = *reinterpret_cast<const float4*>(&lt[c * 64 + 4 * tr]);
      const float4 xa = *reinterpret_cast<const float4*>(&xs[(size_t)c * ldq + a0]);
      const float4 xb = *reinterpret_cast<const float4*>(&xs[(size_t)c * ldq + a0 + 4]);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
      const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 8; ++y) acc[x][y] = fmaf(lv[x], xv[y], acc[x][y]);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int row = r0 + 4 * tr + x;
      if (row >= N) continue;
      // Q = D^-1/2 W T^-T with W = D^-1/2 L: the row factor is 1/d
      const float f = (diag_mode == LO_DIAG_CONST) ? 1.0f / dd[b] : sc[(size_t)b * N + row] * sc[(size_t)b * N + row];
      float4 o0 = make_float4(acc[x][0] * f, acc[x][1] * f, acc[x][2] * f, acc[x][3] * f);
      float4 o1 = make_float4(acc[x][4] * f, acc[x][5] * f, acc[x][6] * f, acc[x][7] * f);
      float* q = Q + ((size_t)b * N + row) * ldq + a0;
      *reinterpret_cast<float4*>(q) = o0;
      *reinterpret_cast<float4*>(q + 4) = o1;
    }
  }
}

static int precond_build_wide(const float* L, LStride ls, const float* d, int diag_mode, int64_t B, int64_t N, int k,
                              float* Q, float* dinv, float* logdet_p, double* gpart, double* logd, float* Xt,
                              float* scale, Split sp, hipStream_t st) {
  const int ldq = padded_k(k);
  dim3 block(kThreads);
  const float* sc = nullptr;
  if (diag_mode == LO_DIAG_FULL) {
    LO_PROF_BEGIN("pb_scale", st);
    hipLaunchKernelGGL(k_pb_scale, dim3(sp.S, (unsigned)B), block, 0, st, d, (int)N, sp.rows, scale, dinv, logd);
    LO_PROF_END(st);
    sc = scale;
  }
  const int nt = (k + 31) / 32;
  LO_PROF_BEGIN("pbw_gram", st);
  hipLaunchKernelGGL(k_pbw_gram, dim3(sp.S, nt * (nt + 1) / 2, (unsigned)B), block, 0, st, L, ls, sc, (int)N, k, sp.rows,
                     gpart);
  LO_PROF_END(st);
  const size_t chol_lds = (size_t)k * (k + 1) * sizeof(double);
  LO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pbw_chol), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)chol_lds));
  LO_PROF_BEGIN("pbw_chol", st);
  hipLaunchKernelGGL(k_pbw_chol, dim3((unsigned)B), block, chol_lds, st, gpart, logd, d, diag_mode, (int)N, k, sp.S, Xt,
                     logdet_p, dinv);
  LO_PROF_END(st);
  const size_t q_lds = ((size_t)k * ldq + (size_t)k * 64) * sizeof(float);
  LO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pbw_q), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)q_lds));
  LO_PROF_BEGIN("pbw_q", st);
  hipLaunchKernelGGL(k_pbw_q, dim3((unsigned)((N + 63) / 64), (unsigned)B), block, q_lds, st, L, ls, sc, d, diag_mode,
                     (int)N, k, ldq, Xt, Q);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
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
  ar.take<float>((size_t)B * N);  // per-row 1/sqrt(d) of the matrix-core path
  return ar.off + 1024;
}

// Q must hold [B, N, ldq] floats with ldq = 4 * pow2ceil(ceil(k/4)); dinv [B,N] (FULL) or [B] (CONST).
int lo_precond_build_f32(const float* L, const float* d, int32_t diag_mode, int64_t B, int64_t N, int32_t k, float* Q,
                         float* dinv, float* logdet_p, void* ws, size_t ws_bytes, void* stream) {
  return lo_precond_build_strided_f32(L, N * k, k, 1, d, diag_mode, B, N, k, Q, dinv, logdet_p, ws, ws_bytes, stream);
}

int lo_precond_build_strided_f32(const float* L, int64_t ld_member, int64_t ld_row, int64_t ld_col, const float* d,
                                 int32_t diag_mode, int64_t B, int64_t N, int32_t k, float* Q, float* dinv,
                                 float* logdet_p, void* ws, size_t ws_bytes, void* stream) {
  const LStride ls{ld_member, ld_row, ld_col};
  if (!L || !d || !Q || !dinv || !logdet_p || !ws) return LO_ERR_BADARG;
  if (diag_mode != LO_DIAG_FULL && diag_mode != LO_DIAG_CONST) return LO_ERR_BADARG;
  if (k < 1 || k > kPbWideMaxK) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Split sp = choose_split(B, N, 256);
  Arena ar(ws, ws_bytes);
  double* gpart = ar.take<double>((size_t)B * sp.S * k * k);
  double* logd = ar.take<double>((size_t)B * sp.S);
  double* Minv = ar.take<double>((size_t)B * k * k);
  float* scale = ar.take<float>((size_t)B * N);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  // wide path: k > 32, and 16 < k <= 32 in the rows layout [B, m, N] the pivoted-Cholesky kernels write (the fp64
  // matrix-core kernels of that layout take 16 columns; the per-row VALU kernels that used to serve this case need
  // 6.4 ms at 512 x 8192 x 17 against ~1 ms here)   (Minv's storage holds the fp32 k x k operand of the Q kernel)
  if (k > kPbMaxK || (k > 16 && ld_col != 1))
    return precond_build_wide(L, ls, d, diag_mode, B, N, k, Q, dinv, logdet_p, gpart, logd,
                              reinterpret_cast<float*>(Minv), scale, sp, st);
  const int ldq = padded_k(k);
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  // rows layout ([B, m, N] as the pivoted-Cholesky kernels write it), k <= 16: fp64 matrix cores, 16-byte loads
  const bool mfma = ld_row == 1 && k <= 16 && (N % 4) == 0 && (ld_col % 4) == 0 && (ld_member % 4) == 0 &&
                    ((uintptr_t)L % 16) == 0 && ((uintptr_t)scale % 16) == 0;
  // reference layout [B, N, k] (ld_col == 1), k <= 32: matrix-core kernels with per-row loads
  const bool mfma_nk = !mfma && ld_col == 1 && k <= 32;
  if (mfma || mfma_nk) {
    const float* sc = nullptr;
    if (diag_mode == LO_DIAG_FULL) {
      LO_PROF_BEGIN("pb_scale", st);
      hipLaunchKernelGGL(k_pb_scale, grid, block, 0, st, d, (int)N, sp.rows, scale, dinv, logd);
      LO_PROF_END(st);
      sc = scale;
    }
    LO_PROF_BEGIN("pb_gram_mfma", st);
    if (mfma) hipLaunchKernelGGL(k_pb_gram_mfma, grid, block, 0, st, L, ls, sc, (int)N, (int)k, sp.rows, gpart);
    else if (k <= 16) hipLaunchKernelGGL((k_pb_gram_mfma_nk<1>), grid, block, 0, st, L, ls, sc, (int)N, (int)k, sp.rows, gpart);
    else hipLaunchKernelGGL((k_pb_gram_mfma_nk<2>), grid, block, 0, st, L, ls, sc, (int)N, (int)k, sp.rows, gpart);
    LO_PROF_END(st);
    hipLaunchKernelGGL(k_pb_chol, dim3((unsigned)B), dim3(64), 0, st, gpart, logd, d, diag_mode, (int)N, (int)k, sp.S,
                       Minv, logdet_p, dinv);
    LO_PROF_BEGIN("pb_q_mfma", st);
    if (mfma)
      hipLaunchKernelGGL(k_pb_q_mfma, grid, block, 0, st, L, ls, sc, d, diag_mode, (int)N, (int)k, ldq, sp.rows, Minv, Q);
    else if (k <= 16)
      hipLaunchKernelGGL((k_pb_q_mfma_nk<1>), grid, block, 0, st, L, ls, sc, d, diag_mode, (int)N, (int)k, ldq, sp.rows,
                         Minv, Q);
    else
      hipLaunchKernelGGL((k_pb_q_mfma_nk<2>), grid, block, 0, st, L, ls, sc, d, diag_mode, (int)N, (int)k, ldq, sp.rows,
                         Minv, Q);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
    return LO_OK;
  }
  LO_PROF_BEGIN("pb_gram", st);
  hipLaunchKernelGGL(k_pb_gram, grid, block, 0, st, L, ls, d, diag_mode, (int)N, (int)k, sp.rows, gpart, logd);
  LO_PROF_END(st);
  hipLaunchKernelGGL(k_pb_chol, dim3((unsigned)B), dim3(64), 0, st, gpart, logd, d, diag_mode, (int)N, (int)k, sp.S,
                     Minv, logdet_p, dinv);
  LO_PROF_BEGIN("pb_q", st);
  if (k <= 16)
    hipLaunchKernelGGL((k_pb_q<16>), grid, block, 0, st, L, ls, d, diag_mode, (int)N, (int)k, ldq, sp.rows, Minv, Q, dinv);
  else
    hipLaunchKernelGGL((k_pb_q<32>), grid, block, 0, st, L, ls, d, diag_mode, (int)N, (int)k, ldq, sp.rows, Minv, Q, dinv);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

size_t lo_precond_root_form_workspace_bytes(int64_t B, int64_t N, int32_t R) {
  Split sp = choose_split(B, N, 256);
  Arena ar(nullptr, 0);
  ar.take<double>((size_t)B * sp.S * R * R);
  ar.take<double>((size_t)B * sp.S);
  ar.take<float>((size_t)B * N);
  return ar.off + 1024;
}

int lo_precond_root_form_f32(const float* C, int32_t R, const float* d, int32_t diag_mode, const float* L,
                             int64_t ld_member, int64_t ld_row, int64_t ld_col, const int64_t* perm, int64_t B,
                             int64_t N, int32_t k, int32_t rf_ld, float* F, float* EF, float* E, float* dinv,
                             float* logdet_p, void* ws, size_t ws_bytes, void* stream) {
  if (!C || !d || !L || !perm || !F || !EF || !E || !dinv || !logdet_p || !ws) return LO_ERR_BADARG;
  if (diag_mode != LO_DIAG_FULL && diag_mode != LO_DIAG_CONST) return LO_ERR_BADARG;
  if (R < 1 || R > kPbMaxK || k < 1 || k > kPbMaxK || rf_ld < R || rf_ld > kPbMaxK) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Split sp = choose_split(B, N, 256);
  Arena ar(ws, ws_bytes);
  double* gpart = ar.take<double>((size_t)B * sp.S * R * R);
  double* logd = ar.take<double>((size_t)B * sp.S);
  float* scale = ar.take<float>((size_t)B * N);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  const float* sc = nullptr;
  if (diag_mode == LO_DIAG_FULL) {
    LO_PROF_BEGIN("pb_scale", st);
    hipLaunchKernelGGL(k_pb_scale, grid, block, 0, st, d, (int)N, sp.rows, scale, dinv, logd);
    LO_PROF_END(st);
    sc = scale;
  }
  const LStride cs{N * (int64_t)R, (int64_t)R, 1};
  LO_PROF_BEGIN("pb_gram_root", st);  // E partials = W^T W, W = C / sqrt(d), fp64 matrix cores
  if ((R % 4) == 0 && ((uintptr_t)C % 16) == 0) {  // LDS-staged rows (coalesced 16-byte loads)
    if (R <= 16) hipLaunchKernelGGL((k_pb_gram_root<1>), grid, block, 0, st, C, sc, (int)N, (int)R, sp.rows, gpart);
    else hipLaunchKernelGGL((k_pb_gram_root<2>), grid, block, 0, st, C, sc, (int)N, (int)R, sp.rows, gpart);
  } else if (R <= 16) {
    hipLaunchKernelGGL((k_pb_gram_mfma_nk<1>), grid, block, 0, st, C, cs, sc, (int)N, (int)R, sp.rows, gpart);
  } else {
    hipLaunchKernelGGL((k_pb_gram_mfma_nk<2>), grid, block, 0, st, C, cs, sc, (int)N, (int)R, sp.rows, gpart);
  }
  LO_PROF_END(st);
  const LStride ls{ld_member, ld_row, ld_col};
  LO_PROF_BEGIN("pb_rootform", st);
  hipLaunchKernelGGL(k_pb_rootform, dim3((unsigned)B), dim3(kThreads), 0, st, gpart, logd, C, (int)R, d, diag_mode, L, ls,
                     (const long long*)perm, (int)N, (int)k, sp.S, (int)rf_ld, F, EF, E, logdet_p, dinv);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

size_t lo_precond_root_form_rs_workspace_bytes(int64_t B, int64_t N, int32_t R) {
  Split sp = choose_split(B, N, 256);
  Arena ar(nullptr, 0);
  ar.take<double>((size_t)B * sp.S * R * R);
  ar.take<double>((size_t)B * sp.S * R * R);
  ar.take<double>((size_t)B * sp.S);
  ar.take<float>((size_t)B * N);
  return ar.off + 1024;
}

int lo_precond_root_form_rs_f32(const float* C, int32_t R, const float* d, int32_t diag_mode, const float* L,
                                int64_t ld_member, int64_t ld_row, int64_t ld_col, const int64_t* perm, int64_t B,
                                int64_t N, int32_t k, int32_t rf_ld, float* F, float* EF, float* E, float* dinv,
                                float* logdet_p, double* RS, void* ws, size_t ws_bytes, void* stream) {
  if (!C || !d || !L || !perm || !F || !EF || !E || !dinv || !logdet_p || !RS || !ws) return LO_ERR_BADARG;
  if (diag_mode != LO_DIAG_FULL && diag_mode != LO_DIAG_CONST) return LO_ERR_BADARG;
  if (R < 1 || R > kPbMaxK || k < 1 || k > kPbMaxK || rf_ld < R || rf_ld > kPbMaxK) return LO_ERR_UNSUPPORTED;
  if ((R % 4) != 0 || ((uintptr_t)C % 16) != 0) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Split sp = choose_split(B, N, 256);
  Arena ar(ws, ws_bytes);
  double* gpartE = ar.take<double>((size_t)B * sp.S * R * R);
  double* gpart2 = ar.take<double>((size_t)B * sp.S * R * R);
  double* logd = ar.take<double>((size_t)B * sp.S);
  float* scale = ar.take<float>((size_t)B * N);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  if (diag_mode == LO_DIAG_FULL) {  // dinv = fp32(1 / d): THE D^-1 of the R-space iteration (and log d partials)
    LO_PROF_BEGIN("pb_scale", st);
    hipLaunchKernelGGL(k_pb_scale, grid, block, 0, st, d, (int)N, sp.rows, scale, dinv, logd);
    LO_PROF_END(st);
  }
  LO_PROF_BEGIN("rs_gram64", st);  // E = C^T diag(dinv) C and C^T C on the fp64 matrix cores (exact products)
  rs_gram64_launch(C, diag_mode == LO_DIAG_FULL ? dinv : nullptr, B, N, (int)R, sp, gpartE, gpart2, st);
  LO_PROF_END(st);
  const LStride ls{ld_member, ld_row, ld_col};
  LO_PROF_BEGIN("pb_rootform", st);
  // FULL: the partials of E already carry dinv (k_pb_rootform divides by sigma = 1); CONST: E = C^T C / sigma
  hipLaunchKernelGGL(k_pb_rootform, dim3((unsigned)B), dim3(kThreads), 0, st,
                     diag_mode == LO_DIAG_FULL ? gpartE : gpart2, logd, C, (int)R, d, diag_mode, L, ls,
                     (const long long*)perm, (int)N, (int)k, sp.S, (int)rf_ld, F, EF, E, logdet_p, dinv,
                     (const float*)nullptr, (float*)nullptr, (const double*)gpart2, RS);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

size_t lo_precond_kron_root_workspace_bytes(int64_t B) {
  Arena ar(nullptr, 0);
  ar.take<double>((size_t)B * 256);
  ar.take<double>((size_t)B);
  ar.take<float>((size_t)B * 256 * 3);
  ar.take<float>((size_t)B * 2);
  return ar.off + 1024;
}

int lo_precond_kron_root_f32(const lo_op_desc* op, const float* L, int64_t ld_member, int64_t ld_row, int64_t ld_col,
                             const int64_t* perm, int32_t k, float* kron_a, float* kron_b, float* kron_F, float* kappa,
                             void* ws, size_t ws_bytes, void* stream) {
  if (!op || !L || !perm || !kron_a || !kron_b || !kron_F || !kappa || !ws) return LO_ERR_BADARG;
  if (op->kind != LO_OP_KRON_DIAG || op->diag_mode != LO_DIAG_CONST || !op->A0 || !op->A1 || !op->d)
    return LO_ERR_UNSUPPORTED;
  if (k < 1 || k > 16 || op->R < 1 || op->n2 < 1 || op->R * op->n2 != op->N) return LO_ERR_UNSUPPORTED;
  static_assert(kThreads == 256, "k_pb_kron_gather: one (m, n) pair per thread");
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = op->B;
  Arena ar(ws, ws_bytes);
  double* gpart = ar.take<double>((size_t)B * 256);
  double* logd = ar.take<double>((size_t)B);
  float* scratch = ar.take<float>((size_t)B * 256 * 3);  // pivot rows of KP, EF, E
  float* small = ar.take<float>((size_t)B * 2);          // logdet, 1 / sigma (both already known from the Q form)
  if (!ar.ok) return LO_ERR_WORKSPACE;
  float* Cpiv = scratch;
  LO_PROF_BEGIN("pb_kron_root", st);
  hipLaunchKernelGGL(k_pb_kron_gather, dim3((unsigned)B), dim3(kThreads), 0, st, op->A0, op->A1, (int)op->R, (int)op->n2,
                     (const long long*)perm, (int)op->N, (int)k, kron_a, kron_b, gpart, Cpiv);
  const LStride ls{ld_member, ld_row, ld_col};
  hipLaunchKernelGGL(k_pb_rootform, dim3((unsigned)B), dim3(kThreads), 0, st, gpart, logd, (const float*)nullptr, 16,
                     op->d, (int)LO_DIAG_CONST, L, ls, (const long long*)perm, (int)op->N, (int)k, 1, 16, kron_F,
                     scratch + (size_t)B * 256, scratch + (size_t)B * 512, small, small + B, (const float*)Cpiv, kappa);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // extern "C"
