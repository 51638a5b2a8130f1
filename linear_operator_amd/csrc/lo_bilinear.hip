// lo_bilinear.hip -- `_bilinear_derivative` contractions of the backward passes: given
// U = left_vecs [B,N,D] and V = right_vecs [B,N,D], the derivative of sum_d u_d^T K v_d with respect to the tensors
// that represent K:
//   Dense  K            : U V^T                         (operators/dense_linear_operator.py:69-71)
//   Diag   diag(k)      : sum_d U o V   [B,N]           (operators/diag_linear_operator.py:37-45)
//   ConstantDiag s I    : sum_{n,d} U o V   [B,1]       (operators/diag_linear_operator.py:337-344)
//   Root   C C^T        : U (V^T C) + V (U^T C)  [B,N,R]  (autograd of root._matmul(root._t_matmul(v)),
//                          operators/_linear_operator.py:336-393 + root_linear_operator.py:68-72)
// All HBM-bound streaming kernels: the dense one writes N^2 outputs per member, the others read the skinny operands
// once or twice.  fp32, fixed summation order (partials over row slices are summed in order by the consumer).
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int kBdTile = 64;   // output tile of the dense outer product
constexpr int kBdMaxD = 64;   // columns of U / V staged per pass

// out[b, i, j] (+)= sum_d U[b,i,d] V[b,j,d]
__global__ __launch_bounds__(kThreads) void k_bil_dense(const float* __restrict__ U, const float* __restrict__ V,
                                                         int N, int D, int d0, int dn, float* __restrict__ out) {
  __shared__ float u_s[kBdTile][kBdMaxD + 1];
  __shared__ float v_s[kBdTile][kBdMaxD + 1];
  const int64_t b = blockIdx.z;
  const int i0 = blockIdx.y * kBdTile, j0 = blockIdx.x * kBdTile;
  for (int e = threadIdx.x; e < kBdTile * dn; e += kThreads) {
    const int r = e / dn, d = e % dn;
    u_s[r][d] = (i0 + r < N) ? U[((size_t)b * N + i0 + r) * D + d0 + d] : 0.f;
    v_s[r][d] = (j0 + r < N) ? V[((size_t)b * N + j0 + r) * D + d0 + d] : 0.f;
  }
  __syncthreads();
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;  // 16 x 16 threads, 4 x 4 outputs each
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
  for (int d = 0; d < dn; ++d) {
    float uu[4], vv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      uu[a] = u_s[ty + 16 * a][d];
      vv[a] = v_s[tx + 16 * a][d];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = fmaf(uu[a], vv[c], acc[a][c]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + ty + 16 * a;
    if (i >= N) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + tx + 16 * c;  // consecutive threads -> consecutive columns
      if (j < N) {
        float* o = out + ((size_t)b * N + i) * N + j;
        *o = (d0 == 0) ? acc[a][c] : *o + acc[a][c];
      }
    }
  }
}

// out[b, n] = sum_d U o V.  A workgroup takes `rb` consecutive rows = rb * D CONTIGUOUS floats of U and V: the products
// are formed with coalesced loads (a thread per row would issue D loads of stride D each), parked in LDS, and thread r
// sums its row in fixed order (stride D reads).
constexpr int kBdiagLds = 8192;  // floats
__global__ __launch_bounds__(kThreads) void k_bil_diag(const float* __restrict__ U, const float* __restrict__ V,
                                                        int64_t rows, int D, int rb, float* __restrict__ out) {
  __shared__ float prod[kBdiagLds];
  const int64_t r0 = (int64_t)blockIdx.x * rb;
  const int nr = (int)min((int64_t)rb, rows - r0);
  if (nr <= 0) return;
  const size_t base = (size_t)r0 * D;
  const int ne = nr * D;
  for (int e = threadIdx.x; e < ne; e += kThreads) prod[e] = U[base + e] * V[base + e];
  __syncthreads();
  for (int r = threadIdx.x; r < nr; r += kThreads) {
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc += prod[r * D + d];
    out[r0 + r] = acc;
  }
}

// out[b] = sum_n rowdot[b, n]  (fixed order: thread-strided partials, then the block tree)
__global__ __launch_bounds__(kThreads) void k_bil_sum_rows(const float* __restrict__ rowdot, int N,
                                                            float* __restrict__ out) {
  __shared__ float red[kThreads];
  const int64_t b = blockIdx.x;
  float acc = 0.f;
  for (int n = threadIdx.x; n < N; n += kThreads) acc += rowdot[(size_t)b * N + n];
  const float tot = block_sum256(acc, red);
  if (threadIdx.x == 0) out[b] = tot;
}

// Root, phase A: partial T1 = V^T C and T2 = U^T C over a slice of rows: tpart[b, s, 2, D, R]
constexpr int kBrRows = 32;

__global__ __launch_bounds__(kThreads) void k_bil_root_t(const float* __restrict__ C, const float* __restrict__ U,
                                                          const float* __restrict__ V, int N, int R, int D,
                                                          int rows_per, float* __restrict__ tpart) {
  extern __shared__ float sh[];  // c_s [kBrRows][R] | u_s [kBrRows][D] | v_s [kBrRows][D]
  float* c_s = sh;
  float* u_s = c_s + kBrRows * R;
  float* v_s = u_s + kBrRows * D;
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const int npair = D * R;
  // thread t owns the pairs t, t + 256, ... (at most 8 per matrix: D * R <= 2048)
  float a1[8], a2[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a1[u] = a2[u] = 0.f;
  for (int base = r0; base < r1; base += kBrRows) {
    const int nr = min(kBrRows, r1 - base);
    __syncthreads();
    for (int e = threadIdx.x; e < nr * R; e += kThreads) c_s[e] = C[((size_t)b * N + base) * R + e];
    for (int e = threadIdx.x; e < nr * D; e += kThreads) {
      u_s[e] = U[((size_t)b * N + base) * D + e];
      v_s[e] = V[((size_t)b * N + base) * D + e];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pr = threadIdx.x + kThreads * u;
      if (pr < npair) {
        const int d = pr / R, rho = pr % R;
        float x1 = a1[u], x2 = a2[u];
        for (int rr = 0; rr < nr; ++rr) {
          const float cv = c_s[rr * R + rho];
          x1 = fmaf(v_s[rr * D + d], cv, x1);
          x2 = fmaf(u_s[rr * D + d], cv, x2);
        }
        a1[u] = x1;
        a2[u] = x2;
      }
    }
  }
  float* tp = tpart + ((size_t)b * S + s) * 2 * npair;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int pr = threadIdx.x + kThreads * u;
    if (pr < npair) {
      tp[pr] = a1[u];
      tp[npair + pr] = a2[u];
    }
  }
}

// Root, phase A on the matrix cores (R <= 32, D <= 64): D-matrix[i = rho][j = d] += C^T[rho][row] * W[row][d] with
// v_mfma_f32_32x32x2_f32, two rows per instruction; lane l loads C[row + (l >> 5)][l & 31] and W[row + (l >> 5)][32 t +
// (l & 31)] -- one wave load = two consecutive rows, contiguous -- for W = V (-> T1) and W = U (-> T2).  C, U, V are
// streamed exactly once.  Same tpart layout as k_bil_root_t.
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NTD>
__global__ __launch_bounds__(kThreads) void k_bil_root_t_mfma(const float* __restrict__ C, const float* __restrict__ U,
                                                               const float* __restrict__ V, int N, int R, int D,
                                                               int rows_per, float* __restrict__ tpart) {
  __shared__ float red[4][32][33];
  const int s = blockIdx.x, S = gridDim.x;
  const int64_t b = blockIdx.y;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, kk = lane >> 5;
  const float* Cb = C + (size_t)b * N * R;
  const float* Ub = U + (size_t)b * N * D;
  const float* Vb = V + (size_t)b * N * D;
  f32x16 acc[2][NTD];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NTD; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.f;
  constexpr int UNR = 4;  // row pairs in flight per wave
  const int npairs = (r1 - r0 + 1) / 2;
  for (int pr0 = wave; pr0 < npairs; pr0 += 4 * UNR) {
    float cv[UNR], uv[UNR][NTD], vv[UNR][NTD];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pr = pr0 + 4 * u;
      const int row = r0 + 2 * pr + kk;
      const bool ok = (pr < npairs) && (row < r1);
      cv[u] = (ok && li < R) ? Cb[(size_t)row * R + li] : 0.f;
#pragma unroll
      for (int t = 0; t < NTD; ++t) {
        const int d = 32 * t + li;
        const bool okd = ok && d < D;
        uv[u][t] = okd ? Ub[(size_t)row * D + d] : 0.f;
        vv[u][t] = okd ? Vb[(size_t)row * D + d] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int t = 0; t < NTD; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cv[u], vv[u][t], acc[0][t], 0, 0, 0);  // T1 = V^T C
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cv[u], uv[u][t], acc[1][t], 0, 0, 0);  // T2 = U^T C
      }
  }
  // cross-wave sum in fixed order; D-matrix element (i = rho, j = d - 32 t) -> tp[m][d][rho]
  const int npair = D * R;
  float* tp = tpart + ((size_t)b * S + s) * 2 * npair;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < NTD; ++t) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * kk][li] = acc[m][t][e];
      __syncthreads();
      for (int idx = threadIdx.x; idx < 32 * 32; idx += kThreads) {
        const int j = idx >> 5, i = idx & 31;  // consecutive threads -> consecutive rho
        const int d = 32 * t + j;
        if (i < R && d < D)
          tp[(size_t)m * npair + (size_t)d * R + i] = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
      }
    }
}

// Root, phase B: out[b, n, rho] = sum_d U[n,d] T1[d,rho] + V[n,d] T2[d,rho], T = sum_s tpart (fixed order)
// Optional `rowdot` [B, N] = sum_d U o V (the Diag derivative of an AddedDiag operator, diag_linear_operator.py:37-45):
// the rows of U and V are in the workgroup's hands anyway.
// = a [rows x 2D] x [2D x R] product: a workgroup stages 32 * nw rows of U and V (contiguous rows * D floats each,
// coalesced) and T1 | T2 (reduced over the S slices once per workgroup) in LDS; wave w forms the 32 rows 32 w .. + 31
// with v_mfma_f32_32x32x2_f32, one 32-column block of the output at a time (stores of 128 contiguous bytes per row).
typedef float bo_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(kThreads) void k_bil_root_out(const float* __restrict__ U, const float* __restrict__ V,
                                                            const float* __restrict__ tpart, int S, int N, int R,
                                                            int D, int nw, int tiles, float* __restrict__ out,
                                                            float* __restrict__ rowdot) {
  extern __shared__ float sh[];  // t1 [DE][RP] | t2 [DE][RP] | u_s [rows][DP] | v_s [rows][DP]
  const int DE = (D + 1) & ~1;     // k extent, even (the instruction takes two k per step)
  const int DP = DE | 1;           // odd row stride of the U / V tiles: conflict-free column reads
  const int RP = (R + 31) & ~31;   // whole 32-column blocks
  const int rows = 32 * nw;
  float* t1 = sh;
  float* t2 = t1 + DE * RP;
  float* u_s = t2 + DE * RP;
  float* v_s = u_s + rows * DP;
  const int64_t b = blockIdx.y;
  const int npair = D * R;
  for (int e = threadIdx.x; e < 2 * DE * RP; e += kThreads) {
    const int which = e / (DE * RP), rem = e % (DE * RP), d = rem / RP, rho = rem % RP;
    float acc = 0.f;
    if (d < D && rho < R)
      for (int s = 0; s < S; ++s) acc += tpart[((size_t)b * S + s) * 2 * npair + (size_t)which * npair + d * R + rho];
    sh[e] = acc;
  }
  // a workgroup forms `tiles` consecutive blocks of `rows` rows with ONE reduction of T1 | T2 (the prologue used to be paid
  // per 128 rows: 32 768 workgroups of 16 KB of output each at the cfg3 shape)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  for (int tile = 0; tile < tiles; ++tile) {
  const int row0 = (blockIdx.x * tiles + tile) * rows;
  if (row0 >= N) break;
  const int nr = min(rows, N - row0);
  if (tile) __syncthreads();  // (the last block's products have read the staged rows)
  const size_t base = ((size_t)b * N + row0) * D;
  {  // the nr * D staged floats are contiguous in U / V; (row, column) of element e advanced without divisions
    const int step_r = kThreads / D, step_d = kThreads % D;
    int r = threadIdx.x / D, d = threadIdx.x % D;
    for (int e = threadIdx.x; e < nr * D; e += kThreads) {
      u_s[r * DP + d] = U[base + e];
      v_s[r * DP + d] = V[base + e];
      r += step_r;
      d += step_d;
      if (d >= D) { d -= D; ++r; }
    }
    // padding: columns D .. DP-1 of every row, and the rows beyond nr
    for (int e = threadIdx.x; e < rows * (DP - D); e += kThreads) {
      const int rr = e / (DP - D), dd = D + e % (DP - D);
      u_s[rr * DP + dd] = 0.f;
      v_s[rr * DP + dd] = 0.f;
    }
    for (int e = nr * DP + threadIdx.x; e < rows * DP; e += kThreads) {
      u_s[e] = 0.f;
      v_s[e] = 0.f;
    }
  }
  __syncthreads();
  if (rowdot && threadIdx.x < nr) {
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot = fmaf(u_s[threadIdx.x * DP + d], v_s[threadIdx.x * DP + d], dot);
    rowdot[(size_t)b * N + row0 + threadIdx.x] = dot;
  }
  if (wave < nw) {
  const float* ua = u_s + (32 * wave + li) * DP + h;
  const float* va = v_s + (32 * wave + li) * DP + h;
  for (int cb = 0; cb < RP; cb += 32) {
    bo_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* b1 = t1 + h * RP + cb + li;
    const float* b2 = t2 + h * RP + cb + li;
    for (int k = 0; k < DE; k += 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[k], b1[k * RP], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[k], b2[k * RP], acc, 0, 0, 0);
    }
    const int col = cb + li;
    if (col < R) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (r < nr) out[((size_t)b * N + row0 + r) * R + col] = acc[e];
      }
    }
  }
  }
  }
}

// out[b, n, :] += U[b, n, :] T[b]   (U [B,N,D], T [B,D,R], out [B,N,R] read and written in place): the N-sized step of the
// pull-back through the pivoted Cholesky of a root (functions/_pivoted_cholesky.py::_dense_root_vjp, bar R = G2 (L11^-1 Rm),
// the reference's PivotedCholesky.backward :107-147 by autograd) ADDED to the gradient the operator's own bilinear
// derivative has already left in `out` -- one pass (U once, out once in, once out) instead of a library GEMM with N x D x R
// of work in a [N x D] [D x R] shape it runs at 1.3 TB/s plus an elementwise sum of two [B,N,R] tensors.
// Layout: the chunk-per-lane layout of lo_lowrank_mv.hip -- lane (g = l / CH, k = l % CH) owns the 16-byte chunk k of the rows
// RPI i + g of its wave's 256 rows: every load / store of `out` is 1 KiB of consecutive addresses; T's rows of chunk k wait in
// registers (DP x 4 floats), the wave's rows of U (contiguous in memory) in its own LDS stage.
template <int RC, int DP>
__global__ __launch_bounds__(kThreads) void k_root_apply_add(const float* __restrict__ U, const float* __restrict__ T,
                                                              int N, int D, float* __restrict__ out) {
  constexpr int CH = RC / 4, RPI = 64 / CH;
  constexpr int LD = DP + 4;  // row stride of the stage (16-byte rows, the 8 rows of an instruction on different banks)
  constexpr int SR = DP <= 8 ? 256 : (DP <= 16 ? 128 : 64);  // rows of U staged at a time (<= 40 KB of LDS per workgroup)
  __shared__ __attribute__((aligned(16))) float ust[4][SR * LD];
  const int64_t b = blockIdx.y;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int k = lane & (CH - 1), g = lane / CH;
  const int row0 = blockIdx.x * 1024 + 256 * wave;
  if (row0 >= N) return;
  const int nr = min(256, N - row0);
  float* us = ust[wave];
  // T's chunk k of every row d (zero beyond D)
  float4 tw[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d)
    tw[d] = d < D ? *reinterpret_cast<const float4*>(T + ((size_t)b * D + d) * RC + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* ob = out + ((size_t)b * N + row0) * RC + 4 * k;
  const int step_r = 64 / D, step_d = 64 % D;
  for (int sb = 0; sb < nr; sb += SR) {
    const int ns = min(SR, nr - sb);
    // the block's ns x D floats of U are contiguous; (row, column) of element e advanced without divisions
    {
      const size_t base = ((size_t)b * N + row0 + sb) * D;
      int r = lane / D, d = lane % D;
      for (int e = lane; e < ns * D; e += 64) {
        us[r * LD + d] = U[base + e];
        r += step_r;
        d += step_d;
        if (d >= D) { d -= D; ++r; }
      }
      if (sb == 0)  // columns D .. DP-1 stay zero
        for (int e = lane; e < SR * (DP - D); e += 64) us[(e / (DP - D)) * LD + D + e % (DP - D)] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int i = 0; i < SR / RPI; ++i) {
      const int rl = RPI * i + g;  // row inside the staged block
      if (rl < ns) {
        float* orow = ob + (size_t)(sb + rl) * RC;
        float4 acc = *reinterpret_cast<const float4*>(orow);
#pragma unroll
        for (int d4 = 0; d4 < DP; d4 += 4) {
          const float4 u4 = *reinterpret_cast<const float4*>(&us[rl * LD + d4]);
          const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc.x = fmaf(uu[j], tw[d4 + j].x, acc.x);
            acc.y = fmaf(uu[j], tw[d4 + j].y, acc.y);
            acc.z = fmaf(uu[j], tw[d4 + j].z, acc.z);
            acc.w = fmaf(uu[j], tw[d4 + j].w, acc.w);
          }
        }
        *reinterpret_cast<float4*>(orow) = acc;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next block reuses the stage
  }
}

}  // namespace lo

using namespace lo;

extern "C" {

int lo_bilinear_dense_f32(const float* U, const float* V, int64_t B, int64_t N, int64_t D, float* out, void* stream) {
  if (!U || !V || !out || B < 1 || N < 1 || D < 1 || B > 65535) return LO_ERR_BADARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((N + kBdTile - 1) / kBdTile), (unsigned)((N + kBdTile - 1) / kBdTile), (unsigned)B);
  for (int64_t d0 = 0; d0 < D; d0 += kBdMaxD) {  // more than 64 columns: accumulate passes
    const int dn = (int)std::min<int64_t>(kBdMaxD, D - d0);
    LO_PROF_BEGIN("bil_dense", st);
    hipLaunchKernelGGL(k_bil_dense, grid, dim3(kThreads), 0, st, U, V, (int)N, (int)D, (int)d0, dn, out);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

int lo_bilinear_diag_f32(const float* U, const float* V, int64_t B, int64_t N, int64_t D, int32_t constant,
                         float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!U || !V || !out || B < 1 || N < 1 || D < 1) return LO_ERR_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = B * N;
  float* rowdot = out;
  if (constant) {
    if (!ws || ws_bytes < sizeof(float) * (size_t)rows) return LO_ERR_WORKSPACE;
    rowdot = (float*)ws;
  }
  if (D > kBdiagLds) return LO_ERR_UNSUPPORTED;
  const int rb = (int)std::min<int64_t>(kThreads, kBdiagLds / D);
  LO_PROF_BEGIN("bil_diag", st);
  hipLaunchKernelGGL(k_bil_diag, dim3((unsigned)((rows + rb - 1) / rb)), dim3(kThreads), 0, st, U, V, rows, (int)D, rb,
                     rowdot);
  LO_PROF_END(st);
  if (constant) hipLaunchKernelGGL(k_bil_sum_rows, dim3((unsigned)B), dim3(kThreads), 0, st, rowdot, (int)N, out);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

size_t lo_bilinear_root_workspace_bytes(int64_t B, int64_t N, int64_t R, int64_t D) {
  Split sp = choose_split(B, N, 256);
  return sizeof(float) * (size_t)B * sp.S * 2 * D * R + 256;
}

int lo_bilinear_root_f32(const float* C, const float* U, const float* V, int64_t B, int64_t N, int64_t R, int64_t D,
                         float* out, float* rowdot, void* ws, size_t ws_bytes, void* stream) {
  if (!C || !U || !V || !out || !ws || B < 1 || N < 1 || R < 1 || D < 1 || B > 65535) return LO_ERR_BADARG;
  if (D * R > 8 * kThreads) return LO_ERR_UNSUPPORTED;
  const size_t lds_a = sizeof(float) * (size_t)kBrRows * (R + 2 * D), lds_b = sizeof(float) * (size_t)2 * D * R;
  if (lds_a > 64 * 1024 || lds_b > 64 * 1024) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Split sp = choose_split(B, N, 256);
  if (ws_bytes < sizeof(float) * (size_t)B * sp.S * 2 * D * R) return LO_ERR_WORKSPACE;
  float* tpart = (float*)ws;
  LO_PROF_BEGIN("bil_root_t", st);
  if (R <= 32 && D <= 32)
    hipLaunchKernelGGL(k_bil_root_t_mfma<1>, dim3(sp.S, (unsigned)B), dim3(kThreads), 0, st, C, U, V, (int)N, (int)R,
                       (int)D, sp.rows, tpart);
  else if (R <= 32 && D <= 64)
    hipLaunchKernelGGL(k_bil_root_t_mfma<2>, dim3(sp.S, (unsigned)B), dim3(kThreads), 0, st, C, U, V, (int)N, (int)R,
                       (int)D, sp.rows, tpart);
  else
    hipLaunchKernelGGL(k_bil_root_t, dim3(sp.S, (unsigned)B), dim3(kThreads), lds_a, st, C, U, V, (int)N, (int)R,
                       (int)D, sp.rows, tpart);
  LO_PROF_END(st);
  const int DE = ((int)D + 1) & ~1, DP = DE | 1, RP = ((int)R + 31) & ~31;
  int nw = 4;  // waves = 32-row blocks per workgroup: as many as fit 64 KB of LDS
  size_t lds_o = 0;
  for (; nw >= 1; nw >>= 1) {
    lds_o = sizeof(float) * ((size_t)2 * DE * RP + (size_t)2 * 32 * nw * DP);
    if (lds_o <= 64 * 1024) break;
  }
  if (nw < 1) return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("bil_root_out", st);
  // blocks of 32 nw rows per workgroup: as many as leave ~4 workgroups per CU
  const int nblk = (int)((N + 32 * nw - 1) / (32 * nw));
  int tiles = 1;
  while (tiles < 8 && (int64_t)B * ((nblk + 2 * tiles - 1) / (2 * tiles)) >= 1024) tiles *= 2;
  hipLaunchKernelGGL(k_bil_root_out, dim3((unsigned)((nblk + tiles - 1) / tiles), (unsigned)B), dim3(kThreads), lds_o,
                     st, U, V, tpart, sp.S, (int)N, (int)R, (int)D, nw, tiles, out, rowdot);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

int lo_root_apply_add_f32(const float* U, const float* T, int64_t B, int64_t N, int64_t D, int64_t R, float* out,
                          void* stream) {
  if (!U || !T || !out || B < 1 || N < 1 || D < 1 || R < 1 || B > 65535) return LO_ERR_BADARG;
  if ((R != 8 && R != 16 && R != 32) || D > 32) return LO_ERR_UNSUPPORTED;  // (the caller runs the library product)
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((N + 1023) / 1024), (unsigned)B), block(kThreads);
  LO_PROF_BEGIN("root_apply_add", st);
#define LO_RAA(R_, D_) hipLaunchKernelGGL((k_root_apply_add<R_, D_>), grid, block, 0, st, U, T, (int)N, (int)D, out)
#define LO_RAA_D(R_)                 \
  if (D <= 8) LO_RAA(R_, 8);         \
  else if (D <= 16) LO_RAA(R_, 16);  \
  else LO_RAA(R_, 32)
  if (R == 32) { LO_RAA_D(32); } else if (R == 16) { LO_RAA_D(16); } else { LO_RAA_D(8); }
#undef LO_RAA_D
#undef LO_RAA
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

size_t lo_bilinear_kron_workspace_bytes(int64_t B, int64_t n1, int64_t n2, int64_t D) {
  return sizeof(float) * (size_t)B * n1 * n2 * D + 256;
}

int lo_bilinear_kron_f32(const float* K1, const float* K2, const float* U, const float* V, int64_t B, int64_t n1,
                         int64_t n2, int64_t D, float* dK1, float* dK2, void* ws, size_t ws_bytes, void* stream) {
  if (!K1 || !K2 || !U || !V || !dK1 || !dK2 || !ws || B < 1 || n1 < 1 || n2 < 1 || D < 1) return LO_ERR_BADARG;
  if (ws_bytes < sizeof(float) * (size_t)B * n1 * n2 * D) return LO_ERR_WORKSPACE;
  return kron_bilinear(K1, K2, U, V, (float*)ws, dK1, dK2, B, (int)n1, (int)n2, D, (hipStream_t)stream);
}

}  // extern "C"
