// lo_skinny.hip -- the two "skinny operand" kernels behind
//   * the low-rank-plus-diagonal matvec  y = C (C^T v) + d o v
//       (reference: AddedDiagLinearOperator._matmul added_diag_linear_operator.py:72-76 over
//        RootLinearOperator._matmul root_linear_operator.py:68-72), and
//   * the Woodbury/QR preconditioner apply  z = r o dinv - Q (Q^T r)
//       (reference: precondition_closure added_diag_linear_operator.py:135-140).
// Both are  y = sgn * A (A^T v) + dd o v  with a tall skinny A [N, R]; the pass structure is
//   skinny_tn : tpart[b,s] = A[rows_s]^T v[rows_s]          (HBM-bound stream of A, 16 B / lane)
//   skinny_nn : y = sgn * A (sum_s tpart[b,s]) + dd o v     (second stream of A; fused dot v.y)
// A member's rows are split over S workgroups (grid = S x B) so that small batches still fill the
// 256 CUs; the S partial t's are summed in a fixed order by the consumer (bitwise reproducible).
//
// Thread map (256 threads): a row of A is RQ float4 "quads"; thread t owns quad q = t % RQ of row slot
// t / RQ, so a wave reads 64 consecutive float4 = 1 KiB of contiguous HBM per load instruction, four
// loads in flight per lane.  RQ is a power of two <= 64 (rank padded to 4*RQ with zero columns).
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

template <int CT>
__global__ __launch_bounds__(kThreads) void k_skinny_tn(const float* __restrict__ A, int lda, int RQ,
                                                         const float* __restrict__ v, int ldv, int c,
                                                         float* __restrict__ tpart, int N, int rows_per,
                                                         const int* __restrict__ stop) {
  if (stop && *stop) return;
  __shared__ float4 red4[kThreads];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int r0 = s * rows_per;
  const int r1 = min(N, r0 + rows_per);
  const int slots = kThreads / RQ;
  const int q = threadIdx.x % RQ;
  const int slot = threadIdx.x / RQ;
  const float* Ab = A + (size_t)b * N * lda + 4 * q;
  const float* vb = v + (size_t)b * N * ldv;

  float acc[4][CT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[j][k] = 0.f;

  int row = r0 + slot;
  // 4 rows in flight per lane
  for (; row + 3 * slots < r1; row += 4 * slots) {
    float4 a[4];
    float pv[4][CT];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(Ab + (size_t)(row + u * slots) * lda);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < CT; ++k) pv[u][k] = (k < c) ? vb[(size_t)(row + u * slots) * ldv + k] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < CT; ++k) {
        acc[0][k] = fmaf(a[u].x, pv[u][k], acc[0][k]);
        acc[1][k] = fmaf(a[u].y, pv[u][k], acc[1][k]);
        acc[2][k] = fmaf(a[u].z, pv[u][k], acc[2][k]);
        acc[3][k] = fmaf(a[u].w, pv[u][k], acc[3][k]);
      }
  }
  for (; row < r1; row += slots) {
    const float4 a = *reinterpret_cast<const float4*>(Ab + (size_t)row * lda);
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      const float p = (k < c) ? vb[(size_t)row * ldv + k] : 0.f;
      acc[0][k] = fmaf(a.x, p, acc[0][k]);
      acc[1][k] = fmaf(a.y, p, acc[1][k]);
      acc[2][k] = fmaf(a.z, p, acc[2][k]);
      acc[3][k] = fmaf(a.w, p, acc[3][k]);
    }
  }

  // reduce over the row slots (fixed tree), one column at a time
  int hpow = 1;
  while (hpow < slots) hpow <<= 1;
  float* out = tpart + ((size_t)b * S + s) * (size_t)(4 * RQ) * c;
#pragma unroll
  for (int k = 0; k < CT; ++k) {
    if (k < c) {
      __syncthreads();
      red4[threadIdx.x] = make_float4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]);
      __syncthreads();
      for (int h = hpow >> 1; h >= 1; h >>= 1) {
        if (slot < h && slot + h < slots) {
          float4 o = red4[threadIdx.x + h * RQ];
          float4 m = red4[threadIdx.x];
          m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
          red4[threadIdx.x] = m;
        }
        __syncthreads();
      }
      if (slot == 0) {
        const float4 m = red4[threadIdx.x];
        out[(size_t)(4 * q + 0) * c + k] = m.x;
        out[(size_t)(4 * q + 1) * c + k] = m.y;
        out[(size_t)(4 * q + 2) * c + k] = m.z;
        out[(size_t)(4 * q + 3) * c + k] = m.w;
      }
    }
  }
}

template <int CT, bool DOT>
__global__ __launch_bounds__(kThreads) void k_skinny_nn(const float* __restrict__ A, int lda, int RQ,
                                                         const float* __restrict__ tpart,
                                                         const float* __restrict__ dd, int dd_mode, float sgn,
                                                         const float* __restrict__ v, int ldv, int c,
                                                         float* __restrict__ y, float* __restrict__ dot_part,
                                                         int ldd, int N, int rows_per,
                                                         const int* __restrict__ stop) {
  if (stop && *stop) return;
  extern __shared__ float t_s[];  // [4*RQ][c]
  __shared__ float red[kThreads];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int R4 = 4 * RQ;
  // t = sum over the S partials, fixed order
  for (int idx = threadIdx.x; idx < R4 * c; idx += kThreads) {
    const float* tp = tpart + (size_t)b * S * R4 * c + idx;
    float acc = 0.f;
    for (int ss = 0; ss < S; ++ss) acc += tp[(size_t)ss * R4 * c];
    t_s[idx] = acc;
  }
  __syncthreads();

  const int r0 = s * rows_per;
  const int r1 = min(N, r0 + rows_per);
  const int slots = kThreads / RQ;
  const int q = threadIdx.x % RQ;
  const int slot = threadIdx.x / RQ;
  const float* Ab = A + (size_t)b * N * lda + 4 * q;
  const float* vb = v + (size_t)b * N * ldv;
  float* yb = y + (size_t)b * N * ldv;
  unsigned mine = 0;  // bit k set <=> this lane owns column k (k % RQ == q)
#pragma unroll
  for (int k = 0; k < CT; ++k) mine |= (((k & (RQ - 1)) == q) ? 1u : 0u) << k;
  const float* ddb = (dd_mode == LO_DIAG_FULL) ? dd + (size_t)b * N : dd;
  const float ddc = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;

  float treg[4][CT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < CT; ++k) treg[j][k] = (k < c) ? t_s[(4 * q + j) * c + k] : 0.f;

  constexpr int NM = (CT + 0) ;  // upper bound on columns a lane may own (RQ >= 1)
  float dacc[NM];
#pragma unroll
  for (int k = 0; k < NM; ++k) dacc[k] = 0.f;

  auto do_row = [&](int row, const float4& a) {
    float part[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      float p = a.x * treg[0][k];
      p = fmaf(a.y, treg[1][k], p);
      p = fmaf(a.z, treg[2][k], p);
      p = fmaf(a.w, treg[3][k], p);
      part[k] = p;
    }
    for (int off = 1; off < RQ; off <<= 1) {
#pragma unroll
      for (int k = 0; k < CT; ++k) part[k] += __shfl_xor(part[k], off, 64);
    }
    const float dv = (dd_mode == LO_DIAG_FULL) ? ddb[row] : ddc;
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      if (k < c && ((mine >> k) & 1u)) {  // lane q of the row's RQ lanes owns columns q, q+RQ, ...
        const float vin = vb[(size_t)row * ldv + k];
        const float yv = fmaf(dv, vin, sgn * part[k]);
        yb[(size_t)row * ldv + k] = yv;
        if (DOT) dacc[k] = fmaf(vin, yv, dacc[k]);
      }
    }
  };

  int row = r0 + slot;
  // NOTE: every lane of a wave must execute the shuffles -> rows beyond r1 are clamped, not skipped
  const int nrow_iter = (r1 - r0 + slots - 1) / slots;
  int it = 0;
  for (; it + 3 < nrow_iter; it += 4, row += 4 * slots) {
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = row + u * slots;
      a[u] = (rr < r1) ? *reinterpret_cast<const float4*>(Ab + (size_t)rr * lda) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = row + u * slots;
      if (rr < r1) do_row(rr, a[u]);  // slots*RQ == 256 and RQ | 64 -> (rr < r1) is uniform per RQ-lane group
    }
  }
  for (; it < nrow_iter; ++it, row += slots) {
    if (row < r1) {
      const float4 a = *reinterpret_cast<const float4*>(Ab + (size_t)row * lda);
      do_row(row, a);
    }
  }

  if (DOT) {
    float* dp = dot_part + ((size_t)b * S + s) * ldd;
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      if (k < c) {
        const float tot = block_sum256(((mine >> k) & 1u) ? dacc[k] : 0.f, red);
        if (threadIdx.x == 0) dp[k] = tot;
      }
    }
  }
}

__global__ void k_pad_rows(const float* __restrict__ src, int R, float* __restrict__ dst, int R4, int64_t rows) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = rows * R4;
  for (; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / R4;
    const int col = (int)(i % R4);
    dst[i] = (col < R) ? src[row * R + col] : 0.f;
  }
}

int pad_rows(const float* src, int R, float* dst, int R4, int64_t rows, hipStream_t st) {
  const int64_t total = rows * R4;
  int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_pad_rows, dim3(grid), dim3(256), 0, st, src, R, dst, R4, rows);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

static const char* tn_name(int R4) {
  switch (R4) {
    case 4: return "skinny_tn_R4";
    case 8: return "skinny_tn_R8";
    case 16: return "skinny_tn_R16";
    case 32: return "skinny_tn_R32";
    case 64: return "skinny_tn_R64";
    case 128: return "skinny_tn_R128";
    default: return "skinny_tn_R256";
  }
}
static const char* nn_name(int R4) {
  switch (R4) {
    case 4: return "skinny_nn_R4";
    case 8: return "skinny_nn_R8";
    case 16: return "skinny_nn_R16";
    case 32: return "skinny_nn_R32";
    case 64: return "skinny_nn_R64";
    case 128: return "skinny_nn_R128";
    default: return "skinny_nn_R256";
  }
}

// A wave's shuffle groups must not straddle row-validity: RQ | 64 guarantees the RQ lanes of one row
// sit in one wave.
static bool rq_ok(int R4) {
  const int RQ = R4 / 4;
  return R4 % 4 == 0 && RQ >= 1 && RQ <= 64 && (RQ & (RQ - 1)) == 0;
}

// Host wrappers: columns are processed in chunks of <= 8 (register tile of the VALU kernels); chunk j of
// tpart lives at tpart + B*S*R4*c0 with its own [B,S,R4,cn] layout.
int skinny_tn(const float* A, int lda, int R4, const float* v, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
              const int* stop, hipStream_t st) {
  if (!rq_ok(R4) || lda % 4 != 0 || c < 1) return LO_ERR_BADARG;
  const int RQ = R4 / 4;
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  for (int64_t c0 = 0; c0 < c; c0 += 8) {
    const int cn = (int)std::min<int64_t>(8, c - c0);
    float* tp = tpart + (size_t)B * sp.S * R4 * c0;
    const float* vp = v + c0;
#define LO_TN(CT) \
  hipLaunchKernelGGL((k_skinny_tn<CT>), grid, block, 0, st, A, lda, RQ, vp, (int)c, cn, tp, (int)N, sp.rows, stop)
    LO_PROF_BEGIN(tn_name(R4), st);
    if (cn == 1) LO_TN(1);
    else if (cn == 2) LO_TN(2);
    else if (cn <= 4) LO_TN(4);
    else LO_TN(8);
#undef LO_TN
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

int skinny_nn(const float* A, int lda, int R4, const float* tpart, const float* dd, int dd_mode, float sgn,
              const float* v, int64_t c, float* y, float* dot_part, int64_t B, int64_t N, Split sp, const int* stop,
              hipStream_t st) {
  if (!rq_ok(R4) || lda % 4 != 0 || c < 1) return LO_ERR_BADARG;
  const int RQ = R4 / 4;
  dim3 grid(sp.S, (unsigned)B), block(kThreads);
  for (int64_t c0 = 0; c0 < c; c0 += 8) {
    const int cn = (int)std::min<int64_t>(8, c - c0);
    const float* tp = tpart + (size_t)B * sp.S * R4 * c0;
    const float* vp = v + c0;
    float* yp = y + c0;
    float* dp = dot_part ? dot_part + c0 : nullptr;
    const size_t shm = (size_t)R4 * cn * sizeof(float);
#define LO_NN(CT)                                                                                                  \
  do {                                                                                                             \
    if (dp)                                                                                                        \
      hipLaunchKernelGGL((k_skinny_nn<CT, true>), grid, block, shm, st, A, lda, RQ, tp, dd, dd_mode, sgn, vp,      \
                         (int)c, cn, yp, dp, (int)c, (int)N, sp.rows, stop);                                       \
    else                                                                                                           \
      hipLaunchKernelGGL((k_skinny_nn<CT, false>), grid, block, shm, st, A, lda, RQ, tp, dd, dd_mode, sgn, vp,     \
                         (int)c, cn, yp, dp, (int)c, (int)N, sp.rows, stop);                                       \
  } while (0)
    LO_PROF_BEGIN(nn_name(R4), st);
    if (cn == 1) LO_NN(1);
    else if (cn == 2) LO_NN(2);
    else if (cn <= 4) LO_NN(4);
    else LO_NN(8);
#undef LO_NN
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

}  // namespace lo
