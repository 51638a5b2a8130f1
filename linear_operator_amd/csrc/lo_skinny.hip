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
// Thread map (256 threads = 4 waves): a row of A is RQ float4 "quads" (RQ = padded rank / 4, compile-time
// power of two <= 64); lane l of a wave owns quad q = l % RQ of row group g = l / RQ, so one load
// instruction of a wave reads G = 64 / RQ consecutive rows = 1 KiB of contiguous HBM, four loads in flight.
//
// Fused vector updates (CG, linear_cg.py line numbers): the tn kernel can build its input vector on the fly
//   VMODE 1:  v = z + beta * p_old            -> p        (:46, update of the search direction)
//   VMODE 2:  v = r - alpha * Ap ; x += alpha p ; rr = sum v^2   (:254-264, :31, :298) with the masked alpha
//             computed in the prologue from the pAp partials
// so the elementwise CG kernels disappear from the preconditioned low-rank loop.
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {


template <int CT, int RQ, int VMODE>
__global__ __launch_bounds__(kThreads) void k_skinny_tn(const float* __restrict__ A, float* __restrict__ v, int ldv,
                                                         int c, float* __restrict__ tpart, int N, int rows_per,
                                                         TnFuse f, const int* __restrict__ stop) {
  if (stop && *stop) return;
  constexpr int lda = 4 * RQ;
  constexpr int slots = kThreads / RQ;
  __shared__ float4 red4[kThreads];
  __shared__ float coef_s[8];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int r0 = s * rows_per;
  const int r1 = min(N, r0 + rows_per);
  const int q = threadIdx.x % RQ;
  const int slot = threadIdx.x / RQ;
  const float* Ab = A + (size_t)b * N * lda + 4 * q;

  float coef[CT];  // beta (VMODE 1) or alpha (VMODE 2) per column
#pragma unroll
  for (int k = 0; k < CT; ++k) coef[k] = 0.f;
  if (VMODE == 1) {
#pragma unroll
    for (int k = 0; k < CT; ++k) coef[k] = (k < c && !f.first) ? f.beta[(size_t)b * ldv + k] : 0.f;
  }
  if (VMODE == 2) {
    if (threadIdx.x < c) {
      const int k = threadIdx.x;
      float pAp = 0.f;
      for (int ss = 0; ss < f.S_dot; ++ss) pAp += f.pAp_part[((size_t)b * f.S_dot + ss) * ldv + k];  // :250-251
      const float rz = f.rz[(size_t)b * ldv + k];
      float a = (pAp < f.eps) ? 0.f : rz / pAp;          // safe division :254-257
      if (f.has_conv[(size_t)b * ldv + k]) a = 0.f;       // :260
      coef_s[k] = a;
      if (s == 0) f.alpha_out[(size_t)b * ldv + k] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CT; ++k) coef[k] = (k < c) ? coef_s[k] : 0.f;
  }

  float acc[4][CT];
  float rr[CT];
#pragma unroll
  for (int k = 0; k < CT; ++k) {
    rr[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][k] = 0.f;
  }

  auto load_v = [&](int row, float (&pv)[CT]) {
    const size_t o = (size_t)b * N * ldv + (size_t)row * ldv;
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      if (k < c) {
        if (VMODE == 0) {
          pv[k] = v[o + k];
        } else if (VMODE == 1) {
          const float zz = f.z[o + k];
          const float nv = f.first ? zz : fmaf(v[o + k], coef[k], zz);  // p.mul_(beta).add_(z)  :46
          pv[k] = nv;
          if (q == 0) v[o + k] = nv;
        } else {
          const float nv = fmaf(-coef[k], f.Ap[o + k], v[o + k]);        // r - alpha * Ap        :264
          pv[k] = nv;
          if (q == 0) {
            v[o + k] = nv;
            f.x[o + k] = fmaf(coef[k], f.p[o + k], f.x[o + k]);          // x + alpha * p         :31
            rr[k] = fmaf(nv, nv, rr[k]);
          }
        }
      } else {
        pv[k] = 0.f;
      }
    }
  };

  int row = r0 + slot;
  for (; row + 3 * slots < r1; row += 4 * slots) {  // 4 rows in flight per lane
    float4 a[4];
    float pv[4][CT];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(Ab + (size_t)(row + u * slots) * lda);
#pragma unroll
    for (int u = 0; u < 4; ++u) load_v(row + u * slots, pv[u]);
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
    float pv[CT];
    load_v(row, pv);
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      acc[0][k] = fmaf(a.x, pv[k], acc[0][k]);
      acc[1][k] = fmaf(a.y, pv[k], acc[1][k]);
      acc[2][k] = fmaf(a.z, pv[k], acc[2][k]);
      acc[3][k] = fmaf(a.w, pv[k], acc[3][k]);
    }
  }

  // reduce over the row slots (fixed tree), one column at a time
  float* out = tpart + ((size_t)b * S + s) * (size_t)lda * c;
#pragma unroll
  for (int k = 0; k < CT; ++k) {
    if (k < c) {
      __syncthreads();
      red4[threadIdx.x] = make_float4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]);
      __syncthreads();
#pragma unroll
      for (int h = slots >> 1; h >= 1; h >>= 1) {
        if (slot < h) {
          const float4 o = red4[threadIdx.x + h * RQ];
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
  if (VMODE == 2) {
    float* red = reinterpret_cast<float*>(red4);
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      if (k < c) {
        const float tot = block_sum256(q == 0 ? rr[k] : 0.f, red);
        if (threadIdx.x == 0) f.rr_part[((size_t)b * S + s) * ldv + k] = tot;
      }
    }
  }
}

// y = sgn * A t + dd o v over contiguous row tiles: a wave owns TILE = 4 * (64 / RQ) consecutive rows per
// step; after the butterfly over the RQ lanes of a row, lanes re-distribute so that the epilogue (diagonal
// term, store, fused dot) touches each row once with consecutive addresses.
template <int CT, int RQ, bool DOT>
__global__ __launch_bounds__(kThreads) void k_skinny_nn(const float* __restrict__ A, const float* __restrict__ tpart,
                                                         const float* __restrict__ dd, int dd_mode, float sgn,
                                                         const float* __restrict__ v, int ldv, int c,
                                                         float* __restrict__ y, float* __restrict__ dot_part, int ldd,
                                                         int N, int rows_per, const int* __restrict__ stop) {
  if (stop && *stop) return;
  constexpr int lda = 4 * RQ;
  constexpr int G = 64 / RQ;        // rows per load instruction of a wave
  constexpr int TILE = 4 * G;       // rows per wave step
  constexpr int UPL = (RQ >= 4) ? 1 : 4 / RQ;  // row groups a lane finishes in the epilogue
  extern __shared__ float t_s[];    // [4*RQ][c]
  __shared__ float red[kThreads];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  for (int idx = threadIdx.x; idx < lda * c; idx += kThreads) {  // t = sum of the S partials, fixed order
    const float* tp = tpart + (size_t)b * S * lda * c + idx;
    float acc = 0.f;
    for (int ss = 0; ss < S; ++ss) acc += tp[(size_t)ss * lda * c];
    t_s[idx] = acc;
  }
  __syncthreads();

  const int r0 = s * rows_per;
  const int r1 = min(N, r0 + rows_per);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane % RQ, g = lane / RQ;
  const float* Ab = A + (size_t)b * N * lda + 4 * q;
  const float* vb = v + (size_t)b * N * ldv;
  float* yb = y + (size_t)b * N * ldv;
  const float* ddb = (dd_mode == LO_DIAG_FULL) ? dd + (size_t)b * N : dd;
  const float ddc = (dd_mode == LO_DIAG_CONST) ? dd[b] : 0.f;

  float treg[4][CT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < CT; ++k) treg[j][k] = (k < c) ? t_s[(4 * q + j) * c + k] : 0.f;

  float dacc[CT];
#pragma unroll
  for (int k = 0; k < CT; ++k) dacc[k] = 0.f;

  for (int base = r0 + wave * TILE; base < r1; base += 4 * TILE) {
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = base + u * G + g;
      a[u] = (rr < r1) ? *reinterpret_cast<const float4*>(Ab + (size_t)rr * lda) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float part[4][CT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < CT; ++k) {
        float p = a[u].x * treg[0][k];
        p = fmaf(a[u].y, treg[1][k], p);
        p = fmaf(a[u].z, treg[2][k], p);
        p = fmaf(a[u].w, treg[3][k], p);
        part[u][k] = p;
      }
#pragma unroll
    for (int off = 1; off < RQ; off <<= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < CT; ++k) part[u][k] += __shfl_xor(part[u][k], off, 64);
    // epilogue: lane (g, q) finishes row group(s) u = q, q + RQ, ... (< 4)
#pragma unroll
    for (int e = 0; e < UPL; ++e) {
      const int u_sel = (RQ >= 4) ? (q & 3) : (q + e * RQ);
      const bool active = (RQ >= 4) ? (q < 4) : true;
      const int rr = base + u_sel * G + g;
      if (active && rr < r1) {
        const float dv = (dd_mode == LO_DIAG_FULL) ? ddb[rr] : ddc;
#pragma unroll
        for (int k = 0; k < CT; ++k) {
          if (k < c) {
            const float pk =
                (u_sel == 0) ? part[0][k] : (u_sel == 1) ? part[1][k] : (u_sel == 2) ? part[2][k] : part[3][k];
            const float vin = vb[(size_t)rr * ldv + k];
            const float yv = fmaf(dv, vin, sgn * pk);
            yb[(size_t)rr * ldv + k] = yv;
            if (DOT) dacc[k] = fmaf(vin, yv, dacc[k]);
          }
        }
      }
    }
  }

  if (DOT) {
    float* dp = dot_part + ((size_t)b * S + s) * ldd;
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      if (k < c) {
        const float tot = block_sum256(dacc[k], red);
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

static bool rq_ok(int R4) {
  const int RQ = R4 / 4;
  return R4 % 4 == 0 && RQ >= 1 && RQ <= 64 && (RQ & (RQ - 1)) == 0;
}

// ---- launch helpers: runtime (cn, RQ, mode) -> template instance -------------------------------------
template <int CT, int VMODE>
static void launch_tn_rq(int RQ, dim3 grid, hipStream_t st, const float* A, float* v, int ldv, int cn, float* tp, int N,
                         int rows, const TnFuse& f, const int* stop) {
  dim3 block(kThreads);
#define LO_TN(RQV) \
  hipLaunchKernelGGL((k_skinny_tn<CT, RQV, VMODE>), grid, block, 0, st, A, v, ldv, cn, tp, N, rows, f, stop)
  switch (RQ) {
    case 1: LO_TN(1); break;
    case 2: LO_TN(2); break;
    case 4: LO_TN(4); break;
    case 8: LO_TN(8); break;
    case 16: LO_TN(16); break;
    case 32: LO_TN(32); break;
    default: LO_TN(64); break;
  }
#undef LO_TN
}

template <int VMODE>
static int skinny_tn_impl(const float* A, int lda, int R4, float* v, int64_t c, float* tpart, int64_t B, int64_t N,
                          Split sp, const TnFuse& f0, const int* stop, hipStream_t st) {
  if (!rq_ok(R4) || lda != R4 || c < 1) return LO_ERR_BADARG;
  if (skinny_mfma_ok(R4, c)) return skinny_tn_mfma(VMODE, A, R4, v, c, tpart, B, N, sp, f0, stop, st);
  const int RQ = R4 / 4;
  dim3 grid(sp.S, (unsigned)B);
  for (int64_t c0 = 0; c0 < c; c0 += 8) {
    const int cn = (int)std::min<int64_t>(8, c - c0);
    float* tp = tpart + (size_t)B * sp.S * R4 * c0;
    float* vp = v + c0;
    TnFuse f = f0;
    if (VMODE == 1) {
      f.z += c0;
      f.beta += c0;
    }
    if (VMODE == 2) {
      f.Ap += c0; f.p += c0; f.x += c0; f.pAp_part += c0; f.rz += c0; f.has_conv += c0; f.alpha_out += c0;
      f.rr_part += c0;
    }
    LO_PROF_BEGIN(tn_name(R4), st);
    if (cn == 1) launch_tn_rq<1, VMODE>(RQ, grid, st, A, vp, (int)c, cn, tp, (int)N, sp.rows, f, stop);
    else if (cn == 2) launch_tn_rq<2, VMODE>(RQ, grid, st, A, vp, (int)c, cn, tp, (int)N, sp.rows, f, stop);
    else if (cn <= 4) launch_tn_rq<4, VMODE>(RQ, grid, st, A, vp, (int)c, cn, tp, (int)N, sp.rows, f, stop);
    else launch_tn_rq<8, VMODE>(RQ, grid, st, A, vp, (int)c, cn, tp, (int)N, sp.rows, f, stop);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

// Host wrappers: columns are processed in chunks of <= 8 (register tile of the VALU kernels); chunk j of
// tpart lives at tpart + B*S*R4*c0 with its own [B,S,R4,cn] layout.
int skinny_tn(const float* A, int lda, int R4, const float* v, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
              const int* stop, hipStream_t st) {
  TnFuse f{};
  return skinny_tn_impl<0>(A, lda, R4, const_cast<float*>(v), c, tpart, B, N, sp, f, stop, st);
}

int skinny_tn_pupdate(const float* A, int lda, int R4, float* p, const float* z, const float* beta, int first, int64_t c,
                      float* tpart, int64_t B, int64_t N, Split sp, const int* stop, hipStream_t st) {
  TnFuse f{};
  f.z = z;
  f.beta = beta;
  f.first = first;
  return skinny_tn_impl<1>(A, lda, R4, p, c, tpart, B, N, sp, f, stop, st);
}

int skinny_tn_rupdate(const float* A, int lda, int R4, float* r, const float* Ap, const float* p, float* x,
                      const float* pAp_part, int S_dot, const float* rz, const int* has_conv, float eps,
                      float* alpha_out, float* rr_part, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
                      const int* stop, hipStream_t st) {
  TnFuse f{};
  f.Ap = Ap; f.p = p; f.x = x; f.pAp_part = pAp_part; f.S_dot = S_dot; f.rz = rz; f.has_conv = has_conv; f.eps = eps;
  f.alpha_out = alpha_out; f.rr_part = rr_part;
  return skinny_tn_impl<2>(A, lda, R4, r, c, tpart, B, N, sp, f, stop, st);
}

template <int CT>
static void launch_nn_rq(int RQ, bool dot, dim3 grid, size_t shm, hipStream_t st, const float* A, const float* tp,
                         const float* dd, int dd_mode, float sgn, const float* v, int ldv, int cn, float* y, float* dp,
                         int ldd, int N, int rows, const int* stop) {
  dim3 block(kThreads);
#define LO_NN(RQV)                                                                                                  \
  do {                                                                                                              \
    if (dot)                                                                                                        \
      hipLaunchKernelGGL((k_skinny_nn<CT, RQV, true>), grid, block, shm, st, A, tp, dd, dd_mode, sgn, v, ldv, cn, y, \
                         dp, ldd, N, rows, stop);                                                                   \
    else                                                                                                            \
      hipLaunchKernelGGL((k_skinny_nn<CT, RQV, false>), grid, block, shm, st, A, tp, dd, dd_mode, sgn, v, ldv, cn,  \
                         y, dp, ldd, N, rows, stop);                                                                \
  } while (0)
  switch (RQ) {
    case 1: LO_NN(1); break;
    case 2: LO_NN(2); break;
    case 4: LO_NN(4); break;
    case 8: LO_NN(8); break;
    case 16: LO_NN(16); break;
    case 32: LO_NN(32); break;
    default: LO_NN(64); break;
  }
#undef LO_NN
}

int skinny_nn(const float* A, int lda, int R4, const float* tpart, const float* dd, int dd_mode, float sgn,
              const float* v, int64_t c, float* y, float* dot_part, int64_t B, int64_t N, Split sp, const int* stop,
              hipStream_t st) {
  if (!rq_ok(R4) || lda != R4 || c < 1) return LO_ERR_BADARG;
  if (skinny_mfma_ok(R4, c))
    return skinny_nn_mfma(A, R4, tpart, dd, dd_mode, sgn, v, c, y, dot_part, B, N, sp, stop, st);
  const int RQ = R4 / 4;
  dim3 grid(sp.S, (unsigned)B);
  for (int64_t c0 = 0; c0 < c; c0 += 8) {
    const int cn = (int)std::min<int64_t>(8, c - c0);
    const float* tp = tpart + (size_t)B * sp.S * R4 * c0;
    const float* vp = v + c0;
    float* yp = y + c0;
    float* dp = dot_part ? dot_part + c0 : nullptr;
    const size_t shm = (size_t)R4 * cn * sizeof(float);
    const bool dot = dp != nullptr;
    LO_PROF_BEGIN(nn_name(R4), st);
    if (cn == 1)
      launch_nn_rq<1>(RQ, dot, grid, shm, st, A, tp, dd, dd_mode, sgn, vp, (int)c, cn, yp, dp, (int)c, (int)N, sp.rows, stop);
    else if (cn == 2)
      launch_nn_rq<2>(RQ, dot, grid, shm, st, A, tp, dd, dd_mode, sgn, vp, (int)c, cn, yp, dp, (int)c, (int)N, sp.rows, stop);
    else if (cn <= 4)
      launch_nn_rq<4>(RQ, dot, grid, shm, st, A, tp, dd, dd_mode, sgn, vp, (int)c, cn, yp, dp, (int)c, (int)N, sp.rows, stop);
    else
      launch_nn_rq<8>(RQ, dot, grid, shm, st, A, tp, dd, dd_mode, sgn, vp, (int)c, cn, yp, dp, (int)c, (int)N, sp.rows, stop);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

}  // namespace lo
