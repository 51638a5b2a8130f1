// lo_lanczos_f64.hip -- the reference's lanczos_tridiag (linear_operator/utils/lanczos.py:9-164) in fp64.
//
// Like lo_cg_f64.hip this is the plain streaming formulation, there so that float64 callers of root_decomposition /
// Lanczos (the reference is dtype-generic) run on the device: the basis [k, B, N, P] lives in HBM in the layout the
// fp32 engine uses, one launch per step of the recurrence, the operator a dense fp64 matrix (+ optional diagonal) or an
// opaque closure.  Full re-orthogonalisation against every stored vector, the reference's signed `inner > tol` test with
// up to ten extra passes (:131-142) and its batch-global stopping rule (:147) -- the two decisions are read back to the
// host once per step, as the reference's own `.item()`-style tests do.
#include "lo_internal.h"

#include <math.h>
#include <string.h>
#include <vector>

namespace lo {
namespace {

// out[j, b, p] = sum_n Q[j, b, n, p] r[b, n, p];  grid (P, B, m)
__global__ __launch_bounds__(kThreads) void k64l_dots(const double* __restrict__ Q, const double* __restrict__ r,
                                                       double* __restrict__ out, int B, int N, int P) {
  __shared__ double s[kThreads];
  const int p = blockIdx.x, b = blockIdx.y, j = blockIdx.z;
  const double* q = Q + (((size_t)j * B + b) * N) * P + p;
  const double* rb = r + ((size_t)b * N) * P + p;
  double t = 0.0;
  for (int i = threadIdx.x; i < N; i += kThreads) t += q[(size_t)i * P] * rb[(size_t)i * P];
  s[threadIdx.x] = t;
  __syncthreads();
  for (int h = kThreads / 2; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) s[threadIdx.x] += s[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[((size_t)j * B + b) * P + p] = s[0];
}

// r[b, n, p] -= sum_{j < m} coef[j, b, p] Q[j, b, n, p]   (the reference sums the projections first, :119-120)
__global__ __launch_bounds__(kThreads) void k64l_project_out(double* __restrict__ r, const double* __restrict__ Q,
                                                              const double* __restrict__ coef, int m, int B, int N,
                                                              int P, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int p = (int)(e % P);
  const size_t b = e / ((size_t)N * P);
  double acc = 0.0;
  for (int j = 0; j < m; ++j) acc += Q[(size_t)j * total + e] * coef[((size_t)j * B + b) * P + p];
  r[e] -= acc;
}

// out[e] = a[e] - coef[b, p] * q[e]  (coef may be NULL: plain copy)
__global__ __launch_bounds__(kThreads) void k64l_sub_scaled(const double* a, const double* __restrict__ q,
                                                             const double* __restrict__ coef, double* out,
                                                             int N, int P, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int p = (int)(e % P);
  const size_t b = e / ((size_t)N * P);
  out[e] = coef ? a[e] - coef[b * P + p] * q[e] : a[e];
}

// nrm[b, p] = sqrt(sq[b, p]);  one thread per (b, p)
__global__ void k64l_sqrt(const double* __restrict__ sq, double* __restrict__ nrm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) nrm[i] = sqrt(sq[i]);
}

// out[e] = a[e] / nrm[b, p]
__global__ __launch_bounds__(kThreads) void k64l_divide(const double* a, const double* __restrict__ nrm,
                                                         double* out, int N, int P, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int p = (int)(e % P);
  const size_t b = e / ((size_t)N * P);
  out[e] = a[e] / nrm[b * P + p];
}

// flags[0] = #{inner[i] > tol, i < n_inner}; flags[1] = #{|beta[i]| > 1e-6, i < n_beta}   (signed compare, :134)
__global__ __launch_bounds__(kThreads) void k64l_flags(const double* __restrict__ inner, int n_inner, double tol,
                                                        const double* __restrict__ beta, int n_beta,
                                                        int* __restrict__ flags) {
  __shared__ int s0[kThreads], s1[kThreads];
  int a = 0, b = 0;
  for (int i = threadIdx.x; i < n_inner; i += kThreads) a += inner[i] > tol;
  for (int i = threadIdx.x; i < n_beta; i += kThreads) b += fabs(beta[i]) > 1e-6;
  s0[threadIdx.x] = a;
  s1[threadIdx.x] = b;
  __syncthreads();
  for (int h = kThreads / 2; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) {
      s0[threadIdx.x] += s0[threadIdx.x + h];
      s1[threadIdx.x] += s1[threadIdx.x + h];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    flags[0] = s0[0];
    flags[1] = s1[0];
  }
}

struct LzLay64 {
  double *r, *tmp, *coef, *sq, *nrm;
  int* flags;
};
void lz_lay64(int64_t B, int64_t N, int64_t P, int max_iter, Arena& ar, LzLay64* l) {
  const size_t V = (size_t)B * N * P, S = (size_t)B * P;
  l->r = ar.take<double>(V);
  l->tmp = ar.take<double>(V);
  l->coef = ar.take<double>(S * (size_t)max_iter);
  l->sq = ar.take<double>(S);
  l->nrm = ar.take<double>(S);
  l->flags = ar.take<int>(4);
}

}  // namespace
}  // namespace lo

using namespace lo;

extern "C" size_t lo_lanczos_f64_workspace_bytes(int64_t B, int64_t N, int64_t P, int32_t max_iter) {
  Arena ar(nullptr, 0);
  LzLay64 l;
  lz_lay64(B, N, P, max_iter, ar, &l);
  return ar.off + 1024;
}

extern "C" int lo_lanczos_tridiag_f64(const double* A, const double* diag, lo_matvec_cb_f64 matvec, void* matvec_user,
                                      const double* init_vecs, int64_t B, int64_t N, int64_t P, int32_t max_iter,
                                      double tol, double* q_mat, double* t_mat, int32_t* iters_out, void* ws,
                                      size_t ws_bytes, void* stream) {
  if (!init_vecs || !q_mat || !t_mat || !iters_out || (!A && !matvec)) return LO_ERR_BADARG;
  if (B < 1 || N < 1 || P < 1 || max_iter < 1) return LO_ERR_BADARG;
  if (B > 65535 || P > 0x7fffffff / 2 || N > 0x7ffffff0 || max_iter > 65535) return LO_ERR_UNSUPPORTED;
  const int K = (int)std::min<int64_t>(max_iter, N);  // lanczos.py:57
  hipStream_t st = (hipStream_t)stream;
  Arena ar(ws, ws_bytes);
  LzLay64 l;
  lz_lay64(B, N, P, max_iter, ar, &l);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const size_t V = (size_t)B * N * P, S = (size_t)B * P;
  const unsigned eg = (unsigned)((V + kThreads - 1) / kThreads);
  const int iB = (int)B, iN = (int)N, iP = (int)P;
  auto qv = [&](int j) { return q_mat + (size_t)j * V; };
  auto tm = [&](int i, int j) { return t_mat + ((size_t)i * max_iter + j) * S; };
  auto apply_op = [&](const double* v, double* y) -> int {
    if (A) return f64_dense_mv(A, diag, v, y, B, N, P, st);
    return matvec(matvec_user, v, y, B, N, P, stream) ? LO_ERR_LAUNCH : LO_OK;
  };
  auto dots = [&](const double* Q, int m, const double* r, double* out) {
    hipLaunchKernelGGL(k64l_dots, dim3((unsigned)P, (unsigned)B, (unsigned)m), dim3(kThreads), 0, st, Q, r, out, iB, iN,
                       iP);
  };
  // nrm = ||a|| per (b, p); out = a / nrm
  auto normalise = [&](const double* a, double* out) {
    dots(a, 1, a, l.sq);
    hipLaunchKernelGGL(k64l_sqrt, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, l.sq, l.nrm, (int)S);
    hipLaunchKernelGGL(k64l_divide, dim3(eg), dim3(kThreads), 0, st, a, l.nrm, out, iN, iP, V);
  };
  auto copy_small = [&](double* dst, const double* src) -> int {
    LO_HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(double) * S, hipMemcpyDeviceToDevice, st));
    return LO_OK;
  };
  int rc;
  LO_HIP_CHECK(hipMemsetAsync(t_mat, 0, sizeof(double) * (size_t)max_iter * max_iter * S, st));

  normalise(init_vecs, qv(0));  // :81
  if ((rc = apply_op(qv(0), l.r))) return rc;  // :85
  dots(qv(0), 1, l.r, l.coef);  // alpha_0
  if ((rc = copy_small(tm(0, 0), l.coef))) return rc;
  hipLaunchKernelGGL(k64l_sub_scaled, dim3(eg), dim3(kThreads), 0, st, l.r, qv(0), l.coef, l.r, iN, iP, V);  // :89
  LO_LAUNCH_CHECK();
  if (K > 1) {
    normalise(l.r, qv(1));  // :98
    if ((rc = copy_small(tm(0, 1), l.nrm)) || (rc = copy_small(tm(1, 0), l.nrm))) return rc;
  }
  int k = 0;
  for (k = 1; k < K; ++k) {  // :101
    if ((rc = apply_op(qv(k), l.tmp))) return rc;
    hipLaunchKernelGGL(k64l_sub_scaled, dim3(eg), dim3(kThreads), 0, st, l.tmp, qv(k - 1), tm(k, k - 1), l.r, iN, iP,
                       V);  // :108
    dots(qv(k), 1, l.r, l.coef);  // alpha_k
    if ((rc = copy_small(tm(k, k), l.coef))) return rc;
    if (k + 1 >= K) break;  // :114
    hipLaunchKernelGGL(k64l_sub_scaled, dim3(eg), dim3(kThreads), 0, st, l.r, qv(k), l.coef, l.r, iN, iP, V);
    const int m = k + 1;
    dots(q_mat, m, l.r, l.coef);  // :118
    hipLaunchKernelGGL(k64l_project_out, dim3(eg), dim3(kThreads), 0, st, l.r, q_mat, l.coef, m, iB, iN, iP, V);
    normalise(l.r, l.r);
    if ((rc = copy_small(tm(k, k + 1), l.nrm)) || (rc = copy_small(tm(k + 1, k), l.nrm))) return rc;
    LO_LAUNCH_CHECK();
    int h[2];
    bool could = false;
    for (int pass = 0; pass < 10; ++pass) {  // :133-142
      dots(q_mat, m, l.r, l.coef);           // inner products with the stored basis
      hipLaunchKernelGGL(k64l_flags, dim3(1), dim3(kThreads), 0, st, l.coef, (int)((size_t)m * S), tol, tm(k, k + 1),
                         (int)S, l.flags);
      LO_LAUNCH_CHECK();
      LO_HIP_CHECK(hipMemcpyAsync(h, l.flags, sizeof(h), hipMemcpyDeviceToHost, st));
      LO_HIP_CHECK(hipStreamSynchronize(st));
      if (!h[0]) {
        could = true;
        break;
      }
      hipLaunchKernelGGL(k64l_project_out, dim3(eg), dim3(kThreads), 0, st, l.r, q_mat, l.coef, m, iB, iN, iP, V);
      normalise(l.r, l.r);
    }
    hipLaunchKernelGGL(k64l_sub_scaled, dim3(eg), dim3(kThreads), 0, st, l.r, (const double*)nullptr,
                       (const double*)nullptr, qv(k + 1), iN, iP, V);  // :145
    LO_LAUNCH_CHECK();
    if (h[1] == 0 || !could) break;  // :147
  }
  if (k >= K) k = K - 1;
  *iters_out = k + 1;  // :151
  return LO_OK;
}
