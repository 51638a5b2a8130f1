// lo_cg_f64.hip -- the reference's linear_cg (linear_operator/utils/linear_cg.py:98-359) in fp64.
//
// The hot path of north_star is fp32; this engine exists so that the reference's own fp64 recipes (every case of its
// test/utils/test_linear_cg.py builds float64 operands) run unmodified.  It is the plain streaming formulation: all
// vectors [B, N, c] in HBM, one launch per step of the recurrence, the operator either a dense fp64 matrix (+ optional
// diagonal) multiplied by k64_dense_mv or an opaque closure called back once per iteration; the preconditioner is a
// closure or absent.  Same masking of alpha / beta, same stopping rule and the same CG-coefficient tridiagonals as the
// fp32 engine (lo_cg.hip); none of the operator-resident machinery.
#include "lo_internal.h"

#include <algorithm>
#include <math.h>
#include <string.h>

namespace lo {

struct Ctl64 {
  int stop;
  int iterations;
  int tol_reached;
  int nan_detected;
  int skipped;
  int last_tridiag_iter;
  int tri_disabled;
  int pad;
  double mean_resid;
};

// out1[b, j] = sum_i a[b, i, j] * b1[b, i, j]   (and out2 from (a2, b2) when given); grid (c, B)
__global__ __launch_bounds__(kThreads) void k64_dots(const double* __restrict__ a, const double* __restrict__ b1,
                                                      double* __restrict__ out1, const double* __restrict__ a2,
                                                      const double* __restrict__ b2, double* __restrict__ out2, int N,
                                                      int c) {
  __shared__ double s1[kThreads], s2[kThreads];
  const int j = blockIdx.x;
  const size_t base = (size_t)blockIdx.y * N * c + j;
  double t1 = 0.0, t2 = 0.0;
  for (int i = threadIdx.x; i < N; i += kThreads) {
    const size_t e = base + (size_t)i * c;
    t1 += a[e] * b1[e];
    if (a2) t2 += a2[e] * b2[e];
  }
  s1[threadIdx.x] = t1;
  s2[threadIdx.x] = t2;
  __syncthreads();
  for (int h = kThreads / 2; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) {
      s1[threadIdx.x] += s1[threadIdx.x + h];
      s2[threadIdx.x] += s2[threadIdx.x + h];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out1[(size_t)blockIdx.y * c + j] = s1[0];
    if (a2) out2[(size_t)blockIdx.y * c + j] = s2[0];
  }
}

// y[b, i, j] = sum_k A[b, i, k] v[b, k, j] (+ d[b, i] v[b, i, j]); one wave per row, 8 columns per pass
__global__ __launch_bounds__(kThreads) void k64_dense_mv(const double* __restrict__ A, const double* __restrict__ d,
                                                          const double* __restrict__ v, double* __restrict__ y, int N,
                                                          int c) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const size_t b = blockIdx.y;
  const double* Ar = A + (b * N + row) * (size_t)N;
  const double* vb = v + b * (size_t)N * c;
  for (int j0 = 0; j0 < c; j0 += 8) {
    const int nj = min(8, c - j0);
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    for (int k = lane; k < N; k += 64) {
      const double av = Ar[k];
      const double* vr = vb + (size_t)k * c + j0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nj) acc[j] += av * vr[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
    }
    if (lane == 0) {
      for (int j = 0; j < nj; ++j) {
        double r = acc[j];
        if (d) r += d[b * N + row] * vb[(size_t)row * c + j0 + j];
        y[(b * N + row) * (size_t)c + j0 + j] = r;
      }
    }
  }
}

// rhs_norm, its zero mask, u = rhs / rhs_norm, x = x0 / rhs_norm (:177-183)
__global__ __launch_bounds__(kThreads) void k64_normalise(const double* __restrict__ rhs, const double* __restrict__ x0,
                                                           const double* __restrict__ nsq, double eps,
                                                           double* __restrict__ u, double* __restrict__ x,
                                                           double* __restrict__ rn, int* __restrict__ rz, int N, int c,
                                                           size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % c);
  const size_t bi = e / c;
  const size_t b = bi / N;
  double nrm = sqrt(nsq[b * c + j]);
  const bool zero = nrm < eps;
  if (zero) nrm = 1.0;
  u[e] = rhs[e] / nrm;
  x[e] = x0 ? x0[e] / nrm : 0.0;
  if (bi % N == 0) {
    rn[b * c + j] = nrm;
    rz[b * c + j] = zero ? 1 : 0;
  }
}

// r = u - A x0 (:186), NaN test (:199)
__global__ __launch_bounds__(kThreads) void k64_residual(const double* __restrict__ u, const double* __restrict__ ax,
                                                          double* __restrict__ r, Ctl64* ctl, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const double v = u[e] - ax[e];
  r[e] = v;
  if (v != v) ctl->nan_detected = 1;
}

// residual norms, has_converged, skip test (:204-208); one workgroup
__global__ __launch_bounds__(kThreads) void k64_ctrl_init(const double* __restrict__ rr, double stop_after, int n_tridiag,
                                                           int* __restrict__ conv, Ctl64* ctl, int BC) {
  __shared__ int all_s;
  __shared__ double sum_s[kThreads];
  if (threadIdx.x == 0) all_s = 1;
  __syncthreads();
  double sum = 0.0;
  int all = 1;
  for (int e = threadIdx.x; e < BC; e += kThreads) {
    const double nrm = sqrt(rr[e]);
    const int cv = nrm < stop_after ? 1 : 0;
    conv[e] = cv;
    all &= cv;
    sum += nrm;
  }
  if (!all) all_s = 0;
  sum_s[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kThreads; ++i) t += sum_s[i];
    ctl->mean_resid = t / BC;
    if (all_s && !n_tridiag) {
      ctl->skipped = 1;
      ctl->stop = 1;
    }
  }
}

__device__ __forceinline__ double masked_ratio(double num, double den, double eps) {
  const bool zero = den < eps;  // (negative values are zeroed too, :254 / :39)
  const double q = num / (zero ? 1.0 : den);
  return zero ? 0.0 : q;
}

// alpha (:250-260) and r -= alpha * Ap (:264)
__global__ __launch_bounds__(kThreads) void k64_update_r(const double* __restrict__ pAp, const double* __restrict__ rzp,
                                                          const int* __restrict__ conv, double eps,
                                                          const double* __restrict__ Ap, double* __restrict__ r,
                                                          double* __restrict__ alpha, int N, int c, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % c);
  const size_t bi = e / c;
  const size_t s = (bi / N) * c + j;
  double a = masked_ratio(rzp[s], pAp[s], eps);
  if (conv[s]) a = 0.0;
  r[e] = r[e] - a * Ap[e];
  if (bi % N == 0) alpha[s] = a;
}

// x += alpha p (:31), beta (:34-43), p = p * beta + z (:46)
__global__ __launch_bounds__(kThreads) void k64_update_xp(const double* __restrict__ alpha,
                                                           const double* __restrict__ rz_old,
                                                           const double* __restrict__ rz_new, double eps,
                                                           const double* __restrict__ z, double* __restrict__ x,
                                                           double* __restrict__ p, double* __restrict__ beta, int N,
                                                           int c, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % c);
  const size_t bi = e / c;
  const size_t s = (bi / N) * c + j;
  const double bt = masked_ratio(rz_new[s], rz_old[s], eps);
  const double pv = p[e];
  x[e] = x[e] + alpha[s] * pv;
  p[e] = pv * bt + z[e];
  if (bi % N == 0) beta[s] = bt;
}

struct Tri64 {
  double* t_mat;  // [n_tridiag, B, T, T]
  double* prev_arec;
  double* prev_beta;  // [B, n_tridiag]
  int n_tridiag, T, B;
};

// residual norms, has_converged, stopping rule (:298-308), tridiagonal update (:311-332); one workgroup
__global__ __launch_bounds__(kThreads) void k64_ctrl(const double* __restrict__ rr, const int* __restrict__ rz,
                                                      double stop_after, double tol, int k, int floor_max_iter,
                                                      int n_tridiag_iter, const double* __restrict__ alpha,
                                                      const double* __restrict__ beta, Tri64 tri,
                                                      int* __restrict__ conv, Ctl64* ctl, int B, int c) {
  __shared__ double sum_s[kThreads];
  __shared__ double max_s[kThreads];
  __shared__ int nan_s[kThreads];
  const int BC = B * c;
  double sum = 0.0;
  for (int e = threadIdx.x; e < BC; e += kThreads) {
    double nrm = sqrt(rr[e]);
    if (rz[e]) nrm = 0.0;
    conv[e] = nrm < stop_after ? 1 : 0;
    sum += nrm;
  }
  sum_s[threadIdx.x] = sum;
  __syncthreads();
  double mean = 0.0;
  for (int i = 0; i < kThreads; ++i) mean += sum_s[i];
  mean /= BC;
  const bool tri_floor = tri.n_tridiag && k < min(n_tridiag_iter, floor_max_iter - 1);
  const bool stop = k >= min(10, floor_max_iter - 1) && mean < tol && !tri_floor;
  if (threadIdx.x == 0) {
    ctl->iterations = k + 1;
    ctl->mean_resid = mean;
    if (stop) {
      ctl->tol_reached = 1;
      ctl->stop = 1;
    }
  }
  if (stop) return;
  if (!(tri.n_tridiag && k < n_tridiag_iter && !ctl->tri_disabled)) return;
  double omax = -INFINITY;
  int onan = 0;
  const int T = tri.T;
  for (int e = threadIdx.x; e < B * tri.n_tridiag; e += kThreads) {
    const int b = e / tri.n_tridiag, j = e % tri.n_tridiag;
    const double at = alpha[(size_t)b * c + j], bt = beta[(size_t)b * c + j];
    const double arec = 1.0 / (at == 0.0 ? 1.0 : at);
    double* Tm = tri.t_mat + ((size_t)j * B + b) * T * T;
    if (k == 0) {
      Tm[0] = arec;
    } else {
      const double pb = tri.prev_beta[e], pa = tri.prev_arec[e];
      Tm[(size_t)k * T + k] = arec + pb * pa;
      const double off = sqrt(pb) * pa;
      Tm[(size_t)k * T + k - 1] = off;
      Tm[(size_t)(k - 1) * T + k] = off;
      if (off != off) onan = 1;
      omax = fmax(omax, off);
    }
    tri.prev_arec[e] = arec;
    tri.prev_beta[e] = bt;
  }
  max_s[threadIdx.x] = omax;
  nan_s[threadIdx.x] = onan;
  __syncthreads();
  if (threadIdx.x == 0) {
    ctl->last_tridiag_iter = k;
    if (k > 0) {
      double m = -INFINITY;
      int an = 0;
      for (int i = 0; i < kThreads; ++i) {
        m = fmax(m, max_s[i]);
        an |= nan_s[i];
      }
      if (!an && m < 1e-6) ctl->tri_disabled = 1;  // :326 (a NaN maximum compares false)
    }
  }
}

__global__ __launch_bounds__(kThreads) void k64_copy(const double* __restrict__ a, double* __restrict__ o, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e < total) o[e] = a[e];
}

// result * rhs_norm (:335)
__global__ __launch_bounds__(kThreads) void k64_final(const double* __restrict__ x, const double* __restrict__ rn,
                                                       double* __restrict__ out, int N, int c, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % c);
  const size_t b = e / c / N;
  out[e] = x[e] * rn[b * c + j];
}

// shared with lo_minres_f64.hip
int f64_dots(const double* a, const double* b1, double* out1, const double* a2, const double* b2, double* out2,
             int64_t B, int64_t N, int64_t c, hipStream_t st) {
  hipLaunchKernelGGL(k64_dots, dim3((unsigned)c, (unsigned)B), dim3(kThreads), 0, st, a, b1, out1, a2, b2, out2, (int)N,
                     (int)c);
  LO_LAUNCH_CHECK();
  return LO_OK;
}
int f64_dense_mv(const double* A, const double* d, const double* v, double* y, int64_t B, int64_t N, int64_t c,
                 hipStream_t st) {
  hipLaunchKernelGGL(k64_dense_mv, dim3((unsigned)((N + 3) / 4), (unsigned)B), dim3(kThreads), 0, st, A, d, v, y, (int)N,
                     (int)c);
  LO_LAUNCH_CHECK();
  return LO_OK;
}
int f64_copy(const double* a, double* o, size_t total, hipStream_t st) {
  hipLaunchKernelGGL(k64_copy, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, a, o, total);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

struct Lay64 {
  double *u, *r, *z, *p, *Ap, *x;
  double *nsq, *rn, *rr, *pAp, *rz_a, *rz_b, *alpha, *beta, *prev_arec, *prev_beta;
  int *rzero, *conv;
  Ctl64* ctl;
};

static void lay64(int64_t B, int64_t N, const lo_cg_params_f64* prm, Arena& ar, Lay64* l) {
  const size_t V = (size_t)B * N * prm->c, S = (size_t)B * prm->c;
  l->ctl = ar.take<Ctl64>(1);
  l->u = ar.take<double>(V);
  l->r = ar.take<double>(V);
  l->z = ar.take<double>(V);
  l->p = ar.take<double>(V);
  l->Ap = ar.take<double>(V);
  l->x = ar.take<double>(V);
  l->nsq = ar.take<double>(S);
  l->rn = ar.take<double>(S);
  l->rr = ar.take<double>(S);
  l->pAp = ar.take<double>(S);
  l->rz_a = ar.take<double>(S);
  l->rz_b = ar.take<double>(S);
  l->alpha = ar.take<double>(S);
  l->beta = ar.take<double>(S);
  l->prev_arec = ar.take<double>(S);
  l->prev_beta = ar.take<double>(S);
  l->rzero = ar.take<int>(S);
  l->conv = ar.take<int>(S);
}

}  // namespace lo

using namespace lo;

extern "C" size_t lo_cg_f64_workspace_bytes(int64_t B, int64_t N, const lo_cg_params_f64* prm) {
  Arena ar(nullptr, 0);
  Lay64 l;
  lay64(B, N, prm, ar, &l);
  return ar.off + 1024;
}

extern "C" int lo_cg_solve_f64(const double* A, const double* diag, lo_matvec_cb_f64 matvec, void* matvec_user,
                               lo_matvec_cb_f64 precond_cb, void* precond_user, const lo_cg_params_f64* prm,
                               int64_t B, int64_t N, const double* rhs, const double* x0, double* x, double* t_mat,
                               void* ws, size_t ws_bytes, lo_cg_info_f64* info, void* stream) {
  if (!prm || !rhs || !x || !info || (!A && !matvec)) return LO_ERR_BADARG;
  const int64_t c = prm->c;
  if (B < 1 || N < 1 || c < 1 || B > 65535 || c > 0x7fffffff / 2 || N > 0x7ffffff0) return LO_ERR_UNSUPPORTED;
  if (prm->n_tridiag < 0 || prm->n_tridiag > c || (prm->n_tridiag && !t_mat)) return LO_ERR_BADARG;
  if ((size_t)B * (size_t)c > (size_t)1 << 30) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Arena ar(ws, ws_bytes);
  Lay64 l;
  lay64(B, N, prm, ar, &l);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const size_t total = (size_t)B * N * c;
  const unsigned eg = (unsigned)((total + kThreads - 1) / kThreads);
  const dim3 dgrid((unsigned)c, (unsigned)B);
  const int BC = (int)(B * c);
  const int floor_max = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter;
  const int T = prm->max_tridiag_iter;
  memset(info, 0, sizeof(*info));

  auto apply_op = [&](const double* v, double* y) -> int {
    if (A) {
      hipLaunchKernelGGL(k64_dense_mv, dim3((unsigned)((N + 3) / 4), (unsigned)B), dim3(kThreads), 0, st, A, diag, v, y,
                         (int)N, (int)c);
      LO_LAUNCH_CHECK();
      return LO_OK;
    }
    return matvec(matvec_user, v, y, B, N, c, stream) ? LO_ERR_LAUNCH : LO_OK;
  };
  auto apply_pre = [&](const double* v, double* y) -> int {
    if (precond_cb) return precond_cb(precond_user, v, y, B, N, c, stream) ? LO_ERR_LAUNCH : LO_OK;
    hipLaunchKernelGGL(k64_copy, dim3(eg), dim3(kThreads), 0, st, v, y, total);
    LO_LAUNCH_CHECK();
    return LO_OK;
  };
  Ctl64 h;
  auto read_ctl = [&]() -> int {
    LO_HIP_CHECK(hipMemcpyAsync(&h, l.ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    LO_HIP_CHECK(hipStreamSynchronize(st));
    return LO_OK;
  };

  LO_HIP_CHECK(hipMemsetAsync(l.ctl, 0, sizeof(Ctl64), st));
  if (prm->n_tridiag)
    LO_HIP_CHECK(hipMemsetAsync(t_mat, 0, sizeof(double) * (size_t)prm->n_tridiag * B * T * T, st));
  hipLaunchKernelGGL(k64_dots, dgrid, dim3(kThreads), 0, st, rhs, rhs, l.nsq, nullptr, nullptr, nullptr, (int)N, (int)c);
  hipLaunchKernelGGL(k64_normalise, dim3(eg), dim3(kThreads), 0, st, rhs, x0, l.nsq, prm->eps, l.u, l.x, l.rn, l.rzero,
                     (int)N, (int)c, total);
  LO_LAUNCH_CHECK();
  int rc = apply_op(l.x, l.Ap);
  if (rc) return rc;
  int matvecs = 1;
  hipLaunchKernelGGL(k64_residual, dim3(eg), dim3(kThreads), 0, st, l.u, l.Ap, l.r, l.ctl, total);
  hipLaunchKernelGGL(k64_dots, dgrid, dim3(kThreads), 0, st, l.r, l.r, l.rr, nullptr, nullptr, nullptr, (int)N, (int)c);
  hipLaunchKernelGGL(k64_ctrl_init, dim3(1), dim3(kThreads), 0, st, l.rr, prm->stop_updating_after, prm->n_tridiag,
                     l.conv, l.ctl, BC);
  LO_LAUNCH_CHECK();
  if ((rc = read_ctl())) return rc;
  if (h.nan_detected) {
    info->nan_detected = 1;
    info->matvecs = matvecs;
    return LO_OK;
  }
  double* rz_old = l.rz_a;
  double* rz_new = l.rz_b;
  if (!h.stop && prm->max_iter > 0) {
    if ((rc = apply_pre(l.r, l.z))) return rc;
    hipLaunchKernelGGL(k64_copy, dim3(eg), dim3(kThreads), 0, st, l.z, l.p, total);
    hipLaunchKernelGGL(k64_dots, dgrid, dim3(kThreads), 0, st, l.z, l.r, rz_old, nullptr, nullptr, nullptr, (int)N,
                       (int)c);
    LO_LAUNCH_CHECK();
    Tri64 tri{t_mat, l.prev_arec, l.prev_beta, prm->n_tridiag, T, (int)B};
    for (int k = 0; k < prm->max_iter; ++k) {
      if ((rc = apply_op(l.p, l.Ap))) return rc;
      ++matvecs;
      hipLaunchKernelGGL(k64_dots, dgrid, dim3(kThreads), 0, st, l.p, l.Ap, l.pAp, nullptr, nullptr, nullptr, (int)N,
                         (int)c);
      hipLaunchKernelGGL(k64_update_r, dim3(eg), dim3(kThreads), 0, st, l.pAp, rz_old, l.conv, prm->eps, l.Ap, l.r,
                         l.alpha, (int)N, (int)c, total);
      LO_LAUNCH_CHECK();
      if ((rc = apply_pre(l.r, l.z))) return rc;
      hipLaunchKernelGGL(k64_dots, dgrid, dim3(kThreads), 0, st, l.r, l.z, rz_new, l.r, l.r, l.rr, (int)N, (int)c);
      hipLaunchKernelGGL(k64_update_xp, dim3(eg), dim3(kThreads), 0, st, l.alpha, rz_old, rz_new, prm->eps, l.z, l.x,
                         l.p, l.beta, (int)N, (int)c, total);
      hipLaunchKernelGGL(k64_ctrl, dim3(1), dim3(kThreads), 0, st, l.rr, l.rzero, prm->stop_updating_after,
                         prm->tolerance, k, floor_max, std::min<int>(prm->max_tridiag_iter, T), l.alpha, l.beta, tri,
                         l.conv, l.ctl, (int)B, (int)c);
      LO_LAUNCH_CHECK();
      std::swap(rz_old, rz_new);
      if ((rc = read_ctl())) return rc;
      if (h.stop) break;
    }
  }
  hipLaunchKernelGGL(k64_final, dim3(eg), dim3(kThreads), 0, st, l.x, l.rn, x, (int)N, (int)c, total);
  LO_LAUNCH_CHECK();
  if ((rc = read_ctl())) return rc;
  info->iterations = h.iterations;
  info->matvecs = matvecs;
  info->tolerance_reached = h.tol_reached;
  info->nan_detected = h.nan_detected;
  info->skipped = h.skipped;
  info->last_tridiag_iter = h.last_tridiag_iter;
  info->mean_residual = h.mean_resid;
  return LO_OK;
}
