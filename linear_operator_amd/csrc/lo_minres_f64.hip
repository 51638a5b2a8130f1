// lo_minres_f64.hip -- the reference's shifted MINRES (linear_operator/utils/minres.py:10-207, update block :210-282)
// in fp64.
//
// Like lo_cg_f64.hip this engine exists so that the reference's own fp64 recipes (test/utils/test_minres.py builds
// float64 operands throughout) run unmodified; the performance path is lo_minres.hip (fp32).  Plain streaming
// formulation: one preconditioned Lanczos recurrence (z, q, alpha, beta) shared by the Q shifts, the Givens recurrences
// of every shift as scalars [Q, B, c], two search vectors and the solution per shift [Q, B, N, c]; a dense fp64 operator
// (k64_dense_mv) or a closure called back once per iteration; the preconditioner is a closure or absent.
#include "lo_internal.h"

#include <algorithm>
#include <math.h>
#include <string.h>

namespace lo {

struct Mr64Ctl {
  int stop;
  int iterations;
  int converged;
  int pad;
  double conv;
};

// u = rhs / ||rhs|| with the zero-column mask (:52-56)
__global__ __launch_bounds__(kThreads) void k64m_normalise(const double* __restrict__ rhs, const double* __restrict__ nsq,
                                                            double* __restrict__ u, double* __restrict__ rn,
                                                            int* __restrict__ rz, int N, int c, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % c);
  const size_t bi = e / c;
  const size_t s = (bi / N) * c + j;
  double nrm = sqrt(nsq[s]);
  const bool zero = nrm < 1e-10;
  if (zero) nrm = 1.0;
  u[e] = rhs[e] / nrm;
  if (bi % N == 0) {
    rn[s] = nrm;
    rz[s] = zero ? 1 : 0;
  }
}

// v[e] /= sqrt-like scalar per (b, col): a /= sc, b /= sc (the normalisations :81-83 and :151-152)
__global__ __launch_bounds__(kThreads) void k64m_div2(double* __restrict__ a, double* __restrict__ b2,
                                                       const double* __restrict__ sc, int N, int c, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const size_t s = (e / c / N) * c + e % c;
  const double d = sc[s];
  a[e] = a[e] / d;
  b2[e] = b2[e] / d;
}

// beta = max(sqrt(dot), eps) (:80 without the clamp when eps < 0, :147-150 with it)
__global__ __launch_bounds__(kThreads) void k64m_sqrt(const double* __restrict__ dot, double eps,
                                                       double* __restrict__ out, int n) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const double v = sqrt(dot[i]);
  out[i] = (eps >= 0.0 && v == v) ? fmax(v, eps) : v;  // (clamp_min keeps a NaN; eps < 0: no clamp, :80)
}

// alpha = value * <prod, q1> (:141-142);  zc = value * prod - alpha z1 - beta_prev z2 (:144)
__global__ __launch_bounds__(kThreads) void k64m_zc(const double* __restrict__ prod, const double* __restrict__ z1,
                                                     const double* __restrict__ z2, const double* __restrict__ dot,
                                                     const double* __restrict__ beta_prev, double value,
                                                     double* __restrict__ zc, double* __restrict__ alpha, int N, int c,
                                                     size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const size_t bi = e / c;
  const size_t s = (bi / N) * c + e % c;
  const double al = value * dot[s];
  zc[e] = value * prod[e] - al * z1[e] - beta_prev[s] * z2[e];
  if (bi % N == 0) alpha[s] = al;
}

struct Mr64Rot {
  double *cos1, *sin1, *cos2, *sin2, *scale_prev;  // [Q, B, c] state
  double *sub, *subsub, *diag, *scale_now;         // [Q, B, c] coefficients of this iteration's vector update
};

// QR (Givens) recurrence of every shift (:236-262 of the update block)
__global__ __launch_bounds__(kThreads) void k64m_givens(Mr64Rot r, const double* __restrict__ alpha,
                                                         const double* __restrict__ beta,
                                                         const double* __restrict__ beta_prev,
                                                         const double* __restrict__ shifts, int per_member, int Q, int B,
                                                         int c) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= Q * B * c) return;
  const int q = i / (B * c), bc = i % (B * c), b = bc / c;
  const double sh = per_member ? shifts[(size_t)q * B + b] : shifts[q];
  const double bp = beta_prev[bc], be = beta[bc];
  const double cos1 = r.cos1[i], sin1 = r.sin1[i], cos2 = r.cos2[i], sin2 = r.sin2[i];
  const double subsub = sin2 * bp;
  double sub = cos2 * bp;
  const double alpha_s = alpha[bc] + sh;
  double diag = alpha_s * cos1 - sin1 * sub;
  sub = sub * cos1 + sin1 * alpha_s;
  const double radius = sqrt(diag * diag + be * be);
  const double cosc = diag / radius, sinc = be / radius;
  diag = diag * cosc + sinc * be;
  const double sp = r.scale_prev[i];
  const double scale_curr = -(sp * sinc);
  r.scale_now[i] = sp * cosc;
  r.scale_prev[i] = scale_curr;  // (next iteration's scale_prev, :198)
  r.sub[i] = sub;
  r.subsub[i] = subsub;
  r.diag[i] = diag;
  r.cos2[i] = cos1; r.cos1[i] = cosc;  // :190-193
  r.sin2[i] = sin1; r.sin1[i] = sinc;
}

// search vector and solution of every shift (:264-282): sc = (q1 - sub s1 - subsub s2) / diag (written over s2, which the
// caller then renames s1), solution += sc * scale
__global__ __launch_bounds__(kThreads) void k64m_update(const double* __restrict__ q1, const double* __restrict__ s1,
                                                         double* __restrict__ s2, double* __restrict__ sol, Mr64Rot r,
                                                         int N, int c, size_t per_shift, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const size_t q = e / per_shift, rem = e % per_shift;  // rem = (b, i, col)
  const size_t B_c = per_shift / N;                     // B * c
  const size_t s = q * B_c + (rem / c / N) * c + rem % c;
  const double sc = (q1[rem] - r.sub[s] * s1[e] - r.subsub[s] * s2[e]) / r.diag[s];
  s2[e] = sc;
  sol[e] = sol[e] + sc * r.scale_now[s];
}

// conv = mean over (shift, member, column) of ||search update|| / ||solution|| (:178-183); one workgroup
__global__ __launch_bounds__(kThreads) void k64m_conv(const double* __restrict__ ss_sc, const double* __restrict__ ss_sol,
                                                       const double* __restrict__ scale_now, double tol, int it, int n,
                                                       Mr64Ctl* ctl) {
  __shared__ double sum_s[kThreads];
  double sum = 0.0;
  for (int i = threadIdx.x; i < n; i += kThreads) sum += fabs(scale_now[i]) * sqrt(ss_sc[i]) / sqrt(ss_sol[i]);
  sum_s[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kThreads; ++i) t += sum_s[i];
    const double conv = t / n;
    ctl->conv = conv;
    ctl->iterations = it + 1;
    if (conv < tol) {  // (a NaN mean -- 0 / 0 of an all-zero column -- compares false: the loop runs on, as the reference's)
      ctl->converged = 1;
      ctl->stop = 1;
    }
  }
}

// solution masked for the zero columns and scaled back (:201, :212)
__global__ __launch_bounds__(kThreads) void k64m_final(const double* __restrict__ sol, const double* __restrict__ rn,
                                                        const int* __restrict__ rz, double* __restrict__ x, int N, int c,
                                                        size_t per_shift, size_t total) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const size_t rem = e % per_shift;
  const size_t s = (rem / c / N) * c + rem % c;
  x[e] = rz[s] ? 0.0 : sol[e] * rn[s];
}

__global__ __launch_bounds__(kThreads) void k64m_fill(double* __restrict__ a, double v, size_t n) {
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e < n) a[e] = v;
}

// scale_prev[q, b, col] = beta_prev[b, col] (:113)
__global__ __launch_bounds__(kThreads) void k64m_bcast(const double* __restrict__ src, double* __restrict__ dst, int bc,
                                                        int n) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) dst[i] = src[i % bc];
}

struct LayM64 {
  Mr64Ctl* ctl;
  double *z1, *z2, *zc, *q1, *qc, *prod;  // [B, N, c]
  double *s1, *s2, *sol;                  // [Q, B, N, c]
  double *nsq, *rn, *dot, *alpha, *beta, *beta_prev;  // [B, c]
  int* rz;
  Mr64Rot rot;
  double *ss_sc, *ss_sol;  // [Q, B, c]
};

static void lay_m64(int64_t B, int64_t N, const lo_minres_params_f64* prm, Arena& ar, LayM64* l) {
  const size_t V = (size_t)B * N * prm->c, S = (size_t)B * prm->c, Q = (size_t)prm->n_shifts;
  l->ctl = ar.take<Mr64Ctl>(1);
  l->z1 = ar.take<double>(V); l->z2 = ar.take<double>(V); l->zc = ar.take<double>(V);
  l->q1 = ar.take<double>(V); l->qc = ar.take<double>(V); l->prod = ar.take<double>(V);
  l->s1 = ar.take<double>(Q * V); l->s2 = ar.take<double>(Q * V); l->sol = ar.take<double>(Q * V);
  l->nsq = ar.take<double>(S); l->rn = ar.take<double>(S); l->dot = ar.take<double>(S);
  l->alpha = ar.take<double>(S); l->beta = ar.take<double>(S); l->beta_prev = ar.take<double>(S);
  l->rz = ar.take<int>(S);
  double** rp[] = {&l->rot.cos1, &l->rot.sin1, &l->rot.cos2, &l->rot.sin2, &l->rot.scale_prev,
                   &l->rot.sub,  &l->rot.subsub, &l->rot.diag, &l->rot.scale_now, &l->ss_sc, &l->ss_sol};
  for (double** p : rp) *p = ar.take<double>(Q * S);
}

}  // namespace lo

using namespace lo;

extern "C" size_t lo_minres_f64_workspace_bytes(int64_t B, int64_t N, const lo_minres_params_f64* prm) {
  Arena ar(nullptr, 0);
  LayM64 l;
  lay_m64(B, N, prm, ar, &l);
  return ar.off + 1024;
}

extern "C" int lo_minres_f64(const double* A, const double* diag, lo_matvec_cb_f64 matvec, void* matvec_user,
                             lo_matvec_cb_f64 precond_cb, void* precond_user, const lo_minres_params_f64* prm,
                             int64_t B, int64_t N, const double* rhs, const double* shifts, double* x, void* ws,
                             size_t ws_bytes, lo_minres_info_f64* info, void* stream) {
  if (!prm || !rhs || !x || !info || !shifts || (!A && !matvec)) return LO_ERR_BADARG;
  const int64_t c = prm->c;
  const int Q = prm->n_shifts;
  if (B < 1 || N < 1 || c < 1 || Q < 1 || N > 0x7ffffff0 || (int64_t)Q * B > 65535 || (int64_t)Q * B * c > (1 << 30))
    return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  Arena ar(ws, ws_bytes);
  LayM64 l;
  lay_m64(B, N, prm, ar, &l);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const size_t V = (size_t)B * N * c, QV = (size_t)Q * V;
  const int S = (int)(B * c), QS = Q * S;
  const double value = prm->has_value ? prm->value : 1.0;
  auto eg = [](size_t n) { return dim3((unsigned)((n + kThreads - 1) / kThreads)); };
  const dim3 blk(kThreads);
  memset(info, 0, sizeof(*info));

  auto apply_op = [&](const double* v, double* y) -> int {
    if (A) return f64_dense_mv(A, diag, v, y, B, N, c, st);
    return matvec(matvec_user, v, y, B, N, c, stream) ? LO_ERR_LAUNCH : LO_OK;
  };
  auto apply_pre = [&](const double* v, double* y) -> int {
    if (precond_cb) return precond_cb(precond_user, v, y, B, N, c, stream) ? LO_ERR_LAUNCH : LO_OK;
    return f64_copy(v, y, V, st);
  };
  int rc;
  LO_HIP_CHECK(hipMemsetAsync(l.ctl, 0, sizeof(Mr64Ctl), st));
  // rhs norms and the zero mask (:52-56); z1 = rhs / norm
  if ((rc = f64_dots(rhs, rhs, l.nsq, nullptr, nullptr, nullptr, B, N, c, st))) return rc;
  hipLaunchKernelGGL(k64m_normalise, eg(V), blk, 0, st, rhs, l.nsq, l.z1, l.rn, l.rz, (int)N, (int)c, V);
  LO_LAUNCH_CHECK();
  int matvecs = 0;
  // the reference spends one product on the shapes (:66-68): a closure sees the same number of calls here
  if (!A) {
    if ((rc = apply_op(l.z1, l.prod))) return rc;
  }
  ++matvecs;
  LO_HIP_CHECK(hipMemsetAsync(l.z2, 0, sizeof(double) * V, st));
  LO_HIP_CHECK(hipMemsetAsync(l.s1, 0, sizeof(double) * QV, st));
  LO_HIP_CHECK(hipMemsetAsync(l.s2, 0, sizeof(double) * QV, st));
  LO_HIP_CHECK(hipMemsetAsync(l.sol, 0, sizeof(double) * QV, st));
  if ((rc = apply_pre(l.z1, l.q1))) return rc;                                              // :76-79
  if ((rc = f64_dots(l.z1, l.q1, l.dot, nullptr, nullptr, nullptr, B, N, c, st))) return rc;  // :80
  hipLaunchKernelGGL(k64m_sqrt, eg(S), blk, 0, st, l.dot, -1.0, l.beta_prev, S);
  hipLaunchKernelGGL(k64m_div2, eg(V), blk, 0, st, l.z1, l.q1, l.beta_prev, (int)N, (int)c, V);  // :81-83
  hipLaunchKernelGGL(k64m_fill, eg(QS), blk, 0, st, l.rot.cos1, 1.0, (size_t)QS);               // :96-111
  hipLaunchKernelGGL(k64m_fill, eg(QS), blk, 0, st, l.rot.cos2, 1.0, (size_t)QS);
  LO_HIP_CHECK(hipMemsetAsync(l.rot.sin1, 0, sizeof(double) * QS, st));
  LO_HIP_CHECK(hipMemsetAsync(l.rot.sin2, 0, sizeof(double) * QS, st));
  hipLaunchKernelGGL(k64m_bcast, eg(QS), blk, 0, st, l.beta_prev, l.rot.scale_prev, S, QS);      // :113
  LO_LAUNCH_CHECK();

  double *z1 = l.z1, *z2 = l.z2, *zc = l.zc, *q1 = l.q1, *qc = l.qc, *s1 = l.s1, *s2 = l.s2;
  double *beta = l.beta, *beta_prev = l.beta_prev;
  Mr64Ctl h;
  memset(&h, 0, sizeof(h));
  const int n_loop = prm->max_iter + 2;  // :134
  int it = 0;
  for (; it < n_loop; ++it) {
    if ((rc = apply_op(q1, l.prod))) return rc;  // :137-139
    ++matvecs;
    if ((rc = f64_dots(l.prod, q1, l.dot, nullptr, nullptr, nullptr, B, N, c, st))) return rc;
    hipLaunchKernelGGL(k64m_zc, eg(V), blk, 0, st, l.prod, z1, z2, l.dot, beta_prev, value, zc, l.alpha, (int)N, (int)c, V);
    LO_LAUNCH_CHECK();
    if ((rc = apply_pre(zc, qc))) return rc;     // :145-146
    if ((rc = f64_dots(zc, qc, l.dot, nullptr, nullptr, nullptr, B, N, c, st))) return rc;
    hipLaunchKernelGGL(k64m_sqrt, eg(S), blk, 0, st, l.dot, prm->eps, beta, S);              // :147-150
    hipLaunchKernelGGL(k64m_div2, eg(V), blk, 0, st, zc, qc, beta, (int)N, (int)c, V);        // :151-152
    hipLaunchKernelGGL(k64m_givens, eg(QS), blk, 0, st, l.rot, l.alpha, beta, beta_prev, shifts,
                       prm->shifts_per_member, Q, (int)B, (int)c);
    hipLaunchKernelGGL(k64m_update, eg(QV), blk, 0, st, q1, s1, s2, l.sol, l.rot, (int)N, (int)c, V, QV);
    LO_LAUNCH_CHECK();
    std::swap(s1, s2);  // the new search vector (written over s2) is s1 from now on, the old s1 becomes s2
    if ((it + 1) % 10 == 0) {  // :178-183
      if ((rc = f64_dots(s1, s1, l.ss_sc, l.sol, l.sol, l.ss_sol, (int64_t)Q * B, N, c, st))) return rc;
      hipLaunchKernelGGL(k64m_conv, dim3(1), blk, 0, st, l.ss_sc, l.ss_sol, l.rot.scale_now, prm->tolerance, it, QS,
                         l.ctl);
      LO_LAUNCH_CHECK();
      LO_HIP_CHECK(hipMemcpyAsync(&h, l.ctl, sizeof(h), hipMemcpyDeviceToHost, st));
      LO_HIP_CHECK(hipStreamSynchronize(st));
      if (h.stop) {
        ++it;
        break;
      }
    }
    // :186-198 (cos / sin / scale were rotated by k64m_givens)
    double* zt = z2;
    z2 = z1; z1 = zc; zc = zt;
    std::swap(q1, qc);
    std::swap(beta, beta_prev);
  }
  hipLaunchKernelGGL(k64m_final, eg(QV), blk, 0, st, l.sol, l.rn, l.rz, x, (int)N, (int)c, V, QV);
  LO_LAUNCH_CHECK();
  LO_HIP_CHECK(hipStreamSynchronize(st));
  info->iterations = it;
  info->matvecs = matvecs;
  info->converged = h.converged;
  info->conv = h.conv;
  return LO_OK;
}
