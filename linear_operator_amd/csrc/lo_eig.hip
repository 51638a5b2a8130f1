// lo_eig.hip -- batched eigendecomposition of the small symmetric tridiagonal matrices CG / Lanczos
// produce, fused with the stochastic-Lanczos-quadrature reduction.  Restates
//   lanczos_tridiag_to_diag   (linear_operator/utils/lanczos.py:167-189): eigh, negative eigenvalues -> 1 and
//                              their eigenvector columns -> 0 (:185-187)
//   StochasticLQ.to_dense     (linear_operator/utils/stochastic_lq.py:67-82) with funcs = [log]:
//                              logdet[b] = (n / P) sum_p sum_i evec[p,b,0,i]^2 log(eval[p,b,i])
// and removes the reference's device -> host -> device round trip (lanczos.py:179-189: eigh runs on the
// CPU when k < 32).  One thread per tridiagonal (k <= 32): implicit-shift QL in fp64 registers/scratch
// (more accurate than the reference's fp32 LAPACK call -- see DESIGN.md "logdet noise floor"),
// eigenvalues sorted ascending like torch.linalg.eigh.
#include <algorithm>
#include <cmath>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int kEigMaxT = 32;

template <bool FULL>
__global__ __launch_bounds__(64) void k_tridiag_eigh(const float* __restrict__ t_mat, int64_t M, int T,
                                                      float* __restrict__ evals, float* __restrict__ evecs,
                                                      double* __restrict__ slq_term, int* __restrict__ fail) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M) return;
  const float* t = t_mat + (size_t)idx * T * T;
  double d[kEigMaxT], e[kEigMaxT];
  double z0[kEigMaxT];                       // first row of the eigenvector matrix
  double Z[FULL ? kEigMaxT * kEigMaxT : 1];  // full eigenvector matrix (row-major) when requested
  const int n = T;
  for (int i = 0; i < n; ++i) {
    d[i] = (double)t[i * T + i];
    e[i] = (i + 1 < n) ? (double)t[i * T + i + 1] : 0.0;
    z0[i] = (i == 0) ? 1.0 : 0.0;
  }
  if (FULL)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) Z[i * kEigMaxT + j] = (i == j) ? 1.0 : 0.0;

  bool ok = true;
  for (int l = 0; l < n && ok; ++l) {
    int iter = 0;
    int m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 80) { ok = false; break; }
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + copysign(r, g));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double bq = c * e[i];
          r = hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * bq;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - bq;
          f = z0[i + 1];
          z0[i + 1] = s * z0[i] + c * f;
          z0[i] = c * z0[i] - s * f;
          if (FULL) {
            for (int k = 0; k < n; ++k) {
              const double fk = Z[k * kEigMaxT + i + 1];
              Z[k * kEigMaxT + i + 1] = s * Z[k * kEigMaxT + i] + c * fk;
              Z[k * kEigMaxT + i] = c * Z[k * kEigMaxT + i] - s * fk;
            }
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  if (!ok) atomicExch(fail, 1);

  // ascending order (selection sort on indices; n <= 32)
  int ord[kEigMaxT];
  for (int i = 0; i < n; ++i) ord[i] = i;
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    for (int j = i + 1; j < n; ++j)
      if (d[ord[j]] < d[ord[k]]) k = j;
    const int tmp = ord[i];
    ord[i] = ord[k];
    ord[k] = tmp;
  }
  double term = 0.0;
  for (int i = 0; i < n; ++i) {
    const int o = ord[i];
    const bool pos = d[o] >= 0.0;                       // mask = evals.ge(0)          lanczos.py:185
    const double ev = pos ? d[o] : 1.0;                 // masked_fill_(~mask, 1)      :187
    const double w0 = pos ? z0[o] : 0.0;                // evecs * mask (columns)      :186
    if (evals) evals[(size_t)idx * T + i] = (float)ev;
    term += w0 * w0 * log(ev);                          // stochastic_lq.py:77-79
    if (FULL)
      for (int k = 0; k < n; ++k) evecs[((size_t)idx * T + k) * T + i] = pos ? (float)Z[k * kEigMaxT + o] : 0.f;
  }
  if (slq_term) slq_term[idx] = term;
}

// logdet[b] = (n / P) * sum_p term[p*B + b]   (fixed order over probes)
__global__ void k_slq_reduce(const double* __restrict__ term, int64_t P, int64_t B, double scale,
                             float* __restrict__ logdet) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double acc = 0.0;
  for (int64_t p = 0; p < P; ++p) acc += scale * term[p * B + b];
  logdet[b] = (float)acc;
}

}  // namespace lo

using namespace lo;

extern "C" {

size_t lo_tridiag_eigh_slq_workspace_bytes(int64_t P, int64_t B) { return (size_t)P * B * sizeof(double) + 512; }

int lo_tridiag_eigh_slq_f32(const float* t_mat, int64_t P, int64_t B, int32_t T, int64_t n, float* evals, float* evecs,
                            float* logdet, void* ws, size_t ws_bytes, void* stream) {
  if (!t_mat || !ws || P < 1 || B < 1 || T < 1) return LO_ERR_BADARG;
  if (T > kEigMaxT) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = P * B;
  Arena ar(ws, ws_bytes);
  int* fail = ar.take<int>(1);
  double* term = ar.take<double>((size_t)M);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  LO_HIP_CHECK(hipMemsetAsync(fail, 0, sizeof(int), st));
  const unsigned grid = (unsigned)((M + 63) / 64);
  if (evecs)
    hipLaunchKernelGGL((k_tridiag_eigh<true>), dim3(grid), dim3(64), 0, st, t_mat, M, (int)T, evals, evecs, term, fail);
  else
    hipLaunchKernelGGL((k_tridiag_eigh<false>), dim3(grid), dim3(64), 0, st, t_mat, M, (int)T, evals, evecs, term,
                       fail);
  LO_LAUNCH_CHECK();
  if (logdet) {
    hipLaunchKernelGGL(k_slq_reduce, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, term, P, B,
                       (double)n / (double)P, logdet);
    LO_LAUNCH_CHECK();
  }
  return LO_OK;
}

}  // extern "C"
