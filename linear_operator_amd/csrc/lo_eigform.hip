// lo_eigform.hip -- the DIAGONAL form of the R-space iteration (lo_precond_desc.RSD, round 5 second half).
//
// lo_rspace.hip runs linear_cg (linear_operator/utils/linear_cg.py:245-332) of A = C C^T + D with the root-form
// preconditioner P^-1 = D^-1 - D^-1 C F C^T D^-1 on the R + 1 coordinates of span{r0, C}; every iteration there costs four
// R x R products and eleven inner products in ONE dependent chain (0.68 us per iteration, 8.9 of the 23.5 us a member stays
// resident).  With C^ = D^-1/2 C = U S V^T (E = C^T D^-1 C = V S^2 V^T) both A^ = I + C^ C^^T and P^^-1 = I - C^ F C^^T map
// span(U) to itself and are the identity on its complement; in U-coordinates A_U = I + S^2 =: Gam^2 (diagonal),
// P_U^-1 = I - S V^T F V S, and the SYMMETRIC matrix Hs = Gam P_U^-1 Gam = Q Lam Q^T gives W = Gam^-1 Q Lam^1/2 with
// W^T P_U W = I and W^T A_U W = Lam.  In the coordinates c of r^ = P^ (U W c + c' b_perp) (b_perp = the part of the
// right-hand side outside span(U)) the preconditioned CG is the CG of a diagonal matrix:
//     r.z = |c|^2 + c'^2 tau2,   p.Ap = sum lam q^2 + q'^2 tau2,   c -= alpha lam q,   q = c + beta q,   eta += alpha q
// -- one reduction of three values per iteration, no matrix product on the chain.  This file builds, per member and in fp64,
//     TinT = (V S^-1 W)^T     c0 = TinT w0                       (w0 = C^T D^-1 r0)
//     Ep   = V S^-2 V^T       (E^+, kept for the tests' identities; the kernel does not use it: it squares cond(S))
//     TuT  = (V S^-1 W^-T)^T  g0 = TuT w0 = W^-1 beta0:  tau2 = s - c0.g0,  y = Tin (eta - xi g0),  x = D^-1 (xi r0 + C y)
//     G2   = C^T C            residual norms in the coordinates of C:  r = c' r0 + C g,  g = Tu (c - c' c0),
//                             r^T r = c'^2 a0 + 2 c' g.u0 + g.G2 g     (u0 = C^T r0)
//     Tin  = V S^-1 W         (kept for reference; the kernel walks the columns of TinT)
//     lam
// from the R-space form RS (E | . | . | G2 = C^T C | F | .) with two cyclic Jacobi eigendecompositions in LDS (E, then Hs).
// Directions of E below 1e-13 of its largest eigenvalue are dropped (S^-1 := 0: they stay in the complement, where both
// operators are the identity to that accuracy) -- rank-deficient roots are fine.  Numerics first:
// tests/proto/proto_eigform.py (alphas / solutions identical to the dense R-space iteration at fp32 resolution on all
// of proto_rspace.py's cases plus duplicated / zero columns of C).
#include <stdlib.h>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int EF_N = 32;       // largest padded root rank
constexpr int EF_LD = EF_N + 1;
constexpr int EF_MAX_SWEEPS = 24;
constexpr double EF_RANK_TOL = 1e-13;
constexpr double EF_AMP_MAX = 1e9;    // (s_max / s_min) (1 + s_max^2) of a member's kept directions: beyond it the dense form is kept (eps x this = 1e-7)
constexpr double EF_ROT_TOL = 1e-15;  // |a_pq| <= tol sqrt(a_pp a_qq): converged (relative criterion: small eigenvalues of E keep their digits)

// Parallel cyclic Jacobi on the symmetric n x n matrix A (LDS, n even, <= 32): n / 2 disjoint rotations per round in the
// round-robin order, thread (a, b) owns the 2 x 2 block of A (rows of pair a, columns of pair b) and of V (V <- V J).
// Ends when a whole sweep applied no rotation (|a_pq| <= EF_ROT_TOL sqrt(a_pp a_qq), or 0) or after EF_MAX_SWEEPS.
__device__ __forceinline__ void pair_of(int r, int k, int n, int& p, int& q) {
  const int m = n - 1;
  if (k == 0) {
    p = m;
    q = r;
  } else {
    p = (r + k) % m;
    q = (r - k + m) % m;
  }
  if (p > q) { const int t = p; p = q; q = t; }
}
// The rotation that annihilates a_pq, J = [[c, s], [-s, c]], from the double angle: with z = a_qq - a_pp and
// h = sqrt(z^2 + 4 a_pq^2), cos 2phi = |z| / h, c = sqrt((1 + cos 2phi) / 2), s = sign(z) a_pq / (h c).  Two
// reciprocal-square-root instructions with one Newton step each and a renormalisation so that c^2 + s^2 = 1 to fp64
// rounding: s keeps a RELATIVE error of ~1e-9 (the entry drops by that factor instead of to zero, and small angles stay
// small), the product of the rotations stays orthogonal.  The IEEE sequence (two divisions,
// two square roots: ~60 dependent instructions) was the longest part of a round.
__device__ __forceinline__ bool rotation(double app, double aqq, double apq, double floor_abs, double& c, double& s) {
  c = 1.0;
  s = 0.0;
  const double lim = EF_ROT_TOL * __builtin_amdgcn_sqrt(fabs(app) * fabs(aqq));
  if (!(fabs(apq) > lim) || apq == 0.0 || (fabs(app) <= floor_abs && fabs(aqq) <= floor_abs)) return false;
  // (v_rsq_f64 is good to ~1e-5 only -- measured: rotations from the raw value left Q orthogonal to 6e-9 -- so each
  //  reciprocal square root gets one Newton step, and the final renormalisation two: c^2 + s^2 = 1 to fp64 rounding)
  const double z = aqq - app;
  const double h2 = fma(z, z, 4.0 * apq * apq);
  double r = __builtin_amdgcn_rsq(h2);                                  // 1 / h
  r = r * fma(-0.5 * h2, r * r, 1.5);
  const double c2 = fma(0.5 * fabs(z), r, 0.5);                         // cos^2 phi in [0.5, 1]
  double rc = __builtin_amdgcn_rsq(c2);
  rc = rc * fma(-0.5 * c2, rc * rc, 1.5);
  const double ct = c2 * rc;
  const double st = (z >= 0.0 ? apq : -apq) * r * rc;
  const double nu = fma(ct, ct, st * st);
  double sc = fma(-0.5, nu, 1.5);                                       // nu^-1/2: first order, then one Newton step
  sc = sc * fma(-0.5 * nu, sc * sc, 1.5);
  if (!(sc == sc)) return false;                                       // (overflow of z^2: the entry is negligible)
  c = ct * sc;
  s = st * sc;
  return true;
}
// `pairs`: the round-robin schedule, [round][pair] = p | q << 8 (filled by the caller: the index arithmetic costs more than a round).
// `floor_abs`: entries between two directions whose diagonal is below it are left alone (the null space of a rank-deficient
// E: its entries are rounding noise without a relative scale and both directions are dropped afterwards).
__device__ int jacobi_eigh(double (*A)[EF_LD], double (*V)[EF_LD], double (*A2)[EF_LD], double (*V2)[EF_LD], int n,
                           const unsigned short (*pairs)[EF_N / 2], double floor_abs) {
  const int t = threadIdx.x;
  const int np = n / 2;
  const bool act = t < np * np;
  const int ia = t / np, ib = t - ia * np;
  for (int e = t; e < n * n; e += kThreads) V[e / n][e % n] = (e / n == e % n) ? 1.0 : 0.0;
  __syncthreads();
  double (*Ac)[EF_LD] = A, (*Vc)[EF_LD] = V, (*An)[EF_LD] = A2, (*Vn)[EF_LD] = V2;
  int sweeps = 0;
  for (; sweeps < EF_MAX_SWEEPS; ++sweeps) {
    int rotated = 0;
    for (int r = 0; r < n - 1; ++r) {
      if (act) {
        const unsigned wa = pairs[r][ia], wb = pairs[r][ib];
        const int pa = wa & 0xff, qa = wa >> 8, pb = wb & 0xff, qb = wb >> 8;
        double ca, sa, cb, sb;
        const bool ra = rotation(Ac[pa][pa], Ac[qa][qa], Ac[pa][qa], floor_abs, ca, sa);
        const bool rb = rotation(Ac[pb][pb], Ac[qb][qb], Ac[pb][qb], floor_abs, cb, sb);
        rotated |= (ra || rb) ? 1 : 0;
        const double b00 = Ac[pa][pb], b01 = Ac[pa][qb], b10 = Ac[qa][pb], b11 = Ac[qa][qb];
        const double v00 = Vc[pa][pb], v01 = Vc[pa][qb], v10 = Vc[qa][pb], v11 = Vc[qa][qb];
        // B' = Ja^T B Jb
        const double r00 = ca * b00 - sa * b10, r01 = ca * b01 - sa * b11;
        const double r10 = sa * b00 + ca * b10, r11 = sa * b01 + ca * b11;
        An[pa][pb] = cb * r00 - sb * r01; An[pa][qb] = sb * r00 + cb * r01;
        An[qa][pb] = cb * r10 - sb * r11; An[qa][qb] = sb * r10 + cb * r11;
        Vn[pa][pb] = cb * v00 - sb * v01; Vn[pa][qb] = sb * v00 + cb * v01;
        Vn[qa][pb] = cb * v10 - sb * v11; Vn[qa][qb] = sb * v10 + cb * v11;
      }
      __syncthreads();
      double (*ta)[EF_LD] = Ac; Ac = An; An = ta;
      double (*tv)[EF_LD] = Vc; Vc = Vn; Vn = tv;
    }
    if (!__syncthreads_or(rotated)) {
      ++sweeps;
      break;
    }
  }
  if (Ac != A) {
    for (int e = t; e < n * n; e += kThreads) {
      A[e / n][e % n] = Ac[e / n][e % n];
      V[e / n][e % n] = Vc[e / n][e % n];
    }
  }
  __syncthreads();
  return sweeps;
}

// dst[i][j] = sum_q X(i, q) Y(q, j), X / Y read through the transposition flags; n x n, all in LDS
template <bool TX, bool TY>
__device__ __forceinline__ void mm(double (*dst)[EF_LD], const double (*X)[EF_LD], const double (*Y)[EF_LD], int n) {
  for (int e = threadIdx.x; e < n * n; e += kThreads) {
    const int i = e / n, j = e % n;
    double acc = 0.0;
    for (int q = 0; q < n; ++q) acc = fma(TX ? X[q][i] : X[i][q], TY ? Y[j][q] : Y[q][j], acc);
    dst[i][j] = acc;
  }
}

__global__ __launch_bounds__(kThreads) void k_rs_eigform(const double* __restrict__ RS, int n, int ld,
                                                         double* __restrict__ RSD) {
  __shared__ double M0[EF_N][EF_LD], M1[EF_N][EF_LD], M2[EF_N][EF_LD], M3[EF_N][EF_LD], M4[EF_N][EF_LD], M5[EF_N][EF_LD];
  __shared__ double S_s[EF_N], Sinv_s[EF_N], Gam_s[EF_N], lam_s[EF_N];
  __shared__ int flag_s;
  __shared__ double floor_s;
  __shared__ unsigned short pairs_s[EF_N - 1][EF_N / 2];
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  for (int e = t; e < (n - 1) * (n / 2); e += kThreads) {
    int p, q;
    pair_of(e / (n / 2), e % (n / 2), n, p, q);
    pairs_s[e / (n / 2)][e % (n / 2)] = (unsigned short)(p | (q << 8));
  }
  const double* Rb = RS + (size_t)b * 6 * ld * ld;
  double* Ob = RSD + (size_t)b * 6 * ld * ld;
  for (int e = t; e < 6 * ld * ld; e += kThreads) Ob[e] = 0.0;
  for (int e = t; e < n * n; e += kThreads) {
    const int i = e / n, j = e % n;
    M0[i][j] = 0.5 * (Rb[i * ld + j] + Rb[j * ld + i]);  // E
    M2[i][j] = Rb[4 * ld * ld + i * ld + j];              // F
  }
  __syncthreads();
  if (t == 0) {
    double mx = 0.0;
    for (int j = 0; j < n; ++j) mx = fmax(mx, fabs(M0[j][j]));
    floor_s = 0.1 * EF_RANK_TOL * mx;
  }
  __syncthreads();
  const int sweeps_e = jacobi_eigh(M0, M1, M3, M4, n, pairs_s, floor_s);  // E = V diag(s2) V^T, V in M1 (M3 / M4: the other buffers)
  if (t == 0) {
    double mx = 0.0;
    for (int j = 0; j < n; ++j) mx = fmax(mx, M0[j][j]);
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double s2 = M0[j][j];
      const bool keep = s2 > EF_RANK_TOL * mx && s2 > 0.0;
      rank += keep ? 1 : 0;
      S_s[j] = keep ? sqrt(s2) : 0.0;
      Sinv_s[j] = keep ? 1.0 / sqrt(s2) : 0.0;
      Gam_s[j] = sqrt(1.0 + (keep ? s2 : 0.0));
    }
    // The coordinates c0 = TinT w0 mix the components of w0 (rounded at eps |w0|max) with weights up to 1 / s_min: the
    // right-hand side's component along a small direction is known to eps s_max / s_min only, and a solution that is the
    // difference of large terms (x = D^-1 (xi b + C y) for b inside span(C): |x^| ~ |b^| / (1 + s_max^2)) inherits that
    // error relative to its own size: eps (s_max / s_min) (1 + s_max^2).  Measured 8e-4 on such a right-hand side with
    // one column of C scaled by 1e-6 and d ~ 1e-3 (the dense form, which stays in the coordinates of C, is at 1e-7).
    // Members beyond EF_AMP_MAX keep the dense form.
    double mn = mx;
    for (int j = 0; j < n; ++j)
      if (S_s[j] > 0.0) mn = fmin(mn, M0[j][j]);
    flag_s = (sqrt(mx / fmax(mn, 1e-300)) * (1.0 + mx) > EF_AMP_MAX) ? -rank - 1 : rank;
  }
  __syncthreads();
  const bool illcond = flag_s < 0;
  const int rank = illcond ? -flag_s - 1 : flag_s;
  __syncthreads();
  mm<false, false>(M3, M2, M1, n);  // F V
  __syncthreads();
  mm<true, false>(M2, M1, M3, n);   // X1 = V^T F V
  __syncthreads();
  for (int e = t; e < n * n; e += kThreads) {  // Hs = Gam (I - S X1 S) Gam
    const int i = e / n, j = e % n;
    const double x = 0.5 * (M2[i][j] + M2[j][i]);
    M0[i][j] = Gam_s[i] * ((i == j ? 1.0 : 0.0) - S_s[i] * x * S_s[j]) * Gam_s[j];
  }
  __syncthreads();
  const int sweeps_h = jacobi_eigh(M0, M3, M4, M5, n, pairs_s, 0.0);  // Hs = Q Lam Q^T, Q in M3 (M2 = X1 is dead; M4 / M5 free)
  if (t < n) lam_s[t] = M0[t][t];
  __syncthreads();
  bool bad = illcond;
  for (int j = 0; j < n; ++j) bad = bad || !(lam_s[j] > 0.0);
  for (int e = t; e < n * n; e += kThreads) {
    const int i = e / n, j = e % n;
    const double q = M3[i][j], sl = sqrt(bad ? 1.0 : lam_s[j]);
    M4[i][j] = q * sl / Gam_s[i];        // W
    M5[i][j] = Gam_s[i] * q / sl;        // W^-T
    M1[i][j] *= Sinv_s[j];               // V S^-1
  }
  __syncthreads();
  mm<false, false>(M0, M1, M4, n);       // Tin = V S^-1 W
  mm<false, false>(M3, M1, M5, n);       // Tu = V S^-1 W^-T
  mm<false, true>(M2, M1, M1, n);        // Ep = V S^-2 V^T
  __syncthreads();
  for (int e = t; e < n * n; e += kThreads) {
    const int i = e / n, j = e % n;
    Ob[j * ld + i] = M0[i][j];                    // slot 0: TinT
    Ob[1 * ld * ld + i * ld + j] = M2[i][j];      // slot 1: Ep
    Ob[2 * ld * ld + j * ld + i] = M3[i][j];      // slot 2: TuT
    Ob[4 * ld * ld + i * ld + j] = M0[i][j];      // slot 4: Tin
  }
  __syncthreads();
  // slot 3: G2 = C^T C itself -- the residual norm is formed in the coordinates of C (g = Tu del, r^T r = rho^2 a0 +
  // 2 rho g.u0 + g.G2 g as the dense form does): through U^T D U = S^-1 V^T G2 V S^-1 its rounding would be amplified by cond(E)
  for (int e = t; e < n * n; e += kThreads) {
    const int i = e / n, jj = e % n;
    Ob[3 * ld * ld + i * ld + jj] = 0.5 * (Rb[3 * ld * ld + i * ld + jj] + Rb[3 * ld * ld + jj * ld + i]);
  }
  if (t < ld) Ob[5 * ld * ld + t] = (t < n && !bad) ? lam_s[t] : 1.0;
  if (t == 0) {
    Ob[5 * ld * ld + ld + 0] = bad ? (illcond ? -2.0 : -1.0) : 1.0;  // status: 1 = usable, -1 = Lam not positive, -2 = ill-conditioned basis
    Ob[5 * ld * ld + ld + 1] = (double)sweeps_e;
    Ob[5 * ld * ld + ld + 2] = (double)sweeps_h;
    Ob[5 * ld * ld + ld + 3] = (double)rank;
  }
}

}  // namespace lo

using namespace lo;

extern "C" {

int lo_precond_eigform_f32(const double* RS, int64_t B, int32_t R, int32_t rf_ld, double* RSD, void* stream) {
  if (!RS || !RSD) return LO_ERR_BADARG;
  if (B < 1 || R < 2 || (R % 2) != 0 || R > EF_N || rf_ld < R || rf_ld > EF_N || rf_ld < 4) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  LO_PROF_BEGIN("rs_eigform", st);
  hipLaunchKernelGGL(k_rs_eigform, dim3((unsigned)B), dim3(kThreads), 0, st, RS, (int)R, (int)rf_ld, RSD);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // extern "C"
