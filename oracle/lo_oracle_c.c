/* lo_oracle_c.c -- CPU ORACLE in C: test infrastructure only, never a product path.
 *
 * A plain-C (C11 + OpenMP over the batch members) restatement of the reference's iterative solve hot path with the
 * argument structures of include/lo_amd.h and HOST pointers: the structured matvecs, linear_cg, pivoted Cholesky and
 * the Woodbury preconditioner -- "a CPU build of the same ABI" in the sense of SURVEY.md section 8(b), kept under
 * oracle/ because that is all it may ever be: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
 * it (oracle/lo_oracle_c.py), and only as the checker / the timed CPU baseline.  The shipped package has no CPU path.
 *
 * Parity is PINNED: tests/test_oracle_c.py checks every entry against the golden vectors the real reference produced
 * (the .npz files under tests/golden, make_golden.py) and against the numpy oracle (oracle/lo_oracle.py): pivots / permutations and the
 * factor L bit for bit (same operation order: products rounded before sequential sums, no FMA -- this file is compiled
 * with -ffp-contract=off), solutions and tridiagonals to 1e-5.
 *
 * Every function cites the reference file:line it restates (paths relative to linear_operator/).
 *
 * Layouts as in lo_amd.h: vectors [B, N, c] (column innermost), roots C [B, N, R], dense K [B, N, N], Kronecker factors
 * [B, n_i, n_i], diagonal [B, N] or [B], permutation int64 [B, N], L_rows [B, max_rank, N], t_mat [n_tridiag, B, T, T].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/lo_amd.h"

#if defined(_OPENMP)
#include <omp.h>
#endif

/* Threads a parallel region over `work` independent members uses: never more than the members, never more than the
 * limit the host wrapper derived from the CPUs this process may actually run on (affinity mask / cgroup quota -- the
 * container of a GPU box may see 128 cores and own a few; 128 spinning OpenMP threads on them ran 100x slower). */
static int g_thread_limit = 0;
int lo_cpu_set_num_threads(int n) {
  g_thread_limit = n > 0 ? n : 0;
  return LO_OK;
}
int lo_cpu_num_threads(void) {
#if defined(_OPENMP)
  const int m = omp_get_max_threads();
  return (g_thread_limit > 0 && g_thread_limit < m) ? g_thread_limit : m;
#else
  return 1;
#endif
}
static int nthr(int64_t work) {
  const int t = lo_cpu_num_threads();
  return (int)(work < t ? (work < 1 ? 1 : work) : t);
}

/* ------------------------------------------------------------------------------------------------------------------
 * structured matvecs: the `_matmul`s that feed CG
 *   AddedDiagLinearOperator._matmul  operators/added_diag_linear_operator.py:72-76   (+ d o v)
 *   RootLinearOperator._matmul       operators/root_linear_operator.py:68-72         (C (C^T v))
 *   DenseLinearOperator._matmul      operators/dense_linear_operator.py:60-64        (K v)
 *   Kronecker module-level _matmul   operators/kronecker_product_linear_operator.py:34-45  (vec(K1 V K2^T))
 *   SumLinearOperator._matmul        operators/sum_linear_operator.py:47-51          (terms left to right)
 * ------------------------------------------------------------------------------------------------------------------ */
static void term_matvec_member(const lo_op_desc* op, int64_t b, const float* v, float* y, int64_t c, float* tmp) {
  const int64_t N = op->N;
  if (op->kind == LO_OP_LOWRANK_DIAG) {
    const int64_t R = op->R;
    const float* C = op->A0 + (size_t)b * N * R;
    float* t = tmp; /* [R, c] */
    for (int64_t e = 0; e < R * c; ++e) t[e] = 0.f;
    for (int64_t i = 0; i < N; ++i)
      for (int64_t r = 0; r < R; ++r) {
        const float cir = C[i * R + r];
        for (int64_t k = 0; k < c; ++k) t[r * c + k] += cir * v[i * c + k];
      }
    for (int64_t i = 0; i < N; ++i)
      for (int64_t k = 0; k < c; ++k) {
        float acc = 0.f;
#pragma omp simd reduction(+ : acc)
        for (int64_t r = 0; r < R; ++r) acc += C[i * R + r] * t[r * c + k];
        y[i * c + k] = acc;
      }
  } else if (op->kind == LO_OP_DENSE_DIAG) {
    const float* K = op->A0 + (size_t)b * N * N;
    for (int64_t i = 0; i < N; ++i) {
      for (int64_t k = 0; k < c; ++k) y[i * c + k] = 0.f;
      for (int64_t j = 0; j < N; ++j) {
        const float kij = K[i * N + j];
        for (int64_t k = 0; k < c; ++k) y[i * c + k] += kij * v[j * c + k];
      }
    }
  } else { /* LO_OP_KRON_DIAG: T = V K2^T per row block, then Y = K1 T */
    const int64_t n1 = op->R, n2 = op->n2;
    const float* K1 = op->A0 + (size_t)b * n1 * n1;
    const float* K2 = op->A1 + (size_t)b * n2 * n2;
    float* T = tmp; /* [n1, n2, c] */
    for (int64_t i1 = 0; i1 < n1; ++i1)
      for (int64_t j2 = 0; j2 < n2; ++j2)
        for (int64_t k = 0; k < c; ++k) {
          float acc = 0.f;
          for (int64_t i2 = 0; i2 < n2; ++i2) acc += v[(i1 * n2 + i2) * c + k] * K2[j2 * n2 + i2];
          T[(i1 * n2 + j2) * c + k] = acc;
        }
    for (int64_t m = 0; m < n1; ++m)
      for (int64_t e = 0; e < n2 * c; ++e) {
        float acc = 0.f;
        for (int64_t i1 = 0; i1 < n1; ++i1) acc += K1[m * n1 + i1] * T[i1 * n2 * c + e];
        y[m * n2 * c + e] = acc;
      }
  }
}

static size_t term_tmp_floats(const lo_op_desc* op, int64_t c) {
  if (op->kind == LO_OP_LOWRANK_DIAG) return (size_t)op->R * c;
  if (op->kind == LO_OP_KRON_DIAG) return (size_t)op->N * c;
  return 1;
}

static size_t op_tmp_floats(const lo_op_desc* op, int64_t c) {
  size_t need = 1;
  if (op->kind == LO_OP_SUM) {
    for (int i = 0; i < op->nterms; ++i) {
      const size_t t = term_tmp_floats(&op->terms[i], c);
      need = t > need ? t : need;
    }
    return need + (size_t)op->N * c; /* + the buffer a term beyond the first is computed into */
  }
  return term_tmp_floats(op, c);
}

static void add_diag_member(const lo_op_desc* op, int64_t b, const float* v, float* y, int64_t c) {
  const int64_t N = op->N;
  if (op->diag_mode == LO_DIAG_FULL) {
    const float* d = op->d + (size_t)b * N;
    for (int64_t i = 0; i < N; ++i)
      for (int64_t k = 0; k < c; ++k) y[i * c + k] += d[i] * v[i * c + k];
  } else if (op->diag_mode == LO_DIAG_CONST) {
    const float s = op->d[b];
    for (int64_t e = 0; e < N * c; ++e) y[e] += s * v[e];
  }
}

/* y[b] = A[b] v[b] for ONE member; tmp holds op_tmp_floats floats */
static void matvec_member(const lo_op_desc* op, int64_t b, const float* v, float* y, int64_t c, float* tmp) {
  if (op->kind == LO_OP_SUM) {
    float* ytmp = tmp;
    float* ttmp = tmp + (size_t)op->N * c;
    for (int i = 0; i < op->nterms; ++i) {
      if (i == 0) {
        term_matvec_member(&op->terms[0], b, v, y, c, ttmp);
      } else {
        term_matvec_member(&op->terms[i], b, v, ytmp, c, ttmp);
        for (int64_t e = 0; e < op->N * c; ++e) y[e] += ytmp[e];
      }
    }
  } else {
    term_matvec_member(op, b, v, y, c, tmp);
  }
  add_diag_member(op, b, v, y, c);
}

static int op_ok(const lo_op_desc* op) {
  if (!op || op->B < 1 || op->N < 1) return 0;
  if (op->kind == LO_OP_SUM) {
    if (op->nterms < 2 || op->nterms > LO_MAX_TERMS || !op->terms) return 0;
    for (int i = 0; i < op->nterms; ++i) {
      const lo_op_desc* t = &op->terms[i];
      if (t->kind != LO_OP_LOWRANK_DIAG && t->kind != LO_OP_DENSE_DIAG && t->kind != LO_OP_KRON_DIAG) return 0;
      if (t->B != op->B || t->N != op->N) return 0;
    }
    return 1;
  }
  return op->kind == LO_OP_LOWRANK_DIAG || op->kind == LO_OP_DENSE_DIAG || op->kind == LO_OP_KRON_DIAG;
}

int lo_cpu_matvec_f32(const lo_op_desc* op, const float* v, float* y, int64_t c) {
  if (!op_ok(op) || !v || !y || c < 1) return LO_ERR_BADARG;
  const int64_t B = op->B, N = op->N;
  const size_t nt = op_tmp_floats(op, c);
  int bad = 0;
#pragma omp parallel num_threads(nthr(B))
  {
    float* tmp = (float*)malloc(sizeof(float) * nt);
    if (!tmp) {
#pragma omp atomic write
      bad = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b)
      if (tmp) matvec_member(op, b, v + (size_t)b * N * c, y + (size_t)b * N * c, c, tmp);
    free(tmp);
  }
  return bad ? LO_ERR_WORKSPACE : LO_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * PivotedCholesky.forward  functions/_pivoted_cholesky.py:14-105 (+ row fetch utils/permutation.py:9-88 ->
 * operator _get_indices: Root root_linear_operator.py:37-50, Dense dense_linear_operator.py:47-50, Kronecker
 * kronecker_product_linear_operator.py:198-216, Sum sum_linear_operator.py:39-41; diagonals root..:22-28, dense..:37-40,
 * kronecker..:188-191, sum..:28-31).  One shared pivot count m for the whole batch (:57); first maximal index wins
 * (:61-63); products rounded before the sequential sums (:83-89).
 * ------------------------------------------------------------------------------------------------------------------ */
static float seq_dot(const float* a, const float* b, int64_t R) {
  float acc = a[0] * b[0];
  for (int64_t r = 1; r < R; ++r) acc = acc + a[r] * b[r];
  return acc;
}

static float term_diag(const lo_op_desc* t, int64_t b, int64_t i) {
  const int64_t N = t->N;
  if (t->kind == LO_OP_LOWRANK_DIAG) {
    const float* ci = t->A0 + ((size_t)b * N + i) * t->R;
    return seq_dot(ci, ci, t->R);
  } else if (t->kind == LO_OP_DENSE_DIAG) {
    return t->A0[((size_t)b * N + i) * N + i];
  }
  const int64_t n1 = t->R, n2 = t->n2, i1 = i / n2, i2 = i % n2;
  return t->A0[((size_t)b * n1 + i1) * n1 + i1] * t->A1[((size_t)b * n2 + i2) * n2 + i2];
}

static float term_entry(const lo_op_desc* t, int64_t b, int64_t p, int64_t i) { /* K[b, p, i] */
  const int64_t N = t->N;
  if (t->kind == LO_OP_LOWRANK_DIAG) {
    return seq_dot(t->A0 + ((size_t)b * N + p) * t->R, t->A0 + ((size_t)b * N + i) * t->R, t->R);
  } else if (t->kind == LO_OP_DENSE_DIAG) {
    return t->A0[((size_t)b * N + p) * N + i];
  }
  const int64_t n1 = t->R, n2 = t->n2;
  return t->A0[((size_t)b * n1 + p / n2) * n1 + i / n2] * t->A1[((size_t)b * n2 + p % n2) * n2 + i % n2];
}

int lo_cpu_pivoted_cholesky_f32(const lo_op_desc* op, int32_t max_rank, float error_tol, float* L_rows, int64_t* perm,
                                int32_t* rank_out) {
  if (!op_ok(op) || !L_rows || !perm || !rank_out || max_rank < 1) return LO_ERR_BADARG;
  const int64_t B = op->B, N = op->N;
  const int nterms = op->kind == LO_OP_SUM ? op->nterms : 1;
  const lo_op_desc* terms = op->kind == LO_OP_SUM ? op->terms : op;
  const int rank = (int)(max_rank < N ? max_rank : N); /* :33 */
  float* diag = (float*)malloc(sizeof(float) * (size_t)B * N);
  float* orig = (float*)malloc(sizeof(float) * (size_t)B);
  float* errors = (float*)malloc(sizeof(float) * (size_t)B);
  if (!diag || !orig || !errors) {
    free(diag); free(orig); free(errors);
    return LO_ERR_WORKSPACE;
  }
  memset(L_rows, 0, sizeof(float) * (size_t)B * max_rank * N); /* :36-42 */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr(B))
  for (int64_t b = 0; b < B; ++b) {
    float mx = -INFINITY, l1 = 0.f;
    for (int64_t i = 0; i < N; ++i) {
      float v = term_diag(&terms[0], b, i);
      for (int it = 1; it < nterms; ++it) v = v + term_diag(&terms[it], b, i); /* left to right */
      diag[(size_t)b * N + i] = v;
      perm[(size_t)b * N + i] = i; /* :47-48 */
      if (v > mx) mx = v;
      l1 += fabsf(v);
    }
    orig[b] = mx;          /* :43 */
    errors[b] = l1 / mx;   /* :44 */
  }
  int m = 0;
  for (;;) {
    if (m > 0) { /* loop condition :57 -- torch.max propagates NaN and (NaN > tol) is False */
      float emax = -INFINITY;
      int anynan = 0;
      for (int64_t b = 0; b < B; ++b) {
        if (errors[b] != errors[b]) anynan = 1;
        if (errors[b] > emax) emax = errors[b];
      }
      if (m >= rank || anynan || !(emax > error_tol)) break;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr(B))
    for (int64_t b = 0; b < B; ++b) {
      int64_t* pi = perm + (size_t)b * N;
      float* dg = diag + (size_t)b * N;
      float* Lb = L_rows + (size_t)b * max_rank * N;
      /* :61-63 first maximal index among the not-yet-pivoted positions */
      int64_t best = m;
      float bv = dg[pi[m]];
      for (int64_t j = m + 1; j < N; ++j) {
        const float v = dg[pi[j]];
        if (v > bv) {
          bv = v;
          best = j;
        }
      }
      const int64_t old = pi[m], pim = pi[best]; /* :67-70 */
      pi[m] = pim;
      pi[best] = old;
      const float piv = sqrtf(bv);
      Lb[(size_t)m * N + pim] = piv; /* :73-74 */
      if (m + 1 < N) { /* :77 */
        float l1 = 0.f;
        for (int64_t j = m + 1; j < N; ++j) {
          const int64_t i = pi[j];
          float row = term_entry(&terms[0], b, pim, i); /* :79-82 */
          for (int it = 1; it < nterms; ++it) row = row + term_entry(&terms[it], b, pim, i);
          float v = row;
          if (m > 0) { /* :83-89 */
            float acc = Lb[pim] * Lb[i];
            for (int jj = 1; jj < m; ++jj) acc = acc + Lb[(size_t)jj * N + pim] * Lb[(size_t)jj * N + i];
            v = row - acc;
          }
          v = v / piv;                    /* :91 */
          Lb[(size_t)m * N + i] = v;      /* :92 */
          const float dn = dg[i] - v * v; /* :94-95 */
          dg[i] = dn;
          l1 += fabsf(dn);
        }
        errors[b] = l1 / orig[b]; /* :99 */
      }
    }
    ++m;
  }
  *rank_out = m;
  free(diag); free(orig); free(errors);
  return LO_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * AddedDiagLinearOperator._init_cache*  operators/added_diag_linear_operator.py:144-184 and precondition_closure :135-140.
 * Only Q Q^T and |R_ii| are used downstream (the QR's sign / rotation freedom drops out), so the thin QR of
 * [L / sqrt(d); I] is formed through its k x k Gram matrix G = I + W^T W = R^T R in double precision:
 * Q = W R^-1 / sqrt(d), logdet P = 2 sum log R_ii + sum log d   (constant diagonal: the same with d = sigma).
 *   L [B, N, k], d [B, N] | [B];  Q [B, N, k], dinv like d, logdet_p [B]
 * ------------------------------------------------------------------------------------------------------------------ */
int lo_cpu_precond_build_f32(const float* L, const float* d, int32_t diag_mode, int64_t B, int64_t N, int32_t k, float* Q,
                             float* dinv, float* logdet_p) {
  if (!L || !d || !Q || !dinv || !logdet_p || B < 1 || N < 1 || k < 1 || k > 256) return LO_ERR_BADARG;
  if (diag_mode != LO_DIAG_FULL && diag_mode != LO_DIAG_CONST) return LO_ERR_BADARG;
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr(B))
  for (int64_t b = 0; b < B; ++b) {
    double* G = (double*)calloc((size_t)k * k, sizeof(double));
    double* X = (double*)calloc((size_t)k * k, sizeof(double));
    if (!G || !X) {
#pragma omp atomic write
      bad = 1;
      free(G); free(X);
      continue;
    }
    const float* Lb = L + (size_t)b * N * k;
    double slogd = 0.0;
    for (int64_t i = 0; i < N; ++i) {
      const double di = diag_mode == LO_DIAG_FULL ? (double)d[(size_t)b * N + i] : (double)d[b];
      slogd += log(di);
      for (int a = 0; a < k; ++a) {
        const double wa = (double)Lb[i * k + a] / di;
        for (int c2 = 0; c2 <= a; ++c2) G[a * k + c2] += wa * (double)Lb[i * k + c2];
      }
    }
    for (int a = 0; a < k; ++a) G[a * k + a] += 1.0;
    /* Cholesky G = T T^T (lower), in place in the lower triangle */
    double ld = 0.0;
    for (int j = 0; j < k; ++j) {
      double s = G[j * k + j];
      for (int p = 0; p < j; ++p) s -= G[j * k + p] * G[j * k + p];
      const double t = sqrt(s);
      G[j * k + j] = t;
      ld += log(t);
      for (int i = j + 1; i < k; ++i) {
        double u = G[i * k + j];
        for (int p = 0; p < j; ++p) u -= G[i * k + p] * G[j * k + p];
        G[i * k + j] = u / t;
      }
    }
    /* X = T^-1 (lower) */
    for (int j = 0; j < k; ++j) {
      X[j * k + j] = 1.0 / G[j * k + j];
      for (int i = j + 1; i < k; ++i) {
        double u = 0.0;
        for (int p = j; p < i; ++p) u -= G[i * k + p] * X[p * k + j];
        X[i * k + j] = u / G[i * k + i];
      }
    }
    /* Q[i, :] = (L[i, :] / d_i) T^-T = rows of  (L / d) X^T   (R = T^T, R^-1 = X^T) */
    for (int64_t i = 0; i < N; ++i) {
      const double di = diag_mode == LO_DIAG_FULL ? (double)d[(size_t)b * N + i] : (double)d[b];
      for (int a = 0; a < k; ++a) {
        double u = 0.0;
        for (int p = 0; p <= a; ++p) u += (double)Lb[i * k + p] * X[a * k + p];
        Q[((size_t)b * N + i) * k + a] = (float)(u / di);
      }
      if (diag_mode == LO_DIAG_FULL) dinv[(size_t)b * N + i] = (float)(1.0 / di);
    }
    if (diag_mode == LO_DIAG_CONST) dinv[b] = (float)(1.0 / (double)d[b]);
    logdet_p[b] = (float)(2.0 * ld + slogd);
    free(G); free(X);
  }
  return bad ? LO_ERR_WORKSPACE : LO_OK;
}

/* z = r o dinv - Q (Q^T r) for ONE member (in both diagonal cases: Q carries the 1/sqrt(sigma) factor, lo_amd.h) */
static void precond_apply_member(const lo_precond_desc* pre, int64_t b, int64_t N, const float* r, float* z, int64_t c,
                                 float* u /* [k, c] */) {
  const int k = pre->k, ldq = pre->ldq;
  const float* Q = pre->Q + (size_t)b * N * ldq;
  for (int64_t e = 0; e < (int64_t)k * c; ++e) u[e] = 0.f;
  for (int64_t i = 0; i < N; ++i)
    for (int a = 0; a < k; ++a) {
      const float q = Q[i * ldq + a];
      for (int64_t j = 0; j < c; ++j) u[a * c + j] += q * r[i * c + j];
    }
  for (int64_t i = 0; i < N; ++i) {
    const float di = pre->constant_diag ? pre->dinv[b] : pre->dinv[(size_t)b * N + i];
    for (int64_t j = 0; j < c; ++j) {
      float acc = 0.f;
#pragma omp simd reduction(+ : acc)
      for (int a = 0; a < k; ++a) acc += Q[i * ldq + a] * u[a * c + j];
      z[i * c + j] = r[i * c + j] * di - acc;
    }
  }
}

int lo_cpu_precond_apply_f32(const lo_precond_desc* pre, const float* r, float* z, int64_t B, int64_t N, int64_t c) {
  if (!pre || !pre->Q || !pre->dinv || !r || !z || pre->k < 1) return LO_ERR_BADARG;
  int bad = 0;
#pragma omp parallel num_threads(nthr(B))
  {
    float* u = (float*)malloc(sizeof(float) * (size_t)pre->k * c);
    if (!u) {
#pragma omp atomic write
      bad = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b)
      if (u) precond_apply_member(pre, b, N, r + (size_t)b * N * c, z + (size_t)b * N * c, c, u);
    free(u);
  }
  return bad ? LO_ERR_WORKSPACE : LO_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * linear_cg  utils/linear_cg.py:98-359 (update helpers :16-95).  Same argument structures as lo_cg_solve_f32; the
 * operator is a descriptor or a host callback, the preconditioner a Woodbury descriptor, a host callback or none.
 * Per-member work runs in parallel; every BATCH-GLOBAL decision (mean residual :304, tridiagonal freeze :326, the
 * NaN check :199-200, the skip rule :207-208) is taken once per iteration from per-member values summed in member order.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  const lo_op_desc* op;
  lo_matvec_cb matvec;
  void* matvec_user;
  const lo_precond_desc* pre;
  lo_matvec_cb precond_cb;
  void* precond_user;
  int64_t B, N, c;
  size_t tmp_floats;
} cg_ctx;

static int cg_matvec(const cg_ctx* cx, const float* v, float* y) {
  if (cx->op->kind == LO_OP_CALLBACK) return cx->matvec(cx->matvec_user, v, y, cx->B, cx->N, cx->c, NULL) ? LO_ERR_LAUNCH : LO_OK;
  return lo_cpu_matvec_f32(cx->op, v, y, cx->c);
}

static int cg_precond(const cg_ctx* cx, const float* r, float* z) {
  if (cx->pre) return lo_cpu_precond_apply_f32(cx->pre, r, z, cx->B, cx->N, cx->c);
  if (cx->precond_cb) return cx->precond_cb(cx->precond_user, r, z, cx->B, cx->N, cx->c, NULL) ? LO_ERR_LAUNCH : LO_OK;
  memcpy(z, r, sizeof(float) * (size_t)cx->B * cx->N * cx->c); /* residual.clone() :82 */
  return LO_OK;
}

/* out[b, j] = sum_i a[b, i, j] * bb[b, i, j] */
static void col_dots(const float* a, const float* bb, float* out, int64_t B, int64_t N, int64_t c) {
#pragma omp parallel for schedule(static) num_threads(nthr(B))
  for (int64_t b = 0; b < B; ++b) {
    const float* ab = a + (size_t)b * N * c;
    const float* bp = bb + (size_t)b * N * c;
    for (int64_t j = 0; j < c; ++j) {
      float acc = 0.f;
#pragma omp simd reduction(+ : acc)
      for (int64_t i = 0; i < N; ++i) acc += ab[i * c + j] * bp[i * c + j];
      out[b * c + j] = acc;
    }
  }
}

int lo_cpu_cg_solve_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const lo_precond_desc* pre,
                        lo_matvec_cb precond_cb, void* precond_user, const lo_cg_params* prm, const float* rhs,
                        const float* x0, float* x, float* t_mat, lo_cg_info* info) {
  if (!op || !prm || !rhs || !x || !info) return LO_ERR_BADARG;
  if (op->kind == LO_OP_CALLBACK ? !matvec : !op_ok(op)) return LO_ERR_BADARG;
  if (prm->c < 1 || prm->n_tridiag < 0 || prm->n_tridiag > prm->c || (prm->n_tridiag && !t_mat)) return LO_ERR_BADARG;
  if (pre && precond_cb) return LO_ERR_BADARG;
  if (pre && (!pre->Q || !pre->dinv)) return LO_ERR_UNSUPPORTED; /* (the root form is a device-kernel representation) */
  const int64_t B = op->B, N = op->N, c = prm->c;
  const size_t nv = (size_t)B * N * c, ns = (size_t)B * c;
  const int nt = prm->n_tridiag, T = prm->max_tridiag_iter;
  const int fmi = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter; /* :303-305 use the caller's value */
  const float eps = prm->eps, stop_after = prm->stop_updating_after;
  cg_ctx cx = {op, matvec, matvec_user, pre, precond_cb, precond_user, B, N, c, 0};
  float* r = (float*)malloc(sizeof(float) * nv);
  float* z = (float*)malloc(sizeof(float) * nv);
  float* p = (float*)malloc(sizeof(float) * nv);
  float* Ap = (float*)malloc(sizeof(float) * nv);
  float* sc = (float*)calloc(8 * ns + 2 * (size_t)B * (nt > 0 ? nt : 1), sizeof(float));
  int* flags = (int*)calloc(2 * ns, sizeof(int));
  if (!r || !z || !p || !Ap || !sc || !flags) {
    free(r); free(z); free(p); free(Ap); free(sc); free(flags);
    return LO_ERR_WORKSPACE;
  }
  float *rhs_norm = sc, *rz = sc + ns, *alpha = sc + 2 * ns, *beta = sc + 3 * ns, *rn = sc + 4 * ns, *dots = sc + 5 * ns;
  float *prev_ar = sc + 8 * ns, *prev_beta = prev_ar + (size_t)B * (nt > 0 ? nt : 1);
  int *rhs_is_zero = flags, *has_conv = flags + ns;
  memset(info, 0, sizeof(*info));
  int rc = LO_OK;
  /* column normalisation :177-183 */
  col_dots(rhs, rhs, dots, B, N, c);
  for (size_t i = 0; i < ns; ++i) {
    float nrm = sqrtf(dots[i]);
    rhs_is_zero[i] = nrm < eps;
    rhs_norm[i] = rhs_is_zero[i] ? 1.0f : nrm;
  }
#pragma omp parallel for schedule(static) num_threads(nthr(B))
  for (int64_t b = 0; b < B; ++b)
    for (int64_t i = 0; i < N; ++i)
      for (int64_t j = 0; j < c; ++j) {
        const size_t e = ((size_t)b * N + i) * c + j;
        x[e] = x0 ? x0[e] / rhs_norm[b * c + j] : 0.f;
        p[e] = rhs[e] / rhs_norm[b * c + j]; /* (p holds the normalised rhs until the residual is formed) */
      }
  /* residual = rhs - A x0 :186 (one product is always spent, also for x0 = 0) */
  rc = cg_matvec(&cx, x, Ap);
  info->matvecs = 1;
  int anynan = 0;
  if (rc == LO_OK) {
    for (size_t e = 0; e < nv; ++e) {
      r[e] = p[e] - Ap[e];
      if (r[e] != r[e]) anynan = 1;
    }
    if (anynan) { /* :199-200 */
      info->nan_detected = 1;
      goto done;
    }
    col_dots(r, r, dots, B, N, c);
    int all_conv = 1;
    for (size_t i = 0; i < ns; ++i) {
      rn[i] = sqrtf(dots[i]); /* :204 */
      has_conv[i] = rn[i] < stop_after;
      all_conv &= has_conv[i];
    }
    int n_iter = prm->max_iter;
    if (all_conv && !nt) { /* :207-208 */
      n_iter = 0;
      info->skipped = 1;
    } else {
      rc = cg_precond(&cx, r, z); /* :213 */
      if (rc) goto done;
      memcpy(p, z, sizeof(float) * nv);
      col_dots(z, r, rz, B, N, c); /* :215 */
    }
    if (nt) memset(t_mat, 0, sizeof(float) * (size_t)nt * B * T * T);
    const int n_tri_iter = (int)(T < N ? T : N); /* :171 */
    int update_tridiag = 1, last_tridiag_iter = 0, tol_reached = 0, k = -1;
    for (k = 0; k < n_iter; ++k) { /* :245 */
      rc = cg_matvec(&cx, p, Ap); /* :248 */
      if (rc) goto done;
      info->matvecs += 1;
      col_dots(p, Ap, dots, B, N, c); /* :250-251 */
      for (size_t i = 0; i < ns; ++i) { /* :254-260 */
        const float a = dots[i];
        alpha[i] = (a < eps) ? 0.f : rz[i] / a;
        if (has_conv[i]) alpha[i] = 0.f;
      }
#pragma omp parallel for schedule(static) num_threads(nthr(B))
      for (int64_t b = 0; b < B; ++b)
        for (int64_t i = 0; i < N; ++i)
          for (int64_t j = 0; j < c; ++j) {
            const size_t e = ((size_t)b * N + i) * c + j;
            const float al = alpha[b * c + j];
            r[e] = r[e] - al * Ap[e]; /* :264 / :78 */
            x[e] = x[e] + al * p[e];  /* :31 */
          }
      rc = cg_precond(&cx, r, z); /* :268 */
      if (rc) goto done;
      col_dots(r, z, dots, B, N, c); /* :35-36 */
      for (size_t i = 0; i < ns; ++i) { /* :34, :39-42 */
        const float old = rz[i];
        rz[i] = dots[i];
        beta[i] = (old < eps) ? 0.f : rz[i] / old;
      }
#pragma omp parallel for schedule(static) num_threads(nthr(B))
      for (int64_t b = 0; b < B; ++b)
        for (int64_t i = 0; i < N; ++i)
          for (int64_t j = 0; j < c; ++j) {
            const size_t e = ((size_t)b * N + i) * c + j;
            p[e] = p[e] * beta[b * c + j] + z[e]; /* :46 */
          }
      col_dots(r, r, dots, B, N, c); /* :298 */
      float sum = 0.f;
      for (size_t i = 0; i < ns; ++i) {
        rn[i] = rhs_is_zero[i] ? 0.f : sqrtf(dots[i]); /* :299 */
        has_conv[i] = rn[i] < stop_after;              /* :300 */
        sum += rn[i];
      }
      const float mean = sum / (float)ns;
      info->mean_residual = mean;
      const int kfl = 10 < fmi - 1 ? 10 : fmi - 1;
      const int ktf = n_tri_iter < fmi - 1 ? n_tri_iter : fmi - 1;
      if (k >= kfl && mean < prm->tolerance && !(nt && k < ktf)) { /* :302-308 */
        tol_reached = 1;
        break;
      }
      if (nt && k < n_tri_iter && update_tridiag) { /* :311-332 */
        float maxoff = -INFINITY;
        for (int64_t b = 0; b < B; ++b)
          for (int j = 0; j < nt; ++j) {
            const float a = alpha[b * c + j];
            const float ar = 1.0f / ((a == 0.f) ? 1.0f : a); /* :314-317 */
            float* t = t_mat + ((size_t)j * B + b) * T * T;
            if (k == 0) {
              t[0] = ar; /* :320 */
            } else {
              const float pb = prev_beta[b * nt + j], par = prev_ar[b * nt + j];
              t[k * T + k] = ar + pb * par; /* :322 */
              const float off = sqrtf(pb) * par;
              t[k * T + k - 1] = off;
              t[(k - 1) * T + k] = off; /* :323-324 */
              if (off > maxoff) maxoff = off;
            }
            prev_ar[b * nt + j] = ar;
            prev_beta[b * nt + j] = beta[b * c + j]; /* :331-332 */
          }
        if (k > 0 && maxoff < 1e-6f) update_tridiag = 0; /* :326-327 */
        last_tridiag_iter = k;
      }
    }
    info->iterations = n_iter > 0 ? (k < n_iter ? k + 1 : n_iter) : 0;
    info->tolerance_reached = tol_reached;
    info->last_tridiag_iter = last_tridiag_iter;
    if (n_iter == 0) {
      float sum = 0.f;
      for (size_t i = 0; i < ns; ++i) sum += rn[i];
      info->mean_residual = sum / (float)ns;
    }
#pragma omp parallel for schedule(static) num_threads(nthr(B))
    for (int64_t b = 0; b < B; ++b)
      for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < c; ++j) {
          const size_t e = ((size_t)b * N + i) * c + j;
          x[e] = x[e] * rhs_norm[b * c + j]; /* :335 */
        }
  }
done:
  free(r); free(z); free(p); free(Ap); free(sc); free(flags);
  return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * lanczos_tridiag (utils/lanczos.py:9-164): Lanczos with full re-orthogonalisation, P probe columns per member.
 *   init_vecs [B, N, P] (the reference's randn default is not reproducible: the caller supplies them, :59-66)
 *   q_mat [max_iter, B, N, P], t_mat [max_iter, max_iter, B, P]: the reference's WORKING layouts (:69-77); the host
 *   wrapper crops to num_iter (:151) and permutes as :154-157.  Every batch-global decision (:133-147: the
 *   re-orthogonalisation loop runs while ANY inner product of ANY member exceeds tol, the iteration ends when NO
 *   beta of any member exceeds 1e-6) is taken from all members together, as the reference's torch.sum(...) does.
 * ------------------------------------------------------------------------------------------------------------------ */
static void lz_col_dots(const float* a, const float* bb, float* out, int64_t B, int64_t N, int64_t P) {
  col_dots(a, bb, out, B, N, P);
}

int lo_cpu_lanczos_tridiag_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const float* init_vecs,
                               int64_t P, int32_t max_iter, float tol, float* q_mat, float* t_mat, int32_t* num_iter_out) {
  if (!op || !init_vecs || !q_mat || !t_mat || !num_iter_out || P < 1 || max_iter < 1) return LO_ERR_BADARG;
  if (op->kind == LO_OP_CALLBACK ? !matvec : !op_ok(op)) return LO_ERR_BADARG;
  const int64_t B = op->B, N = op->N;
  const int num_iter = (int)(max_iter < N ? max_iter : N); /* :57 */
  const size_t nv = (size_t)B * N * P, ns = (size_t)B * P;
  cg_ctx cx = {op, matvec, matvec_user, NULL, NULL, NULL, B, N, P, 0};
  float* r = (float*)malloc(sizeof(float) * nv);
  float* dots = (float*)malloc(sizeof(float) * ns * (size_t)(num_iter + 1));
  float* sc = (float*)malloc(sizeof(float) * ns * 2);
  if (!r || !dots || !sc) {
    free(r); free(dots); free(sc);
    return LO_ERR_WORKSPACE;
  }
  float *alpha = sc, *beta = sc + ns;
  memset(q_mat, 0, sizeof(float) * (size_t)num_iter * nv);
  memset(t_mat, 0, sizeof(float) * (size_t)num_iter * num_iter * ns);
#define QK(k) (q_mat + (size_t)(k) * nv)
#define TM(i, j) (t_mat + ((size_t)(i) * num_iter + (j)) * ns)
  int rc = LO_OK;
  /* q_0 = init / ||init|| :81 */
  lz_col_dots(init_vecs, init_vecs, dots, B, N, P);
#pragma omp parallel for schedule(static) num_threads(nthr(B))
  for (int64_t b = 0; b < B; ++b)
    for (int64_t i = 0; i < N; ++i)
      for (int64_t j = 0; j < P; ++j) {
        const size_t e = ((size_t)b * N + i) * P + j;
        QK(0)[e] = init_vecs[e] / sqrtf(dots[b * P + j]);
      }
  rc = cg_matvec(&cx, QK(0), r); /* :85 */
  if (rc) goto done;
  lz_col_dots(QK(0), r, alpha, B, N, P); /* :88 */
#pragma omp parallel for schedule(static) num_threads(nthr(B))
  for (int64_t b = 0; b < B; ++b)
    for (int64_t i = 0; i < N; ++i)
      for (int64_t j = 0; j < P; ++j) {
        const size_t e = ((size_t)b * N + i) * P + j;
        r[e] = r[e] - alpha[b * P + j] * QK(0)[e]; /* :89 */
      }
  lz_col_dots(r, r, beta, B, N, P);
  for (size_t i = 0; i < ns; ++i) {
    beta[i] = sqrtf(beta[i]); /* :90 */
    TM(0, 0)[i] = alpha[i];   /* :93 */
    if (num_iter > 1) {
      TM(0, 1)[i] = beta[i];
      TM(1, 0)[i] = beta[i]; /* :94-95 */
    }
  }
  if (num_iter > 1) {
#pragma omp parallel for schedule(static) num_threads(nthr(B))
    for (int64_t b = 0; b < B; ++b)
      for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < P; ++j) {
          const size_t e = ((size_t)b * N + i) * P + j;
          QK(1)[e] = r[e] / beta[b * P + j]; /* :98 */
        }
  }
  int k = 0;
  for (k = 1; k < num_iter; ++k) { /* :101 */
    const float* q_prev = QK(k - 1);
    const float* q_curr = QK(k);
    const float* beta_prev = TM(k, k - 1);
    rc = cg_matvec(&cx, q_curr, r);
    if (rc) goto done;
#pragma omp parallel for schedule(static) num_threads(nthr(B))
    for (int64_t b = 0; b < B; ++b)
      for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < P; ++j) {
          const size_t e = ((size_t)b * N + i) * P + j;
          r[e] = r[e] - q_prev[e] * beta_prev[b * P + j]; /* :108 */
        }
    lz_col_dots(q_curr, r, alpha, B, N, P); /* :109 */
    for (size_t i = 0; i < ns; ++i) TM(k, k)[i] = alpha[i]; /* :111 */
    if (k + 1 < num_iter) { /* :114 */
#pragma omp parallel for schedule(static) num_threads(nthr(B))
      for (int64_t b = 0; b < B; ++b)
        for (int64_t i = 0; i < N; ++i)
          for (int64_t j = 0; j < P; ++j) {
            const size_t e = ((size_t)b * N + i) * P + j;
            r[e] = r[e] - alpha[b * P + j] * q_curr[e]; /* :115 */
          }
      int could = 0;
      for (int pass = 0; pass <= 10; ++pass) { /* pass 0 = :117-124, passes 1..10 = the loop :133-142 */
        /* correction = sum_m Q_m (Q_m^T r) over the k + 1 vectors so far (:118-119), subtracted (:120) */
        for (int m = 0; m <= k; ++m) lz_col_dots(r, QK(m), dots + (size_t)m * ns, B, N, P);
#pragma omp parallel for schedule(static) num_threads(nthr(B))
        for (int64_t b = 0; b < B; ++b)
          for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < P; ++j) {
              const size_t e = ((size_t)b * N + i) * P + j;
              float corr = 0.f;
              for (int m = 0; m <= k; ++m) corr += QK(m)[e] * dots[(size_t)m * ns + b * P + j];
              r[e] = r[e] - corr;
            }
        lz_col_dots(r, r, beta, B, N, P);
#pragma omp parallel for schedule(static) num_threads(nthr(B))
        for (int64_t b = 0; b < B; ++b)
          for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < P; ++j) {
              const size_t e = ((size_t)b * N + i) * P + j;
              r[e] = r[e] / sqrtf(beta[b * P + j]); /* :121-122 / :138-139 */
            }
        if (pass == 0) { /* :125-128: beta of THIS step is the norm before any extra pass */
          for (size_t i = 0; i < ns; ++i) {
            const float bv = sqrtf(beta[i]);
            TM(k, k + 1)[i] = bv;
            TM(k + 1, k)[i] = bv;
            sc[ns + i] = bv; /* (kept for the test below) */
          }
        }
        if (pass == 10) break; /* the tenth extra pass is not checked again (:133-142): could_reorthogonalize stays False */
        /* inner products with all previous vectors (:131 / :140): signed compare against tol */
        int over = 0;
        for (int m = 0; m <= k; ++m) {
          lz_col_dots(QK(m), r, dots + (size_t)m * ns, B, N, P);
          for (size_t i = 0; i < ns; ++i) over += dots[(size_t)m * ns + i] > tol;
        }
        if (!over) { /* :134-136 */
          could = 1;
          break;
        }
      }
      memcpy(QK(k + 1), r, sizeof(float) * nv); /* :145 */
      int big = 0;
      for (size_t i = 0; i < ns; ++i) big += fabsf(sc[ns + i]) > 1e-6f;
      if (big == 0 || !could) break; /* :147 */
    }
  }
  *num_iter_out = (k < num_iter ? k : num_iter - 1) + 1; /* :151 */
done:
#undef QK
#undef TM
  free(r); free(dots); free(sc);
  return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * lanczos_tridiag_to_diag (utils/lanczos.py:167-189) + StochasticLQ.to_dense with funcs = [log]
 * (utils/stochastic_lq.py:45-82): eigendecomposition of every k x k tridiagonal, negative eigenvalues clamped (their
 * eigenvector columns zeroed, the eigenvalue set to 1: :186-187), logdet[b] = (n / P) sum_p sum_i V_p[0, i]^2 log(lambda_p,i).
 * The reference calls LAPACK's symmetric eigensolver in the input precision; this restatement runs the implicit QL
 * iteration with Wilkinson shifts (EISPACK tql2) in DOUBLE on the fp32 entries -- the eigenvalues of the same matrix
 * without the fp32 solver's own rounding.  Only the FIRST ROW of the eigenvector matrix is needed and accumulated.
 *   t_mat [P, B, ld, ld] (k x k leading blocks used), evals_out [P, B, k] or NULL, first_out [P, B, k] or NULL.
 * ------------------------------------------------------------------------------------------------------------------ */
static int tridiag_ql(int n, double* d, double* e, double* z0) {
  /* d[0..n): diagonal, e[0..n): e[i] = T[i][i-1] (e[0] unused); z0: first row of the accumulated rotations (starts e_1) */
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    const double t = fabs(d[l]) + fabs(e[l]);
    if (t > tst1) tst1 = t;
    int m = l;
    while (m < n) {
      if (fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 60) return 1;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          h = z0[i + 1];
          z0[i + 1] = s * z0[i] + c * h;
          z0[i] = c * z0[i] - s * h;
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (fabs(e[l]) > eps * tst1);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  return 0;
}

int lo_cpu_tridiag_eigh_slq_f32(const float* t_mat, int64_t P, int64_t B, int32_t k, int32_t ld, int64_t n, float* logdet,
                                double* evals_out, double* first_out) {
  if (!t_mat || !logdet || P < 1 || B < 1 || k < 1 || k > ld || k > 512) return LO_ERR_BADARG;
  int bad = 0;
#pragma omp parallel for schedule(static) num_threads(nthr(B)) reduction(+ : bad)
  for (int64_t b = 0; b < B; ++b) {
    double d[512], e[512], z0[512];
    double acc = 0.0;
    for (int64_t p = 0; p < P; ++p) {
      const float* T = t_mat + ((size_t)p * B + b) * ld * ld;
      for (int i = 0; i < k; ++i) {
        d[i] = (double)T[(size_t)i * ld + i];
        e[i] = i ? (double)T[(size_t)i * ld + i - 1] : 0.0;
        z0[i] = i ? 0.0 : 1.0;
      }
      bad += tridiag_ql(k, d, e, z0);
      double dots = 0.0;
      for (int i = 0; i < k; ++i) {
        double ev = d[i], v0 = z0[i];
        if (!(ev >= 0.0)) { /* mask = evals.ge(0): zero the eigenvector COLUMN, eigenvalue := 1 (:185-187) */
          v0 = 0.0;
          ev = 1.0;
        }
        if (evals_out) evals_out[((size_t)p * B + b) * k + i] = ev;
        if (first_out) first_out[((size_t)p * B + b) * k + i] = v0;
        dots += v0 * v0 * log(ev); /* stochastic_lq.py:72-78 */
      }
      acc += ((double)n / (double)P) * dots; /* :80 */
    }
    logdet[b] = (float)acc;
  }
  return bad ? LO_ERR_LAUNCH : LO_OK;
}
