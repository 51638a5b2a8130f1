// lo_minres.hip -- shifted MINRES: (value * K + shift_q I) x_q = rhs for all shifts q at once
// (reference: linear_operator/utils/minres.py:10-207, update block :210-282; SURVEY 8(f) rank 4 -- the second
// iterative solver of the reference, used by contour_integral_quad / sqrt_inv_matmul).
//
// One preconditioned Lanczos recurrence (alpha, beta, z, q) is shared by the shifts; each shift carries its own QR
// (Givens) recurrence and pair of search vectors.  Per iteration:
//   matvec (operator plan of lo_matvec.hip, partial sums of q . Kq fused where the plan supports it)
//   k_mr_lanczos   z_c = value K q - alpha z_1 - beta_prev z_2 (in place on the product, :144), partials of z_c . z_c
//   [preconditioner: Woodbury kernels of lo_skinny.hip or the caller's closure, partials of z_c . q_c]
//   k_mr_givens    beta (:147-150) and the per-(shift, member, column) rotation coefficients (:236-262)
//   k_mr_update    z_c, q_c /= beta; per shift: search vector (:265-267), solution += search * scale (:270-271),
//                  every 10th iteration the partial norms of the update and of the solution (:178-180)
// All streaming, HBM-bound: (3 + 4 Q) N c floats per iteration and member next to the operator's matvec.
// The stop test (:178-183) is evaluated on the device and polled by the host every 10th iteration.
#include <algorithm>
#include <math.h>
#include <string.h>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

struct MrCtrl {
  float conv;
  int stop;
};

struct MrDev {
  int64_t B, N;
  int c, Q, S, S_dot, S_b;
  float value, eps, tol;
  int shifts_per_member;
  const float* shifts;  // [Q] or [Q, B]
  float *z[3], *q[2];   // Lanczos vectors (q aliases z without a preconditioner)
  float *sA, *sB;       // search vectors [Q,B,N,c]
  float* sol;           // [Q,B,N,c] (the caller's output buffer)
  float *rhs_norm, *alpha, *beta_prev, *inv_beta;  // [B,c]
  int* rhs_zero;                                    // [B,c]
  float *cos1, *sin1, *cos2, *sin2, *scale_prev;    // [Q,B,c]
  float *sub, *subsub, *diag, *scale_upd;           // [Q,B,c] coefficients of the current iteration
  float *dot_part, *b_part;                         // [B,S_dot,c], [B,S_b,c]
  float *upd_part, *sol_part;                       // [Q,B,S,c]
  MrCtrl* ctrl;
};

// rhs_norm, rhs_is_zero (:52-55) from the partials of rhs . rhs
__global__ __launch_bounds__(kThreads) void k_mr_norm(MrDev d, const float* __restrict__ part) {
  const int64_t n = d.B * d.c;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / d.c;
    const int col = (int)(i % d.c);
    float acc = 0.f;
    for (int s = 0; s < d.S; ++s) acc += part[(b * d.S + s) * d.c + col];
    const float nrm = sqrtf(acc);
    const int zero = nrm < 1e-10f;
    d.rhs_zero[i] = zero;
    d.rhs_norm[i] = zero ? 1.0f : nrm;
  }
}

// z_1 = rhs / rhs_norm (:55, :75)
__global__ __launch_bounds__(kThreads) void k_mr_init_vec(MrDev d, const float* __restrict__ rhs, float* __restrict__ z1,
                                                           int rows_per) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int c = d.c, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float nrm = d.rhs_norm[(size_t)b * c + col];
  const size_t base = (size_t)b * N * c + col;
  for (int row = r0 + slot; row < r1; row += nrs) z1[base + (size_t)row * c] = rhs[base + (size_t)row * c] / nrm;
}

// beta_prev = sqrt(z_1 . q_1) (:80, not clamped), scale_prev = beta_prev (:113), rotations = identity (:87-91)
__global__ __launch_bounds__(kThreads) void k_mr_init_scal(MrDev d) {
  const int64_t n = d.B * d.c;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / d.c;
    const int col = (int)(i % d.c);
    float acc = 0.f;
    for (int s = 0; s < d.S_b; ++s) acc += d.b_part[(b * d.S_b + s) * d.c + col];
    const float bp = sqrtf(acc);
    d.beta_prev[i] = bp;
    d.inv_beta[i] = 1.0f / bp;  // (0 -> inf: z_1 / beta_prev is 0 / 0 = NaN for an all-zero column, as in the reference)
    for (int q = 0; q < d.Q; ++q) {
      const size_t o = (size_t)q * n + i;
      d.cos1[o] = 1.f; d.cos2[o] = 1.f; d.sin1[o] = 0.f; d.sin2[o] = 0.f;
      d.scale_prev[o] = bp;
    }
  }
}

// v /= beta (element-wise; :83-84)
__global__ __launch_bounds__(kThreads) void k_mr_div(MrDev d, float* __restrict__ v, float* __restrict__ w, int rows_per) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int c = d.c, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float bp = d.beta_prev[(size_t)b * c + col];
  const size_t base = (size_t)b * N * c + col;
  for (int row = r0 + slot; row < r1; row += nrs) {
    v[base + (size_t)row * c] = v[base + (size_t)row * c] / bp;
    if (w) w[base + (size_t)row * c] = w[base + (size_t)row * c] / bp;
  }
}

// alpha = value * sum(prod o q_1) (:141-142); z_c = value * prod - alpha z_1 - beta_prev z_2, in place (:144);
// without a preconditioner also the partials of z_c . z_c
__global__ __launch_bounds__(kThreads) void k_mr_lanczos(MrDev d, float* __restrict__ zc, const float* __restrict__ q1,
                                                          const float* __restrict__ z1, const float* __restrict__ z2,
                                                          int want_bpart, int rows_per) {
  __shared__ float red[kThreads];
  __shared__ float alpha_s[kMaxCols];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int c = d.c, N = (int)d.N;
  if (threadIdx.x < c) {
    const int col = threadIdx.x;
    float a = 0.f;
    for (int ss = 0; ss < d.S_dot; ++ss) a += d.dot_part[((size_t)b * d.S_dot + ss) * c + col];
    a *= d.value;
    alpha_s[col] = a;
    if (s == 0) d.alpha[(size_t)b * c + col] = a;
  }
  __syncthreads();
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  float acc = 0.f;
  if (slot < nrs) {
    const float a = alpha_s[col];
    const float bp = d.beta_prev[(size_t)b * c + col];
    const size_t base = (size_t)b * N * c + col;
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      float v = d.value * zc[i];
      v = fmaf(-a, z1[i], v);    // addcmul_(alpha, z_1, value=-1)
      v = fmaf(-bp, z2[i], v);   // addcmul_(beta_prev, z_2, value=-1)
      zc[i] = v;
      acc = fmaf(v, v, acc);
    }
  }
  if (want_bpart) {
    const float tot = block_colsum(slot < nrs ? acc : 0.f, c, nrs, red);
    if (threadIdx.x < c) d.b_part[((size_t)b * S + s) * c + col] = tot;
  }
}

// beta_curr (:147-150) and the Givens recurrences of every shift (:236-262); rotates the scalar state (:186-198)
__global__ __launch_bounds__(kThreads) void k_mr_givens(MrDev d) {
  const int64_t n = d.B * d.c;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / d.c;
    const int col = (int)(i % d.c);
    float acc = 0.f;
    for (int s = 0; s < d.S_b; ++s) acc += d.b_part[(b * d.S_b + s) * d.c + col];
    const float beta = fmaxf(sqrtf(acc), d.eps);  // sqrt_().clamp_min_(eps)   (NaN propagates like clamp_min)
    const float beta_c = (acc != acc) ? acc : beta;
    const float bp = d.beta_prev[i];
    const float alpha = d.alpha[i];
    d.inv_beta[i] = 1.0f / beta_c;
    for (int q = 0; q < d.Q; ++q) {
      const size_t o = (size_t)q * n + i;
      const float shift = d.shifts_per_member ? d.shifts[(size_t)q * d.B + b] : d.shifts[q];
      const float c2 = d.cos2[o], s2 = d.sin2[o], c1 = d.cos1[o], s1 = d.sin1[o];
      const float subsub = s2 * bp;                       // :238
      float sub = c2 * bp;                                // :239
      const float as = alpha + shift;                     // :242
      float diag = fmaf(-s1, sub, as * c1);               // :245
      sub = fmaf(s1, as, sub * c1);                       // :246
      const float radius = sqrtf(fmaf(beta_c, beta_c, diag * diag));  // :249
      const float cc = diag / radius;                     // :250
      const float sc = beta_c / radius;                   // :251
      diag = fmaf(sc, beta_c, diag * cc);                 // :253
      const float sp = d.scale_prev[o];
      d.sub[o] = sub;
      d.subsub[o] = subsub;
      d.diag[o] = diag;
      d.scale_upd[o] = sp * cc;                           // :260 (scale_prev.mul_(cos_curr))
      d.scale_prev[o] = -(sp * sc);                       // :259, then scale_prev <- scale_curr (:198)
      d.cos2[o] = c1; d.sin2[o] = s1; d.cos1[o] = cc; d.sin1[o] = sc;  // :192-193
    }
    d.beta_prev[i] = beta_c;                              // :190
  }
}

// normalise the new Lanczos vectors (:152-153) and update every shift's search vector and solution (:263-271)
__global__ __launch_bounds__(kThreads) void k_mr_update(MrDev d, float* __restrict__ zc, float* __restrict__ qc,
                                                         const float* __restrict__ q1, const float* __restrict__ sA,
                                                         float* __restrict__ sB, int check, int rows_per) {
  __shared__ float red[kThreads];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int c = d.c, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const bool act = slot < nrs;
  const size_t bc = (size_t)b * c + col;
  const size_t base = (size_t)b * N * c + col;
  const size_t nbc = (size_t)d.B * c, nvec = (size_t)d.B * N * c;
  if (act) {
    const float ib = d.inv_beta[bc];
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      zc[i] = zc[i] * ib;
      if (qc != zc) qc[i] = qc[i] * ib;
    }
  }
  for (int q = 0; q < d.Q; ++q) {
    float au = 0.f, as = 0.f;
    if (act) {
      const size_t o = (size_t)q * nbc + bc;
      const float sub = d.sub[o], subsub = d.subsub[o], diag = d.diag[o], su = d.scale_upd[o];
      const float* a1 = sA + (size_t)q * nvec;
      float* a2 = sB + (size_t)q * nvec;
      float* so = d.sol + (size_t)q * nvec;
      for (int row = r0 + slot; row < r1; row += nrs) {
        const size_t i = base + (size_t)row * c;
        float v = fmaf(-sub, a1[i], q1[i]);      // addcmul(q_prev1, sub_diag, search_prev1, value=-1)
        v = fmaf(-subsub, a2[i], v);             // addcmul_(subsub_diag, search_prev2, value=-1)
        v = v / diag;
        a2[i] = v;                               // search_curr takes the place of search_prev2 (:194-197)
        const float u = v * su;
        const float sn = so[i] + u;
        so[i] = sn;
        au = fmaf(u, u, au);
        as = fmaf(sn, sn, as);
      }
    }
    if (check) {
      const float tu = block_colsum(act ? au : 0.f, c, nrs, red);
      if (threadIdx.x < c) d.upd_part[(((size_t)q * d.B + b) * S + s) * c + col] = tu;
      const float ts = block_colsum(act ? as : 0.f, c, nrs, red);
      if (threadIdx.x < c) d.sol_part[(((size_t)q * d.B + b) * S + s) * c + col] = ts;
    }
  }
}

// conv = mean(||update|| / ||solution||) over shifts, members, columns (:178-181); 0/0 = NaN never stops the loop
__global__ __launch_bounds__(kThreads) void k_mr_conv(MrDev d) {
  __shared__ float red[kThreads];
  const int64_t n = (int64_t)d.Q * d.B * d.c;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const int64_t qb = i / d.c;
    const int col = (int)(i % d.c);
    float u = 0.f, sn = 0.f;
    for (int s = 0; s < d.S; ++s) {
      u += d.upd_part[(qb * d.S + s) * d.c + col];
      sn += d.sol_part[(qb * d.S + s) * d.c + col];
    }
    acc += sqrtf(u) / sqrtf(sn);
  }
  const float tot = block_sum256(acc, red);
  if (threadIdx.x == 0) {
    const float conv = tot / (float)n;
    d.ctrl->conv = conv;
    d.ctrl->stop = (conv < d.tol) ? 1 : 0;
  }
}

// solution.masked_fill_(rhs_is_zero, 0) * rhs_norm (:201, :212)
__global__ __launch_bounds__(kThreads) void k_mr_final(MrDev d, int rows_per) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int c = d.c, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float nrm = d.rhs_norm[(size_t)b * c + col];
  const int zero = d.rhs_zero[(size_t)b * c + col];
  const size_t base = (size_t)b * N * c + col, nvec = (size_t)d.B * N * c;
  for (int q = 0; q < d.Q; ++q)
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = (size_t)q * nvec + base + (size_t)row * c;
      d.sol[i] = zero ? 0.f : d.sol[i] * nrm;
    }
}

static int mr_padded_rank_k(int k) {
  int rq = (k + 3) / 4, p = 1;
  while (p < rq) p <<= 1;
  return 4 * p;
}

struct MrHost {
  Split sp;
  int preR4;
  const float* Qp;
  float* upart;
};

static size_t mr_layout(const lo_op_desc* op, const lo_precond_desc* pre, bool pre_cb, const lo_minres_params* prm,
                        void* ws, size_t ws_bytes, MrDev* dout, MatvecPlan* pl, lo_matvec_cb cb, void* cb_user,
                        MrHost* h, hipStream_t st, int* rc_out, bool init) {
  const int64_t B = op->B, N = op->N, c = prm->c;
  const int Q = prm->n_shifts;
  Split sp = choose_split(B, N, 256);
  Arena ar(ws, ws_bytes);
  const size_t nv = (size_t)B * N * c, ns = (size_t)B * c;
  const bool precond = pre != nullptr || pre_cb;
  MrDev d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.N = N; d.c = (int)c; d.Q = Q; d.S = sp.S;
  d.ctrl = ar.take<MrCtrl>(1);
  for (int i = 0; i < 3; ++i) d.z[i] = ar.take<float>(nv);
  for (int i = 0; i < 2; ++i) d.q[i] = precond ? ar.take<float>(nv) : nullptr;
  d.sA = ar.take<float>(nv * Q);
  d.sB = ar.take<float>(nv * Q);
  int S_dot = sp.S;
  if (op->kind == LO_OP_DENSE_DIAG) S_dot = dense_S_dot(B, N, c);
  else if (op->kind == LO_OP_KRON_DIAG) S_dot = kron_S_dot((int)op->R, (int)op->n2, c, sp.S);
  d.S_dot = S_dot;
  d.S_b = sp.S;
  d.dot_part = ar.take<float>((size_t)B * std::max(S_dot, sp.S) * c);
  d.b_part = ar.take<float>((size_t)B * sp.S * c);
  d.upd_part = ar.take<float>((size_t)Q * B * sp.S * c);
  d.sol_part = ar.take<float>((size_t)Q * B * sp.S * c);
  d.rhs_norm = ar.take<float>(ns);
  d.alpha = ar.take<float>(ns);
  d.beta_prev = ar.take<float>(ns);
  d.inv_beta = ar.take<float>(ns);
  d.rhs_zero = ar.take<int>(ns);
  float** per_q[] = {&d.cos1, &d.sin1, &d.cos2, &d.sin2, &d.scale_prev, &d.sub, &d.subsub, &d.diag, &d.scale_upd};
  for (float** p : per_q) *p = ar.take<float>(ns * Q);
  if (h) {
    h->sp = sp;
    h->preR4 = 0;
    h->Qp = nullptr;
    h->upart = nullptr;
  }
  if (pre) {
    const int R4 = mr_padded_rank_k(pre->k);
    float* up = ar.take<float>((size_t)B * sp.S * R4 * c);
    const float* qp = pre->Q;
    if (pre->ldq != R4) {
      float* pad = ar.take<float>((size_t)B * N * R4);
      if (init && ar.ok && ws) {
        if (pre->ldq != pre->k) { if (rc_out) *rc_out = LO_ERR_BADARG; }
        else {
          int rc = pad_rows(pre->Q, pre->k, pad, R4, B * N, st);
          if (rc && rc_out) *rc_out = rc;
        }
      }
      qp = pad;
    }
    if (h) { h->preR4 = R4; h->Qp = qp; h->upart = up; }
  }
  if (init) {
    int rc = matvec_plan_init(pl, op, cb, cb_user, c, sp, &ar, st);
    if (rc && rc_out && *rc_out == LO_OK) *rc_out = rc;
  } else {
    ar.off += matvec_plan_bytes(op, c, sp);
  }
  if (dout) *dout = d;
  if (init && !ar.ok && rc_out && *rc_out == LO_OK) *rc_out = LO_ERR_WORKSPACE;
  return ar.off + 1024;
}

}  // namespace lo

using namespace lo;

extern "C" {

size_t lo_minres_workspace_bytes(const lo_op_desc* op, const lo_precond_desc* pre, const lo_minres_params* prm) {
  if (!op || !prm || prm->n_shifts < 1 || prm->c < 1) return 0;
  lo_precond_desc dummy;
  const lo_precond_desc* p = pre;
  if (!p) {  // worst case: a closure preconditioner needs the q vectors as well
    dummy.k = 4; dummy.ldq = 4; dummy.constant_diag = 0; dummy.reserved = 0; dummy.Q = nullptr; dummy.dinv = nullptr;
    p = &dummy;
  }
  return mr_layout(op, p, true, prm, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false);
}

int lo_minres_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const lo_precond_desc* pre,
                  lo_matvec_cb precond_cb, void* precond_user, const lo_minres_params* prm, const float* rhs,
                  const float* shifts, float* x, void* ws, size_t ws_bytes, lo_minres_info* info, void* stream) {
  if (!op || !prm || !rhs || !shifts || !x || !ws || !info) return LO_ERR_BADARG;
  if (prm->c < 1 || prm->c > kMaxCols || prm->n_shifts < 1 || prm->max_iter < 0) return LO_ERR_UNSUPPORTED;
  if (pre && precond_cb) return LO_ERR_BADARG;
  if (pre && (pre->k < 1 || pre->k > kMaxRank || !pre->Q || !pre->dinv)) return LO_ERR_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = op->B, N = op->N;
  const int c = (int)prm->c, Q = prm->n_shifts;
  MrDev d;
  MatvecPlan pl;
  PlanGuard pl_guard(&pl);
  MrHost h;
  int rc = LO_OK;
  mr_layout(op, pre, precond_cb != nullptr, prm, ws, ws_bytes, &d, &pl, matvec, matvec_user, &h, st, &rc, true);
  if (rc) return rc;
  const Split sp = h.sp;
  const bool precond = pre != nullptr || precond_cb != nullptr;
  d.value = prm->has_value ? prm->value : 1.0f;
  d.eps = prm->eps;
  d.tol = prm->tolerance;
  d.shifts = shifts;
  d.shifts_per_member = prm->shifts_per_member;
  d.sol = x;
  d.S_dot = pl.S_dot;
  dim3 gridv(sp.S, (unsigned)B), block(kThreads);
  const unsigned gscal = (unsigned)std::min<int64_t>(256, ((int64_t)B * c + kThreads - 1) / kThreads);
  const size_t nv = (size_t)B * N * c;

  auto apply_precond = [&](const float* r, float* z, float* dotp) -> int {
    if (pre) {
      int e = skinny_tn(h.Qp, h.preR4, h.preR4, r, c, h.upart, B, N, sp, nullptr, st);
      if (e) return e;
      return skinny_nn(h.Qp, h.preR4, h.preR4, h.upart, pre->dinv, pre->constant_diag ? LO_DIAG_CONST : LO_DIAG_FULL,
                       -1.0f, r, c, z, dotp, B, N, sp, nullptr, st);
    }
    int e = precond_cb(precond_user, r, z, B, N, c, (void*)st);
    if (e) return LO_ERR_LAUNCH;
    return vec_dot_part(r, z, c, dotp, B, N, sp, nullptr, st);
  };

  LO_HIP_CHECK(hipMemsetAsync(d.ctrl, 0, sizeof(MrCtrl), st));
  LO_HIP_CHECK(hipMemsetAsync(x, 0, sizeof(float) * nv * Q, st));
  LO_HIP_CHECK(hipMemsetAsync(d.sA, 0, sizeof(float) * nv * Q, st));
  LO_HIP_CHECK(hipMemsetAsync(d.sB, 0, sizeof(float) * nv * Q, st));
  // ---- initialisation (:52-113) ----
  float *z2 = d.z[0], *z1 = d.z[1], *zc = d.z[2];
  float *q1 = precond ? d.q[0] : z1, *qc = precond ? d.q[1] : zc;
  LO_HIP_CHECK(hipMemsetAsync(z2, 0, sizeof(float) * nv, st));
  rc = vec_dot_part(rhs, rhs, c, d.b_part, B, N, sp, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_mr_norm, dim3(gscal), block, 0, st, d, d.b_part);
  hipLaunchKernelGGL(k_mr_init_vec, gridv, block, 0, st, d, rhs, z1, sp.rows);
  LO_LAUNCH_CHECK();
  if (precond) rc = apply_precond(z1, q1, d.b_part);
  else rc = vec_dot_part(z1, z1, c, d.b_part, B, N, sp, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_mr_init_scal, dim3(gscal), block, 0, st, d);
  hipLaunchKernelGGL(k_mr_div, gridv, block, 0, st, d, z1, precond ? q1 : nullptr, sp.rows);
  LO_LAUNCH_CHECK();

  // ---- iterations (:134-198) ----
  const int n_loop = prm->max_iter + 2;
  MrCtrl hc;
  memset(&hc, 0, sizeof(hc));
  int it = 0, matvecs = 0;
  float *sA = d.sA, *sB = d.sB;
  for (int i = 0; i < n_loop; ++i) {
    rc = matvec_run(&pl, q1, zc, d.dot_part, nullptr, st);  // prod = K q_1, partials of q_1 . prod
    if (rc) return rc;
    ++matvecs;
    LO_PROF_BEGIN("mr_lanczos", st);
    hipLaunchKernelGGL(k_mr_lanczos, gridv, block, 0, st, d, zc, q1, z1, z2, precond ? 0 : 1, sp.rows);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
    if (precond) {
      rc = apply_precond(zc, qc, d.b_part);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_mr_givens, dim3(gscal), block, 0, st, d);
    const int check = ((i + 1) % 10 == 0) ? 1 : 0;
    LO_PROF_BEGIN("mr_update", st);
    hipLaunchKernelGGL(k_mr_update, gridv, block, 0, st, d, zc, qc, q1, sA, sB, check, sp.rows);
    LO_PROF_END(st);
    LO_LAUNCH_CHECK();
    it = i + 1;
    if (check) {
      hipLaunchKernelGGL(k_mr_conv, dim3(1), block, 0, st, d);
      LO_LAUNCH_CHECK();
      LO_HIP_CHECK(hipMemcpyAsync(&hc, d.ctrl, sizeof(MrCtrl), hipMemcpyDeviceToHost, st));
      LO_HIP_CHECK(hipStreamSynchronize(st));
      if (hc.stop) break;
    }
    // rotate (:186-198): z_2 <- z_1 <- z_c; q_1 <- q_c; search_prev2 <- search_prev1 <- search_curr
    float* t = z2; z2 = z1; z1 = zc; zc = t;
    if (precond) { t = q1; q1 = qc; qc = t; }
    else { q1 = z1; qc = zc; }
    t = sA; sA = sB; sB = t;
  }
  hipLaunchKernelGGL(k_mr_final, gridv, block, 0, st, d, sp.rows);
  LO_LAUNCH_CHECK();
  LO_HIP_CHECK(hipStreamSynchronize(st));
  info->iterations = it;
  info->matvecs = matvecs;
  info->converged = hc.stop;
  info->conv = hc.conv;
  return LO_OK;
}

}  // extern "C"
