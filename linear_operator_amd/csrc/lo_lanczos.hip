// lo_lanczos.hip -- batched Lanczos tridiagonalisation with full re-orthogonalisation, restating
// linear_operator/utils/lanczos.py:9-164 (reference).  Vectors are [B, N, P] (probe index innermost),
// q_mat is kept in the reference's storage order [k, B, N, P] (:69-76) and t_mat as [k, k, B, P] (:77).
// Per step k (lanczos.py:101-148):
//   r = A q_k - beta_{k-1} q_{k-1} ; alpha_k = q_k . r ; r -= alpha_k q_k
//   full re-orthogonalisation  r -= Q_{<=k} (Q_{<=k}^T r)   (classical Gram-Schmidt: all k+1 inner products
//   use the same r, :118-120) ; normalise ; up to 10 extra passes while any SIGNED inner product > tol
//   (:131-142) ; store q_{k+1} ; stop if all |beta| <= 1e-6 or re-orthogonalisation failed (:147).
// The two data-dependent decisions are batch-global; a single-workgroup control kernel takes them on the
// device and the host reads one word per decision (the reference does the same through bool(tensor)).
#include <algorithm>
#include <cstring>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

struct LzCtrl {
  int need_reorth;  // sum(inner_products > tol) != 0
  int all_small;    // sum(|beta| > 1e-6) == 0
};

struct LzDev {
  int64_t B, N;
  int P, S, rows, max_iter;
  float tol;
  float* q;     // [max_iter, B, N, P]
  float* t;     // [max_iter, max_iter, B, P]
  float* r;     // [B, N, P]
  float* part;  // [B, S, (max_iter+1), P]
  float* coef;  // [B, (max_iter+1), P]
  float* scal;  // [B, P]
  LzCtrl* ctrl;
};

__device__ __forceinline__ size_t qoff(const LzDev& d, int k) { return (size_t)k * d.B * d.N * d.P; }
__device__ __forceinline__ float* tptr(const LzDev& d, int i, int j) {
  return d.t + ((size_t)i * d.max_iter + j) * d.B * d.P;
}

// part[b,s,0,p] = sum_rows a o b
__global__ __launch_bounds__(kThreads) void k_lz_dot(LzDev d, const float* __restrict__ a,
                                                      const float* __restrict__ bv) {
  __shared__ float red[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  float acc = 0.f;
  if (slot < nrs) {
    const size_t base = (size_t)b * N * c + col;
    for (int row = r0 + slot; row < r1; row += nrs) acc = fmaf(a[base + (size_t)row * c], bv[base + (size_t)row * c], acc);
  }
  const float tot = block_colsum(slot < nrs ? acc : 0.f, c, nrs, red);
  if (threadIdx.x < c) d.part[(((size_t)b * d.S + s) * (d.max_iter + 1)) * c + col] = tot;
}

// part[b,s,j,p] = sum_rows r o q_j  for j = 0..nq-1
__global__ __launch_bounds__(kThreads) void k_lz_multidot(LzDev d, int nq) {
  __shared__ float red[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + col;
  for (int j = 0; j < nq; ++j) {
    const float* qj = d.q + qoff(d, j);
    float acc = 0.f;
    if (slot < nrs)
      for (int row = r0 + slot; row < r1; row += nrs)
        acc = fmaf(d.r[base + (size_t)row * c], qj[base + (size_t)row * c], acc);
    const float tot = block_colsum(slot < nrs ? acc : 0.f, c, nrs, red);
    if (threadIdx.x < c) d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * c + col] = tot;
  }
}

// coef[b,j,p] = sum_s part[b,s,j,p]; mode 1: also flag any coef > tol (signed, :134)
__global__ __launch_bounds__(kThreads) void k_lz_reduce(LzDev d, int nq, int check) {
  __shared__ float red[kThreads];
  const int64_t n = d.B * nq * d.P;
  float flag = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const int p = (int)(i % d.P);
    const int j = (int)((i / d.P) % nq);
    const int64_t b = i / ((int64_t)d.P * nq);
    float acc = 0.f;
    for (int s = 0; s < d.S; ++s) acc += d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * d.P + p];
    d.coef[((size_t)b * (d.max_iter + 1) + j) * d.P + p] = acc;
    if (check && acc > d.tol) flag = 1.f;
  }
  if (check) {
    const float any = block_sum256(flag, red);
    if (threadIdx.x == 0) d.ctrl->need_reorth = any > 0.f ? 1 : 0;
  }
}

// r -= sum_j coef[b,j,p] * q_j   (:119-120)
__global__ __launch_bounds__(kThreads) void k_lz_correct(LzDev d, int nq) {
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + col;
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    float corr = 0.f;
    for (int j = 0; j < nq; ++j)
      corr = fmaf(d.q[qoff(d, j) + i], d.coef[((size_t)b * (d.max_iter + 1) + j) * c + col], corr);
    d.r[i] -= corr;
  }
}

// elementwise helpers on [B,N,P] with per-(b,p) scalars
// mode 0: out = a / sqrt(scal)                      (normalise)
// mode 1: r = r - scal * a                          (r.sub_(alpha * q))
__global__ __launch_bounds__(kThreads) void k_lz_axpy(LzDev d, int mode, const float* __restrict__ a,
                                                       float* __restrict__ out, const float* __restrict__ scal) {
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + col;
  const float sc = scal[(size_t)b * c + col];
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    if (mode == 0) out[i] = a[i] / sc;
    else out[i] = out[i] - sc * a[i];
  }
}

// scalar stage after a dot: what = 0 -> scal = sqrt(sum) (norm);   what = 1 -> scal = sum (alpha)
// optionally stores into t_mat[i,j] (and its transpose position), and checks |beta| > 1e-6
__global__ __launch_bounds__(kThreads) void k_lz_scal(LzDev d, int what, int ti, int tj, int sym, int check_small) {
  __shared__ float red[kThreads];
  const int64_t n = d.B * d.P;
  float big = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const int64_t b = i / d.P;
    const int p = (int)(i % d.P);
    float acc = 0.f;
    for (int s = 0; s < d.S; ++s) acc += d.part[(((size_t)b * d.S + s) * (d.max_iter + 1)) * d.P + p];
    const float v = what == 0 ? sqrtf(acc) : acc;
    d.scal[i] = v;
    if (ti >= 0) {
      tptr(d, ti, tj)[i] = v;
      if (sym) tptr(d, tj, ti)[i] = v;
    }
    if (check_small && fabsf(v) > 1e-6f) big = 1.f;
  }
  if (check_small) {
    const float any = block_sum256(big, red);
    if (threadIdx.x == 0) d.ctrl->all_small = any > 0.f ? 0 : 1;
  }
}

// r = Aq - beta_prev * q_prev  (:108), beta_prev = t[k, k-1]
__global__ __launch_bounds__(kThreads) void k_lz_sub_prev(LzDev d, int k) {
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + col;
  const float beta = tptr(d, k, k - 1)[(size_t)b * c + col];
  const float* qp = d.q + qoff(d, k - 1);
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    d.r[i] = d.r[i] - qp[i] * beta;
  }
}

}  // namespace lo

using namespace lo;

extern "C" {

static void lz_layout(const lo_op_desc* op, int64_t P, int max_iter, Arena& ar, LzDev* d, Split* spo) {
  Split sp = choose_split(op->B, op->N, 256);
  *spo = sp;
  d->B = op->B; d->N = op->N; d->P = (int)P; d->S = sp.S; d->rows = sp.rows; d->max_iter = max_iter;
  d->ctrl = ar.take<LzCtrl>(1);
  d->r = ar.take<float>((size_t)op->B * op->N * P);
  d->part = ar.take<float>((size_t)op->B * sp.S * (max_iter + 1) * P);
  d->coef = ar.take<float>((size_t)op->B * (max_iter + 1) * P);
  d->scal = ar.take<float>((size_t)op->B * P);
}

size_t lo_lanczos_workspace_bytes(const lo_op_desc* op, int64_t P, int32_t max_iter) {
  if (!op) return 0;
  Arena ar(nullptr, 0);
  LzDev d;
  Split sp;
  lz_layout(op, P, max_iter, ar, &d, &sp);
  return ar.off + matvec_plan_bytes(op, P, sp) + 1024;
}

int lo_lanczos_tridiag_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const float* init_vecs,
                           int64_t P, int32_t max_iter, float tol, float* q_mat, float* t_mat, int32_t* iters_out,
                           void* ws, size_t ws_bytes, void* stream) {
  if (!op || !init_vecs || !q_mat || !t_mat || !iters_out || !ws || max_iter < 1) return LO_ERR_BADARG;
  if (P < 1 || P > kMaxCols) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = op->B, N = op->N;
  const int num_iter = (int)std::min<int64_t>(max_iter, N);  // :57
  Arena ar(ws, ws_bytes);
  LzDev d;
  Split sp;
  lz_layout(op, P, max_iter, ar, &d, &sp);
  d.tol = tol;
  d.q = q_mat;
  d.t = t_mat;
  MatvecPlan pl;
  int rc = matvec_plan_init(&pl, op, matvec, matvec_user, P, sp, &ar, st);
  if (rc) return rc;
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const size_t nv = (size_t)B * N * P;
  dim3 grid(sp.S, (unsigned)B), block(kThreads), one(1);
  LO_HIP_CHECK(hipMemsetAsync(q_mat, 0, sizeof(float) * nv * max_iter, st));
  LO_HIP_CHECK(hipMemsetAsync(t_mat, 0, sizeof(float) * (size_t)max_iter * max_iter * B * P, st));
  LzCtrl h;
  auto q = [&](int k) { return q_mat + (size_t)k * nv; };

  // q_0 = init / ||init||  (:81-82)
  hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, init_vecs, init_vecs);
  hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, -1, -1, 0, 0);
  hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, init_vecs, q(0), d.scal);
  LO_LAUNCH_CHECK();
  // r = A q_0 ; alpha_0 ; r -= alpha_0 q_0 ; beta_0 = ||r||  (:85-95)
  rc = matvec_run(&pl, q(0), d.r, nullptr, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, q(0), d.r);
  hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 1, 0, 0, 0, 0);
  hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 1, q(0), d.r, d.scal);
  LO_LAUNCH_CHECK();
  int k = 0;
  if (num_iter > 1) {
    hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, d.r, d.r);
    hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, 0, 1, 1, 0);
    hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, d.r, q(1), d.scal);  // q_1 = r / beta_0  (:98)
    LO_LAUNCH_CHECK();
    for (k = 1; k < num_iter; ++k) {
      rc = matvec_run(&pl, q(k), d.r, nullptr, nullptr, st);  // :108
      if (rc) return rc;
      hipLaunchKernelGGL(k_lz_sub_prev, grid, block, 0, st, d, k);
      hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, q(k), d.r);
      hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 1, k, k, 0, 0);  // alpha_k -> t[k,k]  (:109-111)
      LO_LAUNCH_CHECK();
      if (k + 1 < num_iter) {  // :114
        hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 1, q(k), d.r, d.scal);  // r -= alpha q_k  (:116)
        // full re-orthogonalisation (:118-120)
        hipLaunchKernelGGL(k_lz_multidot, grid, block, 0, st, d, k + 1);
        hipLaunchKernelGGL(k_lz_reduce, one, block, 0, st, d, k + 1, 0);
        hipLaunchKernelGGL(k_lz_correct, grid, block, 0, st, d, k + 1);
        // normalise; beta_k -> t[k,k+1], t[k+1,k]  (:121-128)
        hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, d.r, d.r);
        hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, k, k + 1, 1, 1);
        hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, d.r, d.r, d.scal);
        // inner products with the normalised r (:131)
        hipLaunchKernelGGL(k_lz_multidot, grid, block, 0, st, d, k + 1);
        hipLaunchKernelGGL(k_lz_reduce, one, block, 0, st, d, k + 1, 1);
        LO_LAUNCH_CHECK();
        LO_HIP_CHECK(hipMemcpyAsync(&h, d.ctrl, sizeof(LzCtrl), hipMemcpyDeviceToHost, st));
        LO_HIP_CHECK(hipStreamSynchronize(st));
        const int all_small = h.all_small;
        bool could = false;
        for (int it = 0; it < 10; ++it) {  // :133-142
          if (!h.need_reorth) {
            could = true;
            break;
          }
          hipLaunchKernelGGL(k_lz_correct, grid, block, 0, st, d, k + 1);  // uses the coefficients just computed
          hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, d.r, d.r);
          hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, -1, -1, 0, 0);
          hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, d.r, d.r, d.scal);
          hipLaunchKernelGGL(k_lz_multidot, grid, block, 0, st, d, k + 1);
          hipLaunchKernelGGL(k_lz_reduce, one, block, 0, st, d, k + 1, 1);
          LO_LAUNCH_CHECK();
          LO_HIP_CHECK(hipMemcpyAsync(&h, d.ctrl, sizeof(LzCtrl), hipMemcpyDeviceToHost, st));
          LO_HIP_CHECK(hipStreamSynchronize(st));
        }
        LO_HIP_CHECK(hipMemcpyAsync(q(k + 1), d.r, sizeof(float) * nv, hipMemcpyDeviceToDevice, st));  // :145
        if (all_small || !could) break;  // :147
      }
    }
    if (k == num_iter) k = num_iter - 1;
  }
  LO_HIP_CHECK(hipStreamSynchronize(st));
  *iters_out = k + 1;  // :151
  return LO_OK;
}

}  // extern "C"
