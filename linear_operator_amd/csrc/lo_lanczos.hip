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
#include <stdlib.h>
#include <cstring>
#include <stdint.h>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

struct LzCtrl {
  int need_reorth;  // sum(inner_products > tol) != 0
  int all_small;    // sum(|beta| > 1e-6) == 0
  int any_big;      // fused path: some |beta| > 1e-6 (all_small = !any_big)
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
  float* dot_part;  // [B, S_dot, P] partials of q_k . (A q_k) from the matvec epilogue (fused path)
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

// part[b,s,j,p] = sum_rows r o q_j  for j = 0..nq-1 in ONE pass over the rows: a thread keeps one accumulator per
// previous vector (compile-time bound MAXQ so that the accumulators stay in registers), r is read once.
constexpr int kLzMaxQ = 24;
constexpr int kLzFusedQ = 20;  // fused step: basis vectors per pass (max_lanczos_quadrature_iterations = 20 -> nq <= 19)

template <int MAXQ>
__global__ __launch_bounds__(kThreads) void k_lz_multidot(LzDev d, int nq) {
  __shared__ float red[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + col;
  const size_t qs = (size_t)d.B * d.N * d.P;
  float acc[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) acc[j] = 0.f;
  if (slot < nrs) {
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      const float rv = d.r[i];
#pragma unroll
      for (int j = 0; j < MAXQ; ++j)
        if (j < nq) acc[j] = fmaf(rv, d.q[(size_t)j * qs + i], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    if (j < nq) {  // nq is uniform: no divergent barrier
      const float tot = block_colsum(slot < nrs ? acc[j] : 0.f, c, nrs, red);
      if (threadIdx.x < c) d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * c + col] = tot;
    }
  }
}

// float4 variants for P % 4 == 0 (P <= 64): a thread owns 4 consecutive probe columns, so a wave instruction moves
// 1 KiB contiguous instead of 256 B.  Thread t: column quad t % (P/4), row slot t / (P/4).  The sums over the row
// slots of a wave are butterflies over the lane bits above log2(P/4) (DPP / permlane, no LDS, no barrier); the 4
// wave partials meet in LDS once.
template <int MAXQ>
__global__ __launch_bounds__(kThreads) void k_lz_multidot4(LzDev d, int nq) {
  __shared__ float red[4][MAXQ * 64];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, c4 = c >> 2, N = (int)d.N;
  const int nrs = kThreads / c4;  // power-of-two P/4 only: 1, 2, 4, 8, 16
  const int cq = threadIdx.x % c4, slot = threadIdx.x / c4;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + 4 * cq;
  const size_t qs = (size_t)d.B * d.N * d.P;
  float4 acc[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    const float4 rv = *reinterpret_cast<const float4*>(d.r + i);
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
      if (j < nq) {
        const float4 qv = *reinterpret_cast<const float4*>(d.q + (size_t)j * qs + i);
        acc[j].x = fmaf(rv.x, qv.x, acc[j].x);
        acc[j].y = fmaf(rv.y, qv.y, acc[j].y);
        acc[j].z = fmaf(rv.z, qv.z, acc[j].z);
        acc[j].w = fmaf(rv.w, qv.w, acc[j].w);
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    if (j < nq) {
      float v[4] = {acc[j].x, acc[j].y, acc[j].z, acc[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (c4 <= 1) x = bfly_add<1>(x);
        if (c4 <= 2) x = bfly_add<2>(x);
        if (c4 <= 4) x = bfly_add<4>(x);
        if (c4 <= 8) x = bfly_add<8>(x);
        if (c4 <= 16) x = bfly_add<16>(x);
        x = bfly_add<32>(x);
        if (lane < c4) red[wave][j * 64 + 4 * lane + e] = x;  // lane < c4: column quad = lane
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nq * c; i += kThreads) {
    const int j = i / c, p = i % c;
    const float tot = (red[0][j * 64 + p] + red[1][j * 64 + p]) + (red[2][j * 64 + p] + red[3][j * 64 + p]);
    d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * c + p] = tot;
  }
}

template <int MAXQ>
__global__ __launch_bounds__(kThreads) void k_lz_correct4(LzDev d, int nq) {
  __shared__ float red[4][64];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, c4 = c >> 2, N = (int)d.N;
  const int nrs = kThreads / c4;
  const int cq = threadIdx.x % c4, slot = threadIdx.x / c4;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + 4 * cq;
  const size_t qs = (size_t)d.B * d.N * d.P;
  float4 cf[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j)
    cf[j] = (j < nq) ? *reinterpret_cast<const float4*>(d.coef + ((size_t)b * (d.max_iter + 1) + j) * c + 4 * cq)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    float4 corr = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
      if (j < nq) {
        const float4 qv = *reinterpret_cast<const float4*>(d.q + (size_t)j * qs + i);
        corr.x = fmaf(qv.x, cf[j].x, corr.x);
        corr.y = fmaf(qv.y, cf[j].y, corr.y);
        corr.z = fmaf(qv.z, cf[j].z, corr.z);
        corr.w = fmaf(qv.w, cf[j].w, corr.w);
      }
    }
    float4 rv = *reinterpret_cast<const float4*>(d.r + i);
    rv.x -= corr.x; rv.y -= corr.y; rv.z -= corr.z; rv.w -= corr.w;
    *reinterpret_cast<float4*>(d.r + i) = rv;
    rr.x = fmaf(rv.x, rv.x, rr.x);
    rr.y = fmaf(rv.y, rv.y, rr.y);
    rr.z = fmaf(rv.z, rv.z, rr.z);
    rr.w = fmaf(rv.w, rv.w, rr.w);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float v[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = v[e];
    if (c4 <= 1) x = bfly_add<1>(x);
    if (c4 <= 2) x = bfly_add<2>(x);
    if (c4 <= 4) x = bfly_add<4>(x);
    if (c4 <= 8) x = bfly_add<8>(x);
    if (c4 <= 16) x = bfly_add<16>(x);
    x = bfly_add<32>(x);
    if (lane < c4) red[wave][4 * lane + e] = x;
  }
  __syncthreads();
  if ((int)threadIdx.x < c) {
    const int p = threadIdx.x;
    d.part[(((size_t)b * d.S + s) * (d.max_iter + 1)) * c + p] = (red[0][p] + red[1][p]) + (red[2][p] + red[3][p]);
  }
}

// the same, any number of previous vectors (one pass per vector)
__global__ __launch_bounds__(kThreads) void k_lz_multidot_any(LzDev d, int nq) {
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

// coef[b,j,p] = sum_s part[b,s,j,p]; check: also flag any coef > tol (signed, :134).  One workgroup per member;
// the flag word is cleared by the host-side memset before the launch.
__global__ __launch_bounds__(kThreads) void k_lz_reduce(LzDev d, int nq, int check) {
  __shared__ float red[kThreads];
  const int64_t b = blockIdx.x;
  float flag = 0.f;
  for (int i = threadIdx.x; i < nq * d.P; i += kThreads) {
    const int p = i % d.P, j = i / d.P;
    float acc = 0.f;
    for (int s = 0; s < d.S; ++s) acc += d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * d.P + p];
    d.coef[((size_t)b * (d.max_iter + 1) + j) * d.P + p] = acc;
    if (check && acc > d.tol) flag = 1.f;
  }
  if (check) {
    const float any = block_sum256(flag, red);
    if (threadIdx.x == 0 && any > 0.f) atomicExch(&d.ctrl->need_reorth, 1);
  }
}

// r -= sum_j coef[b,j,p] * q_j   (:119-120), and the partial of ||r||^2 of the corrected r (slot 0 of `part`,
// what the normalisation that follows needs: saves a pass).  Coefficients of the thread's column in registers.
template <int MAXQ>
__global__ __launch_bounds__(kThreads) void k_lz_correct(LzDev d, int nq) {
  __shared__ float red[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + col;
  const size_t qs = (size_t)d.B * d.N * d.P;
  float cf[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j)
    cf[j] = (j < nq && slot < nrs) ? d.coef[((size_t)b * (d.max_iter + 1) + j) * c + col] : 0.f;
  float rr = 0.f;
  if (slot < nrs) {
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      float corr = 0.f;
#pragma unroll
      for (int j = 0; j < MAXQ; ++j)
        if (j < nq) corr = fmaf(d.q[(size_t)j * qs + i], cf[j], corr);
      const float rn = d.r[i] - corr;
      d.r[i] = rn;
      rr = fmaf(rn, rn, rr);
    }
  }
  const float tot = block_colsum(slot < nrs ? rr : 0.f, c, nrs, red);
  if (threadIdx.x < c) d.part[(((size_t)b * d.S + s) * (d.max_iter + 1)) * c + col] = tot;
}

__global__ __launch_bounds__(kThreads) void k_lz_correct_any(LzDev d, int nq) {
  __shared__ float red[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + col;
  float rr = 0.f;
  if (slot < nrs) {
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      float corr = 0.f;
      for (int j = 0; j < nq; ++j)
        corr = fmaf(d.q[qoff(d, j) + i], d.coef[((size_t)b * (d.max_iter + 1) + j) * c + col], corr);
      const float rn = d.r[i] - corr;
      d.r[i] = rn;
      rr = fmaf(rn, rn, rr);
    }
  }
  const float tot = block_colsum(slot < nrs ? rr : 0.f, c, nrs, red);
  if (threadIdx.x < c) d.part[(((size_t)b * d.S + s) * (d.max_iter + 1)) * c + col] = tot;
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


// r -= beta_{k-1} q_{k-1} (:108) and the partials of alpha_k = q_k . r (:109) in one pass
__global__ __launch_bounds__(kThreads) void k_lz_sub_prev_dot(LzDev d, int k) {
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
    const float beta = tptr(d, k, k - 1)[(size_t)b * c + col];
    const float* qp = d.q + qoff(d, k - 1);
    const float* qk = d.q + qoff(d, k);
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      const float rv = d.r[i] - qp[i] * beta;
      d.r[i] = rv;
      acc = fmaf(qk[i], rv, acc);
    }
  }
  const float tot = block_colsum(slot < nrs ? acc : 0.f, c, nrs, red);
  if (threadIdx.x < c) d.part[(((size_t)b * d.S + s) * (d.max_iter + 1)) * c + col] = tot;
}


// ---- fused step (P a power of two in 4 .. 64, at most kLzMaxQ basis vectors) ---------------------------------------
// The step of lanczos.py:108-145 with Q_{<=k} read from HBM TWICE instead of three times and without the separate
// vector passes: the raw product r = A q_k stays untouched; both passes form
//     rv = (r - beta_{k-1} q_{k-1}) - alpha_k q_k                      (:108, :116)
// on the fly from vectors they read anyway.  alpha_k = q_k . (A q_k) - beta_{k-1} (q_{k-1} . q_k) comes from the dot the
// matvec epilogue fuses and from the orthogonality check of the previous step (:109-111 in exact arithmetic).
//   pass 1 (k_lz_dots_fused):     c_j = q_j . rv for all j <= k                                      (:118)
//   pass 2 (k_lz_correct_check):  rv -= sum_j c_j q_j, scaled with the predicted norm and written into q_{k+1}; partials of
//                                 its squared norm and of the check products q_j . q_{k+1} (:119-131) -- the row's q_j
//                                 values are still in registers
//   k_lz_finish:                  beta_k = exact norm, t entries, check products, the two batch-global flags
template <int MAXQ>
__global__ __launch_bounds__(kThreads) void k_lz_dots_fused(LzDev d, int k) {
  __shared__ float red[4][(MAXQ + 1) * 64];
  const int nq = k + 1;
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, c4 = c >> 2, N = (int)d.N;
  const int nrs = kThreads / c4;
  const int cq = threadIdx.x % c4, slot = threadIdx.x / c4;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + 4 * cq;
  const size_t qs = (size_t)d.B * d.N * d.P;
  const float4 al = *reinterpret_cast<const float4*>(d.scal + (size_t)b * c + 4 * cq);
  const float4 be = *reinterpret_cast<const float4*>(tptr(d, k, k - 1) + (size_t)b * c + 4 * cq);
  float4 acc[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 nn = make_float4(0.f, 0.f, 0.f, 0.f);  // ||rv||^2 (slot nq)
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    float4 rv = *reinterpret_cast<const float4*>(d.r + i);
    const float4 qp = *reinterpret_cast<const float4*>(d.q + (size_t)(k - 1) * qs + i);
    const float4 qk = *reinterpret_cast<const float4*>(d.q + (size_t)k * qs + i);
    rv.x = (rv.x - qp.x * be.x) - al.x * qk.x;
    rv.y = (rv.y - qp.y * be.y) - al.y * qk.y;
    rv.z = (rv.z - qp.z * be.z) - al.z * qk.z;
    rv.w = (rv.w - qp.w * be.w) - al.w * qk.w;
    nn.x = fmaf(rv.x, rv.x, nn.x);
    nn.y = fmaf(rv.y, rv.y, nn.y);
    nn.z = fmaf(rv.z, rv.z, nn.z);
    nn.w = fmaf(rv.w, rv.w, nn.w);
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
      if (j < nq) {
        const float4 qv = *reinterpret_cast<const float4*>(d.q + (size_t)j * qs + i);
        acc[j].x = fmaf(rv.x, qv.x, acc[j].x);
        acc[j].y = fmaf(rv.y, qv.y, acc[j].y);
        acc[j].z = fmaf(rv.z, qv.z, acc[j].z);
        acc[j].w = fmaf(rv.w, qv.w, acc[j].w);
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j <= MAXQ; ++j) {
    if (j <= nq) {
      const float4 a4 = (j < MAXQ && j < nq) ? acc[j < MAXQ ? j : 0] : nn;
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (c4 <= 1) x = bfly_add<1>(x);
        if (c4 <= 2) x = bfly_add<2>(x);
        if (c4 <= 4) x = bfly_add<4>(x);
        if (c4 <= 8) x = bfly_add<8>(x);
        if (c4 <= 16) x = bfly_add<16>(x);
        x = bfly_add<32>(x);
        if (lane < c4) red[wave][j * 64 + 4 * lane + e] = x;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (nq + 1) * c; i += kThreads) {
    const int j = i / c, p = i % c;
    const float tot = (red[0][j * 64 + p] + red[1][j * 64 + p]) + (red[2][j * 64 + p] + red[3][j * 64 + p]);
    d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * c + p] = tot;
  }
}

// sum of S partials at stride `sstride`, eight loads in flight, added in slice order (the bits of the plain loop)
__device__ __forceinline__ float lz_sum_slices(const float* __restrict__ src, size_t sstride, int S) {
  float acc = 0.f;
  for (int s0 = 0; s0 < S; s0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (s0 + u < S) ? src[(size_t)(s0 + u) * sstride] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < S) acc += v[u];
  }
  return acc;
}

// coef[b, j, p] = sum_s part (j < nq) and the PREDICTED norm of the corrected vector: with an orthonormal basis
// ||rv - Q c||^2 = ||rv||^2 - |c|^2 (the c_j are rounding-size: the components along q_k, q_{k-1} were just removed), so
// pass 2 can write q_{k+1} already scaled by 1 / beta_pred.  The exact norm is measured in pass 2 and corrects beta_k.
__global__ __launch_bounds__(kThreads) void k_lz_reduce_pred(LzDev d, int nq) {
  // thread = one (j, p) pair: its S partials are loaded eight at a time and added in slice order (the bits of the
  // one-thread-per-probe loop this replaces, which kept 16 of 256 threads busy with (nq + 1) S dependent loads:
  // 31 us per step on average, 100 us at step 19); the probe's |c|^2 is then summed over j in order by thread p
  __shared__ float acc_s[(kLzFusedQ + 2) * 64];
  const int64_t b = blockIdx.x;
  const int P = d.P, items = (nq + 1) * P;
  const bool fits = items <= (kLzFusedQ + 2) * 64 && P <= 64;
  if (fits) {
    for (int e = threadIdx.x; e < items; e += kThreads) {
      const int j = e / P, p = e - j * P;
      const float* src = d.part + ((size_t)b * d.S * (d.max_iter + 1) + j) * P + p;
      const size_t sstride = (size_t)(d.max_iter + 1) * P;
      const float acc = lz_sum_slices(src, sstride, d.S);
      acc_s[e] = acc;
      if (j < nq) d.coef[((size_t)b * (d.max_iter + 1) + j) * P + p] = acc;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += kThreads) {
      float cc = 0.f;
      for (int j = 0; j < nq; ++j) cc = fmaf(acc_s[j * P + p], acc_s[j * P + p], cc);
      const float nrm2 = acc_s[nq * P + p];
      d.scal[(size_t)b * P + p + (size_t)d.B * P] = sqrtf(fmaxf(nrm2 - cc, 1e-30f));  // beta_pred (second half of scal)
    }
    return;
  }
  for (int p = threadIdx.x; p < d.P; p += kThreads) {
    float cc = 0.f, nrm2 = 0.f;
    for (int j = 0; j <= nq; ++j) {
      float acc = 0.f;
      for (int s = 0; s < d.S; ++s) acc += d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * d.P + p];
      if (j < nq) {
        d.coef[((size_t)b * (d.max_iter + 1) + j) * d.P + p] = acc;
        cc = fmaf(acc, acc, cc);
      } else {
        nrm2 = acc;
      }
    }
    d.scal[(size_t)b * d.P + p + (size_t)d.B * d.P] = sqrtf(fmaxf(nrm2 - cc, 1e-30f));  // beta_pred (second half of scal)
  }
}

// pass 2: a thread owns TWO probe columns (coefficients, the row's basis values and the check accumulators of MAXQ
// vectors stay in registers: 3 x 2 x MAXQ).  part[b, s, j] (j < nq) = check products, part[b, s, nq] = ||rv||^2.
template <int MAXQ>
__global__ __launch_bounds__(kThreads) void k_lz_correct_check(LzDev d, int k, float* __restrict__ out) {
  __shared__ float red[4][(MAXQ + 1) * 64];
  const int nq = k + 1;
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c = d.P, c2 = c >> 1, N = (int)d.N;
  const int nrs = kThreads / c2;  // c2 = 2 .. 32 (power of two)
  const int cp = threadIdx.x % c2, slot = threadIdx.x / c2;
  const int r0 = s * d.rows, r1 = min(N, r0 + d.rows);
  const size_t base = (size_t)b * N * c + 2 * cp;
  const size_t qs = (size_t)d.B * d.N * d.P;
  const float2 al = *reinterpret_cast<const float2*>(d.scal + (size_t)b * c + 2 * cp);
  const float2 be = *reinterpret_cast<const float2*>(tptr(d, k, k - 1) + (size_t)b * c + 2 * cp);
  const float2 bp = *reinterpret_cast<const float2*>(d.scal + (size_t)d.B * c + (size_t)b * c + 2 * cp);  // beta_pred
  const float2 sc = make_float2(1.0f / bp.x, 1.0f / bp.y);
  float2 cf[MAXQ], chk[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    cf[j] = (j < nq) ? *reinterpret_cast<const float2*>(d.coef + ((size_t)b * (d.max_iter + 1) + j) * c + 2 * cp)
                     : make_float2(0.f, 0.f);
    chk[j] = make_float2(0.f, 0.f);
  }
  float2 rr = make_float2(0.f, 0.f);
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    float2 qv[MAXQ];
#pragma unroll
    for (int j = 0; j < MAXQ; ++j)
      qv[j] = (j < nq) ? *reinterpret_cast<const float2*>(d.q + (size_t)j * qs + i) : make_float2(0.f, 0.f);
    float2 rv = *reinterpret_cast<const float2*>(d.r + i);
    float2 qp = make_float2(0.f, 0.f), qk = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {  // (k is uniform: selects, no divergence)
      if (j == k - 1) qp = qv[j];
      if (j == k) qk = qv[j];
    }
    rv.x = (rv.x - qp.x * be.x) - al.x * qk.x;
    rv.y = (rv.y - qp.y * be.y) - al.y * qk.y;
    float2 corr = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
      corr.x = fmaf(qv[j].x, cf[j].x, corr.x);
      corr.y = fmaf(qv[j].y, cf[j].y, corr.y);
    }
    rv.x = (rv.x - corr.x) * sc.x;   // (:119-128) corrected and scaled with the predicted norm
    rv.y = (rv.y - corr.y) * sc.y;
    *reinterpret_cast<float2*>(out + i) = rv;
    rr.x = fmaf(rv.x, rv.x, rr.x);
    rr.y = fmaf(rv.y, rv.y, rr.y);
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
      chk[j].x = fmaf(rv.x, qv[j].x, chk[j].x);
      chk[j].y = fmaf(rv.y, qv[j].y, chk[j].y);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j <= MAXQ; ++j) {
    if (j <= nq) {
      const float2 val = (j < MAXQ && j < nq) ? chk[j < MAXQ ? j : 0] : rr;  // slot nq carries ||rv||^2
      float v[2] = {val.x, val.y};
      if (j < nq || j == nq) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float x = v[e];
          if (c2 <= 1) x = bfly_add<1>(x);
          if (c2 <= 2) x = bfly_add<2>(x);
          if (c2 <= 4) x = bfly_add<4>(x);
          if (c2 <= 8) x = bfly_add<8>(x);
          if (c2 <= 16) x = bfly_add<16>(x);
          x = bfly_add<32>(x);
          if (lane < c2) red[wave][j * 64 + 2 * lane + e] = x;
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (nq + 1) * c; i += kThreads) {
    const int j = i / c, p = i % c;
    const float tot = (red[0][j * 64 + p] + red[1][j * 64 + p]) + (red[2][j * 64 + p] + red[3][j * 64 + p]);
    d.part[(((size_t)b * d.S + s) * (d.max_iter + 1) + j) * c + p] = tot;
  }
}

// alpha_k = sum_s dot_part - beta_{k-1} (q_{k-1} . q_k) -> scal and t[k, k]  (:109-111).  The product q_{k-1} . q_k is
// the normalised check product j = k-1 of the previous step (coef); no check exists before step 1.
__global__ __launch_bounds__(kThreads) void k_lz_alpha(LzDev d, int k, const float* __restrict__ dot_part, int S_dot) {
  const int64_t n = d.B * d.P;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / d.P;
    const int p = (int)(i % d.P);
    float acc = 0.f;
    for (int s = 0; s < S_dot; ++s) acc += dot_part[((size_t)b * S_dot + s) * d.P + p];
    if (k > 1) acc -= tptr(d, k, k - 1)[i] * d.coef[((size_t)b * (d.max_iter + 1) + (k - 1)) * d.P + p];
    d.scal[i] = acc;
    tptr(d, k, k)[i] = acc;
  }
}

// after pass 2: beta_k = sqrt(sum ||rv||^2) -> scal, t[k, k+1], t[k+1, k] (:121-128); normalised check products -> coef,
// need_reorth if any exceeds tol (signed, :131-134); any_big if some |beta_k| > 1e-6 (:147).  One workgroup per member.
__global__ __launch_bounds__(kThreads) void k_lz_finish(LzDev d, int k) {
  __shared__ float beta_s[kMaxCols];
  __shared__ float red[kThreads];
  const int nq = k + 1;
  const int64_t b = blockIdx.x;
  float big = 0.f, flag = 0.f;
  for (int p = threadIdx.x; p < d.P; p += kThreads) {
    const float acc = lz_sum_slices(d.part + ((size_t)b * d.S * (d.max_iter + 1) + nq) * d.P + p,
                                    (size_t)(d.max_iter + 1) * d.P, d.S);
    const float nrm = sqrtf(acc);  // norm of the stored (scaled) vector: 1 up to rounding
    const float beta = d.scal[(size_t)d.B * d.P + (size_t)b * d.P + p] * nrm;  // the exact ||rv - Q c||
    beta_s[p] = nrm;
    d.scal[(size_t)b * d.P + p] = beta;
    tptr(d, k, k + 1)[(size_t)b * d.P + p] = beta;
    tptr(d, k + 1, k)[(size_t)b * d.P + p] = beta;
    if (fabsf(beta) > 1e-6f) big = 1.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nq * d.P; i += kThreads) {
    const int p = i % d.P, j = i / d.P;
    const float acc = lz_sum_slices(d.part + ((size_t)b * d.S * (d.max_iter + 1) + j) * d.P + p,
                                    (size_t)(d.max_iter + 1) * d.P, d.S);
    const float v = acc / beta_s[p];  // product with the NORMALISED vector (:131)
    d.coef[((size_t)b * (d.max_iter + 1) + j) * d.P + p] = v;
    if (v > d.tol) flag = 1.f;
  }
  const float anyflag = block_sum256(flag, red);
  const float anybig = block_sum256(big, red);
  if (threadIdx.x == 0) {
    if (anyflag > 0.f) atomicExch(&d.ctrl->need_reorth, 1);
    if (anybig > 0.f) atomicExch(&d.ctrl->any_big, 1);
  }
}

// q_mat [kstore, B, N, P] (working order, :69-76) -> [P, B, N, k] (returned order, lanczos.py:154): LDS-tiled
// transpose, reads contiguous along P (and rows), writes contiguous along (row, k).  One workgroup = 32 rows of a member.
// Tile [k][kLzTr][P + 1] with ONE extra float per vector j: kLzTr (P + 1) is a multiple of the 64 banks for P = 16, and
// the write-out walks j fastest.
constexpr int kLzTr = 32;
__host__ __device__ constexpr int lz_tile_js(int P) { return kLzTr * (P + 1) + 1; }

__global__ __launch_bounds__(kThreads) void k_lz_permute(const float* __restrict__ qin, float* __restrict__ qout,
                                                          int k, int64_t B, int N, int P) {
  extern __shared__ float tile[];  // [k][kLzTr][P + 1] (+ 1 per j)
  const int64_t b = blockIdx.y;
  const int n0 = blockIdx.x * kLzTr, nr = min(kLzTr, N - n0);
  const int ld = P + 1, js = lz_tile_js(P);
  for (int e = threadIdx.x; e < k * nr * P; e += kThreads) {
    const int j = e / (nr * P), rem = e % (nr * P);  // rem = row * P + p: contiguous in qin
    tile[j * js + (rem / P) * ld + rem % P] = qin[(((size_t)j * B + b) * N + n0) * P + rem];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < P * nr * k; e += kThreads) {
    const int p = e / (nr * k), rem = e % (nr * k);  // rem = row * k + j: contiguous in qout
    qout[(((size_t)p * B + b) * N + n0) * k + rem] = tile[(rem % k) * js + (rem / k) * ld + p];
  }
}

// the same with 16-byte global accesses (P % 4 == 0 and k % 4 == 0: four consecutive outputs share a row)
__global__ __launch_bounds__(kThreads) void k_lz_permute4(const float* __restrict__ qin, float* __restrict__ qout,
                                                           int k, int64_t B, int N, int P) {
  extern __shared__ float tile[];  // [k][kLzTr][P + 1] (+ 1 per j)
  const int64_t b = blockIdx.y;
  const int n0 = blockIdx.x * kLzTr, nr = min(kLzTr, N - n0);
  const int ld = P + 1, js = lz_tile_js(P);
  const int nin = nr * P / 4;
  for (int e = threadIdx.x; e < k * nin; e += kThreads) {
    const int j = e / nin, rem = 4 * (e % nin);  // rem = row * P + p
    const float4 v = *reinterpret_cast<const float4*>(qin + (((size_t)j * B + b) * N + n0) * P + rem);
    float* t = tile + j * js + (rem / P) * ld + rem % P;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  const int nout = nr * k / 4;
  for (int e = threadIdx.x; e < P * nout; e += kThreads) {
    const int p = e / nout, rem = 4 * (e % nout);  // rem = row * k + j, j % 4 == 0
    const int row = rem / k, j = rem % k;
    const float* t = tile + j * js + row * ld + p;
    const float4 v = make_float4(t[0], t[js], t[2 * js], t[3 * js]);
    *reinterpret_cast<float4*>(qout + (((size_t)p * B + b) * N + n0) * k + rem) = v;
  }
}

// Epilogue of RootDecomposition.forward (functions/_root_decomposition.py:73-85): QV = Q V, root = QV o sqrt(lambda),
// inverse = QV / sqrt(lambda) for a batch of tall Q [N, k] and small V [k, k] (k <= 32): one thread per row, the row
// of Q in registers, V and sqrt(lambda) in LDS.  HBM-bound: 4 N k (1 + outputs) bytes per member.
template <int KM>
__global__ __launch_bounds__(kThreads) void k_lz_root(const float* __restrict__ q, const float* __restrict__ evecs,
                                                       const float* __restrict__ evals, int N, int k,
                                                       float* __restrict__ qv, float* __restrict__ root,
                                                       float* __restrict__ inverse) {
  __shared__ float v_s[KM * KM];
  __shared__ float s_s[KM];
  const int64_t b = blockIdx.y;
  for (int e = threadIdx.x; e < k * k; e += kThreads) v_s[e] = evecs[(size_t)b * k * k + e];
  for (int e = threadIdx.x; e < k; e += kThreads) s_s[e] = sqrtf(evals[(size_t)b * k + e]);
  __syncthreads();
  const int row = blockIdx.x * kThreads + threadIdx.x;
  if (row >= N) return;
  const size_t o = ((size_t)b * N + row) * k;
  float x[KM];
#pragma unroll
  for (int a = 0; a < KM; ++a) x[a] = (a < k) ? q[o + a] : 0.f;
  for (int j = 0; j < k; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < KM; ++a)
      if (a < k) acc = fmaf(x[a], v_s[a * k + j], acc);
    if (qv) qv[o + j] = acc;
    if (root) root[o + j] = acc * s_s[j];
    if (inverse) inverse[o + j] = acc / s_s[j];
  }
}

// The same epilogue reading the Lanczos basis in the layout the step kernels WRITE it -- q_native [k, B, N, P] (vector a
// of probe p of member b: q[((a B + b) N + n) P + p]) -- so that the [P, B, N, k] copy of lanczos.py:154 is never made
// (k_lz_permute4 cost 0.9 - 2.2 ms per call at the cfg3 shape).  A workgroup takes TR rows x all P probes of a member:
// thread (row, p) gathers its k coefficients with P-contiguous loads (a wave instruction reads whole lines), multiplies
// by V_p (LDS, per-probe stride k k + 1: the 16 probes of a wave hit different banks) and the results leave through an
// LDS tile as TR k contiguous floats per probe.
template <int KM>
__global__ __launch_bounds__(kThreads) void k_lz_root_native(const float* __restrict__ q, const float* __restrict__ evecs,
                                                              const float* __restrict__ evals, int64_t B, int N, int P,
                                                              int k, int TR, float* __restrict__ qv,
                                                              float* __restrict__ root, float* __restrict__ inverse) {
  extern __shared__ float lz_sm[];
  const int vld = k * k + 1;
  float* v_s = lz_sm;                     // [P][k k + 1]
  float* s_s = v_s + (size_t)P * vld;     // [P][k]  sqrt(lambda)
  float* tile = s_s + (size_t)P * k;      // [P][TR][k]
  const int64_t b = blockIdx.y;
  const int r0 = blockIdx.x * TR;
  for (int e = threadIdx.x; e < P * k * k; e += kThreads) {
    const int pp = e / (k * k), ij = e % (k * k);
    v_s[pp * vld + ij] = evecs[((size_t)pp * B + b) * k * k + ij];
  }
  for (int e = threadIdx.x; e < P * k; e += kThreads) s_s[e] = sqrtf(evals[((size_t)(e / k) * B + b) * k + e % k]);
  const int e0 = threadIdx.x;
  const int p = e0 % P, row = e0 / P;
  const bool live = e0 < TR * P && r0 + row < N;
  float x[KM];
#pragma unroll
  for (int a = 0; a < KM; ++a)
    x[a] = (live && a < k) ? q[(((size_t)a * B + b) * N + r0 + row) * P + p] : 0.f;
  __syncthreads();
  float acc[KM];
#pragma unroll
  for (int j = 0; j < KM; ++j) acc[j] = 0.f;
  if (live) {
    const float* vp = v_s + p * vld;
#pragma unroll
    for (int a = 0; a < KM; ++a) {
      if (a < k) {  // (uniform)
        const float xa = x[a];
#pragma unroll
        for (int j = 0; j < KM; ++j)
          if (j < k) acc[j] = fmaf(xa, vp[a * k + j], acc[j]);
      }
    }
  }
  const int nrow = min(TR, N - r0);
  for (int which = 0; which < 3; ++which) {
    float* out = which == 0 ? qv : (which == 1 ? root : inverse);
    if (!out) continue;
    __syncthreads();
    if (live) {
#pragma unroll
      for (int j = 0; j < KM; ++j)
        if (j < k) {
          const float sj = s_s[p * k + j];
          tile[((size_t)p * TR + row) * k + j] = which == 0 ? acc[j] : (which == 1 ? acc[j] * sj : acc[j] / sj);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < P * nrow * k; e += kThreads) {
      const int pp = e / (nrow * k), rem = e % (nrow * k);
      out[(((size_t)pp * B + b) * N + r0) * k + rem] = tile[(size_t)pp * TR * k + rem];
    }
  }
}

}  // namespace lo

using namespace lo;

extern "C" {

int lo_root_from_lanczos_native_f32(const float* q_native, const float* evecs, const float* evals, int64_t B, int64_t N,
                                    int64_t P, int32_t k, float* qv, float* root, float* inverse, void* stream) {
  if (!q_native || !evecs || !evals || B < 1 || N < 1 || P < 1 || k < 1 || B > 65535) return LO_ERR_BADARG;
  if (k > 32 || P > kThreads) return LO_ERR_UNSUPPORTED;
  const int TR = kThreads / (int)P;
  const size_t lds = sizeof(float) * ((size_t)P * (k * k + 1) + (size_t)P * k + (size_t)P * TR * k);
  if (lds > 64 * 1024) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((N + TR - 1) / TR), (unsigned)B);
  LO_PROF_BEGIN("lz_root", st);
  if (k <= 16)
    hipLaunchKernelGGL((k_lz_root_native<16>), grid, dim3(kThreads), lds, st, q_native, evecs, evals, B, (int)N, (int)P,
                       (int)k, TR, qv, root, inverse);
  else
    hipLaunchKernelGGL((k_lz_root_native<32>), grid, dim3(kThreads), lds, st, q_native, evecs, evals, B, (int)N, (int)P,
                       (int)k, TR, qv, root, inverse);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

static void lz_layout(const lo_op_desc* op, int64_t P, int max_iter, Arena& ar, LzDev* d, Split* spo) {
  Split sp = choose_split(op->B, op->N, 256, 64);  // (the fused step keeps a member's dot partials in 64 slots)
  *spo = sp;
  d->B = op->B; d->N = op->N; d->P = (int)P; d->S = sp.S; d->rows = sp.rows; d->max_iter = max_iter;
  d->ctrl = ar.take<LzCtrl>(1);
  d->r = ar.take<float>((size_t)op->B * op->N * P);
  d->part = ar.take<float>((size_t)op->B * sp.S * (max_iter + 1) * P);
  d->coef = ar.take<float>((size_t)op->B * (max_iter + 1) * P);
  d->scal = ar.take<float>((size_t)op->B * P * 2);  // [B, P] scalars of the step | [B, P] predicted norms
  d->dot_part = ar.take<float>((size_t)op->B * 64 * P);  // matvec dot partials [B, S_dot <= 64, P]
}

size_t lo_lanczos_workspace_bytes(const lo_op_desc* op, int64_t P, int32_t max_iter) {
  if (!op) return 0;
  Arena ar(nullptr, 0);
  LzDev d;
  Split sp;
  lz_layout(op, P, max_iter, ar, &d, &sp);
  return ar.off + matvec_plan_bytes(op, P, sp) + 1024;
}

int lo_root_from_lanczos_f32(const float* q, const float* evecs, const float* evals, int64_t PB, int64_t N, int32_t k,
                             float* qv, float* root, float* inverse, void* stream) {
  if (!q || !evecs || !evals || PB < 1 || N < 1 || k < 1 || PB > 65535) return LO_ERR_BADARG;
  if (k > 32) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((N + kThreads - 1) / kThreads), (unsigned)PB);
  LO_PROF_BEGIN("lz_root", st);
  if (k <= 16) hipLaunchKernelGGL((k_lz_root<16>), grid, dim3(kThreads), 0, st, q, evecs, evals, (int)N, (int)k, qv, root, inverse);
  else hipLaunchKernelGGL((k_lz_root<32>), grid, dim3(kThreads), 0, st, q, evecs, evals, (int)N, (int)k, qv, root, inverse);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

int lo_lanczos_permute_f32(const float* q_in, int32_t k, int64_t B, int64_t N, int64_t P, float* q_out, void* stream) {
  if (!q_in || !q_out || k < 1 || B < 1 || N < 1 || P < 1 || B > 65535) return LO_ERR_BADARG;
  const size_t lds = sizeof(float) * (size_t)k * lz_tile_js((int)P);
  if (lds > 64 * 1024) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((N + kLzTr - 1) / kLzTr), (unsigned)B);
  const bool v4 = (P % 4 == 0) && (k % 4 == 0) && ((uintptr_t)q_in % 16 == 0) && ((uintptr_t)q_out % 16 == 0);
  LO_PROF_BEGIN("lz_permute", st);
  if (v4) hipLaunchKernelGGL(k_lz_permute4, grid, dim3(kThreads), lds, st, q_in, q_out, (int)k, B, (int)N, (int)P);
  else hipLaunchKernelGGL(k_lz_permute, grid, dim3(kThreads), lds, st, q_in, q_out, (int)k, B, (int)N, (int)P);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
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
  PlanGuard pl_guard(&pl);
  int rc = matvec_plan_init(&pl, op, matvec, matvec_user, P, sp, &ar, st);
  if (rc) return rc;
  if (!ar.ok) return LO_ERR_WORKSPACE;
  const size_t nv = (size_t)B * N * P;
  dim3 grid(sp.S, (unsigned)B), block(kThreads), one(1);
  // (q_mat is NOT cleared up front -- 5.4 GB at the cfg3 shape, 0.8 ms: every stored vector is written in full; only
  //  the vectors an early stop leaves unwritten are cleared at the end, as the reference's zeros-initialised q_mat :69)
  int q_hi = 0;  // highest basis vector written so far
  LO_HIP_CHECK(hipMemsetAsync(t_mat, 0, sizeof(float) * (size_t)max_iter * max_iter * B * P, st));
  LzCtrl h;
  auto q = [&](int k) { return q_mat + (size_t)k * nv; };
  // float4 kernels: P a power of two in 4 .. 64 (column quads per row 1 .. 16), 16-byte aligned rows
  const bool vec4 = (P == 4 || P == 8 || P == 16 || P == 32 || P == 64) && ((uintptr_t)q_mat % 16) == 0 &&
                    ((uintptr_t)d.r % 16) == 0 && ((uintptr_t)d.coef % 16) == 0;
  auto multidot = [&](const LzDev& d, int nq) {
    LO_PROF_BEGIN("lz_multidot", st);
    if (nq <= kLzMaxQ && vec4) hipLaunchKernelGGL((k_lz_multidot4<kLzMaxQ>), grid, block, 0, st, d, nq);
    else if (nq <= kLzMaxQ) hipLaunchKernelGGL((k_lz_multidot<kLzMaxQ>), grid, block, 0, st, d, nq);
    else hipLaunchKernelGGL(k_lz_multidot_any, grid, block, 0, st, d, nq);
    LO_PROF_END(st);
  };
  auto correct = [&](const LzDev& d, int nq) {  // also leaves the partials of ||r||^2 in slot 0 of `part`
    LO_PROF_BEGIN("lz_correct", st);
    if (nq <= kLzMaxQ && vec4) hipLaunchKernelGGL((k_lz_correct4<kLzMaxQ>), grid, block, 0, st, d, nq);
    else if (nq <= kLzMaxQ) hipLaunchKernelGGL((k_lz_correct<kLzMaxQ>), grid, block, 0, st, d, nq);
    else hipLaunchKernelGGL(k_lz_correct_any, grid, block, 0, st, d, nq);
    LO_PROF_END(st);
  };
  auto reduce = [&](const LzDev& d, int nq, int check) {
    if (check) (void)hipMemsetAsync(&d.ctrl->need_reorth, 0, sizeof(int), st);
    LO_PROF_BEGIN("lz_reduce", st);
    hipLaunchKernelGGL(k_lz_reduce, dim3((unsigned)B), block, 0, st, d, nq, check);
    LO_PROF_END(st);
  };

  // q_0 = init / ||init||  (:81-82)
  LO_PROF_BEGIN("lz_dot", st);
  hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, init_vecs, init_vecs);
  LO_PROF_END(st);
  LO_PROF_BEGIN("lz_scal", st);
  hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, -1, -1, 0, 0);
  LO_PROF_END(st);
  LO_PROF_BEGIN("lz_axpy", st);
  hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, init_vecs, q(0), d.scal);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  // r = A q_0 ; alpha_0 ; r -= alpha_0 q_0 ; beta_0 = ||r||  (:85-95)
  rc = matvec_run(&pl, q(0), d.r, nullptr, nullptr, st);
  if (rc) return rc;
  LO_PROF_BEGIN("lz_dot", st);
  hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, q(0), d.r);
  LO_PROF_END(st);
  LO_PROF_BEGIN("lz_scal", st);
  hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 1, 0, 0, 0, 0);
  LO_PROF_END(st);
  LO_PROF_BEGIN("lz_axpy", st);
  hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 1, q(0), d.r, d.scal);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  int k = 0;
  if (num_iter > 1) {
    LO_PROF_BEGIN("lz_dot", st);
    hipLaunchKernelGGL(k_lz_dot, grid, block, 0, st, d, d.r, d.r);
    LO_PROF_END(st);
    LO_PROF_BEGIN("lz_scal", st);
    hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, 0, 1, 1, 0);
    LO_PROF_END(st);
    LO_PROF_BEGIN("lz_axpy", st);
    hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, d.r, q(1), d.scal);  // q_1 = r / beta_0  (:98)
    LO_PROF_END(st);
    q_hi = 1;
    LO_LAUNCH_CHECK();
    // fused path: float4 / float2 kernels, all basis vectors of a step in registers, matvec dot partials fit
    // (kLzFusedQ accumulators keep pass 1 under 128 VGPRs: four waves per SIMD)
    const bool fused = vec4 && num_iter <= kLzFusedQ + 1 && pl.S_dot <= 64 && !getenv("LO_LZ_UNFUSED");
    for (k = 1; fused && k < num_iter; ++k) {
      rc = matvec_run(&pl, q(k), d.r, d.dot_part, nullptr, st);  // r = A q_k (:108) + partials of q_k . r
      if (rc) return rc;
      LO_PROF_BEGIN("lz_alpha", st);
      hipLaunchKernelGGL(k_lz_alpha, dim3((unsigned)std::min<int64_t>((B * P + kThreads - 1) / kThreads, 1024)), block,
                         0, st, d, k, d.dot_part, pl.S_dot);
      LO_PROF_END(st);
      LO_LAUNCH_CHECK();
      if (k + 1 >= num_iter) continue;  // :114 (the last step only needs alpha)
      LO_PROF_BEGIN("lz_dots_fused", st);
      hipLaunchKernelGGL((k_lz_dots_fused<kLzFusedQ>), grid, block, 0, st, d, k);
      LO_PROF_END(st);
      LO_PROF_BEGIN("lz_reduce", st);
      hipLaunchKernelGGL(k_lz_reduce_pred, dim3((unsigned)B), block, 0, st, d, k + 1);
      LO_PROF_END(st);
      (void)hipMemsetAsync(d.ctrl, 0, sizeof(LzCtrl), st);
      LO_PROF_BEGIN("lz_correct_check", st);
      hipLaunchKernelGGL((k_lz_correct_check<kLzFusedQ>), grid, block, 0, st, d, k, q(k + 1));
      LO_PROF_END(st);
      q_hi = std::max(q_hi, k + 1);
      LO_PROF_BEGIN("lz_finish", st);
      hipLaunchKernelGGL(k_lz_finish, dim3((unsigned)B), block, 0, st, d, k);
      LO_PROF_END(st);
      LO_LAUNCH_CHECK();  // (q_{k+1} is stored normalised by pass 2: no separate normalisation pass)
      LO_HIP_CHECK(hipMemcpyAsync(&h, d.ctrl, sizeof(LzCtrl), hipMemcpyDeviceToHost, st));
      LO_HIP_CHECK(hipStreamSynchronize(st));
      const int all_small = h.any_big ? 0 : 1;
      LzDev dn = d;
      dn.r = q(k + 1);
      bool could = false;
      for (int it = 0; it < 10; ++it) {  // :133-142 (rare: one Gram-Schmidt pass normally suffices)
        if (!h.need_reorth) {
          could = true;
          break;
        }
        correct(dn, k + 1);  // uses the normalised check products as coefficients
        LO_PROF_BEGIN("lz_scal", st);
        hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, -1, -1, 0, 0);
        LO_PROF_END(st);
        LO_PROF_BEGIN("lz_axpy", st);
        hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, dn.r, dn.r, d.scal);
        LO_PROF_END(st);
        multidot(dn, k + 1);
        reduce(dn, k + 1, 1);
        LO_LAUNCH_CHECK();
        LO_HIP_CHECK(hipMemcpyAsync(&h, d.ctrl, sizeof(LzCtrl), hipMemcpyDeviceToHost, st));
        LO_HIP_CHECK(hipStreamSynchronize(st));
      }
      if (all_small || !could) break;  // :147
    }
    for (; !fused && k < num_iter; ++k) {
      rc = matvec_run(&pl, q(k), d.r, nullptr, nullptr, st);  // :108
      if (rc) return rc;
      LO_PROF_BEGIN("lz_sub_prev_dot", st);
      hipLaunchKernelGGL(k_lz_sub_prev_dot, grid, block, 0, st, d, k);
      LO_PROF_END(st);
      LO_PROF_BEGIN("lz_scal", st);
      hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 1, k, k, 0, 0);  // alpha_k -> t[k,k]  (:109-111)
      LO_PROF_END(st);
      LO_LAUNCH_CHECK();
      if (k + 1 < num_iter) {  // :114
        LO_PROF_BEGIN("lz_axpy", st);
        hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 1, q(k), d.r, d.scal);  // r -= alpha q_k  (:116)
        LO_PROF_END(st);
        // full re-orthogonalisation (:118-120)
        multidot(d, k + 1);
        reduce(d, k + 1, 0);
        correct(d, k + 1);
        // normalise (||r||^2 partials come out of the correction kernel); beta_k -> t[k,k+1], t[k+1,k]  (:121-128);
        // the normalised vector is written straight into its place q_{k+1} (:145) and checked / re-orthogonalised there
        LO_PROF_BEGIN("lz_scal", st);
        hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, k, k + 1, 1, 1);
        LO_PROF_END(st);
        LO_PROF_BEGIN("lz_axpy", st);
        hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, d.r, q(k + 1), d.scal);
        LO_PROF_END(st);
        q_hi = std::max(q_hi, k + 1);
        LzDev dn = d;
        dn.r = q(k + 1);
        // inner products with the normalised r (:131)
        multidot(dn, k + 1);
        reduce(dn, k + 1, 1);
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
          correct(dn, k + 1);  // uses the coefficients just computed
          LO_PROF_BEGIN("lz_scal", st);
          hipLaunchKernelGGL(k_lz_scal, one, block, 0, st, d, 0, -1, -1, 0, 0);
          LO_PROF_END(st);
          LO_PROF_BEGIN("lz_axpy", st);
          hipLaunchKernelGGL(k_lz_axpy, grid, block, 0, st, d, 0, dn.r, dn.r, d.scal);
          LO_PROF_END(st);
          multidot(dn, k + 1);
          reduce(dn, k + 1, 1);
          LO_LAUNCH_CHECK();
          LO_HIP_CHECK(hipMemcpyAsync(&h, d.ctrl, sizeof(LzCtrl), hipMemcpyDeviceToHost, st));
          LO_HIP_CHECK(hipStreamSynchronize(st));
        }
        if (all_small || !could) break;  // :147
      }
    }
    if (k == num_iter) k = num_iter - 1;
  }
  if (q_hi + 1 < max_iter)
    LO_HIP_CHECK(hipMemsetAsync(q(q_hi + 1), 0, sizeof(float) * nv * (size_t)(max_iter - q_hi - 1), st));
  LO_HIP_CHECK(hipStreamSynchronize(st));
  *iters_out = k + 1;  // :151
  return LO_OK;
}

}  // extern "C"
