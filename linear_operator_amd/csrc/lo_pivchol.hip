// lo_pivchol.hip -- greedy partial pivoted Cholesky, restating PivotedCholesky.forward
// (linear_operator/functions/_pivoted_cholesky.py:14-105) with the row fetch
// (utils/permutation.py:9-88 -> LinearOperator.__getitem__ -> operator _get_indices) generated on the fly
// from the operator descriptor:
//   Root/LowRankRoot: row[i] = sum_r C[p,r] C[i,r]          (root_linear_operator.py:37-50, diag :22-28)
//   Dense:            row[i] = K[p,i]                        (dense_linear_operator.py:47-50, diag :37-40)
//   Kron:             row[i] = K1[p1,i1] K2[p2,i2]           (kronecker_product_linear_operator.py:198-216)
// Integer results (pivots, permutation) must match the CPU path bit for bit, so every value that feeds
// a pivot decision is computed per element in a FIXED order with individually rounded operations
// (products rounded before summation, sequential in r and in j; this file is compiled with
// -ffp-contract=off AND carries the pragma below, because HIP's __fmul_rn/__fadd_rn are plain * and +
// that hipcc's default fp-contract=fast would fuse, and __fsqrt_rn is the 1-ulp native sqrt; sqrtf and /
// are correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt) -- the same
// order oracle/lo_oracle.py uses.  Batch-global control flow (one shared rank
// m, loop while max_b error > tol, :57) is decided on the device by a single-workgroup control kernel;
// the host enqueues all max_rank pivots and reads m back once.
//
// HBM traffic per pivot m and member: row source (C: 4NR B; dense: 4N; kron: ~0) + 4 m N (L rows 0..m-1)
// + ~24 N (diag, perm, L row m).
#include <algorithm>
#pragma clang fp contract(off)

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

struct PcCtrl {
  int stop;
  int m;  // pivots taken so far
};

struct PcDev {
  lo_op_desc op;
  int64_t B, N;
  int S, rows;     // position split
  int max_rank;
  float tol;
  float* diag;       // [B,N]
  float* L;          // [B,max_rank,N]
  long long* perm;   // [B,N]
  long long* pim;    // [B]
  float* maxval;     // [B]
  float* part_a;     // [B,S]  (max / error partials)
  float* part_b;     // [B,S]
  float* orig;       // [B]
  float* errors;     // [B]
  PcCtrl* ctrl;
};

__device__ __forceinline__ float seq_dot(const float* __restrict__ a, const float* __restrict__ b, int R) {
  // acc = a0*b0; acc = acc + a_r*b_r  (sequential, product rounded first)
  float acc = a[0] * b[0];
  for (int r = 1; r < R; ++r) acc = acc + a[r] * b[r];
  return acc;
}

__device__ __forceinline__ float src_diag(const PcDev& d, int64_t b, int i) {
  const lo_op_desc& op = d.op;
  if (op.kind == LO_OP_LOWRANK_DIAG) {
    const float* ci = op.A0 + ((size_t)b * d.N + i) * op.R;
    return seq_dot(ci, ci, (int)op.R);
  } else if (op.kind == LO_OP_DENSE_DIAG) {
    return op.A0[((size_t)b * d.N + i) * d.N + i];
  } else {
    const int n1 = (int)op.R, n2 = (int)op.n2;
    const int i1 = i / n2, i2 = i % n2;
    return op.A0[((size_t)b * n1 + i1) * n1 + i1] * op.A1[((size_t)b * n2 + i2) * n2 + i2];
  }
}

__global__ __launch_bounds__(kThreads) void k_pc_init(PcDev d) {
  __shared__ float red[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int j0 = s * d.rows, j1 = min((int)d.N, j0 + d.rows);
  float lmax = -INFINITY, lsum = 0.f;
  for (int i = j0 + threadIdx.x; i < j1; i += kThreads) {
    const float v = src_diag(d, b, i);
    d.diag[(size_t)b * d.N + i] = v;
    d.perm[(size_t)b * d.N + i] = i;
    lmax = fmaxf(lmax, v);
    lsum += fabsf(v);
  }
  const float m = block_max256(lmax, red);
  const float t = block_sum256(lsum, red);
  if (threadIdx.x == 0) {
    d.part_a[b * d.S + s] = m;
    d.part_b[b * d.S + s] = t;
  }
}

__global__ __launch_bounds__(kThreads) void k_pc_ctrl0(PcDev d) {
  for (int64_t b = threadIdx.x; b < d.B; b += kThreads) {
    float m = -INFINITY, t = 0.f;
    for (int s = 0; s < d.S; ++s) {
      m = fmaxf(m, d.part_a[b * d.S + s]);
      t += d.part_b[b * d.S + s];
    }
    d.orig[b] = m;           // orig_error = max(diag)                    :43
    d.errors[b] = t / m;     // ||diag||_1 / orig_error                    :44
  }
}

// decides whether pivot m is taken (loop condition :57) from the errors of pivot m-1
__global__ __launch_bounds__(kThreads) void k_pc_ctrl(PcDev d, int m) {
  if (d.ctrl->stop) return;
  __shared__ float red[kThreads];
  float lmax = -INFINITY, lnan = 0.f;
  if (m > 0) {
    for (int64_t b = threadIdx.x; b < d.B; b += kThreads) {
      float t = 0.f;
      for (int s = 0; s < d.S; ++s) t += d.part_b[b * d.S + s];
      const float e = t / d.orig[b];                                    // :99
      d.errors[b] = e;
      if (e != e) lnan = 1.f;
      lmax = fmaxf(lmax, e);
    }
    const float mx = block_max256(lmax, red);
    const float anynan = block_sum256(lnan, red);
    // torch.max propagates NaN and (NaN > tol) is False -> the reference stops
    const bool cont = (anynan == 0.f) && (mx > d.tol);
    if (!cont) {
      if (threadIdx.x == 0) d.ctrl->stop = 1;
      return;
    }
  }
  if (threadIdx.x == 0) d.ctrl->m = m + 1;
}

// argmax over the not-yet-pivoted positions, FIRST maximal position wins (torch.max on CPU, :61-63);
// swap permutation entries (:67-70); L[m, pi_m] = sqrt(max) (:73-74)
__global__ __launch_bounds__(kThreads) void k_pc_argmax(PcDev d, int m) {
  if (d.ctrl->stop) return;
  __shared__ float vbest[kThreads];
  __shared__ int jbest[kThreads];
  const int64_t b = blockIdx.x;
  const int N = (int)d.N;
  long long* perm = d.perm + (size_t)b * N;
  const float* diag = d.diag + (size_t)b * N;
  float bv = -INFINITY;
  int bj = 0x7fffffff;
  for (int j = m + threadIdx.x; j < N; j += kThreads) {
    const float v = diag[perm[j]];
    if (v > bv || (v == bv && j < bj) || bj == 0x7fffffff) {
      bv = v;
      bj = j;
    }
  }
  vbest[threadIdx.x] = bv;
  jbest[threadIdx.x] = bj;
  __syncthreads();
  for (int h = kThreads / 2; h >= 1; h >>= 1) {
    if (threadIdx.x < h) {
      const float ov = vbest[threadIdx.x + h];
      const int oj = jbest[threadIdx.x + h];
      const float mv = vbest[threadIdx.x];
      const int mj = jbest[threadIdx.x];
      if (oj != 0x7fffffff && (mj == 0x7fffffff || ov > mv || (ov == mv && oj < mj))) {
        vbest[threadIdx.x] = ov;
        jbest[threadIdx.x] = oj;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int j = jbest[0];
    const float v = vbest[0];
    const long long old = perm[m];
    const long long piv = perm[j];
    perm[m] = piv;
    perm[j] = old;
    d.pim[b] = piv;
    d.maxval[b] = v;
    d.L[((size_t)b * d.max_rank + m) * N + piv] = sqrtf(v);
  }
}

// Schur update of row m (:77-99) over the positions j > m of this workgroup's slice
__global__ __launch_bounds__(kThreads) void k_pc_update(PcDev d, int m) {
  if (d.ctrl->stop) return;
  extern __shared__ float sh[];  // [m] L[j][pi_m] | [R] C[pi_m,:]
  __shared__ float red[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int N = (int)d.N;
  const lo_op_desc& op = d.op;
  const long long* perm = d.perm + (size_t)b * N;
  float* diag = d.diag + (size_t)b * N;
  float* Lb = d.L + (size_t)b * d.max_rank * N;
  const int pim = (int)d.pim[b];
  const float piv = sqrtf(d.maxval[b]);
  float* upd = sh;
  float* crow = sh + m;
  for (int j = threadIdx.x; j < m; j += kThreads) upd[j] = Lb[(size_t)j * N + pim];
  const int R = (op.kind == LO_OP_LOWRANK_DIAG) ? (int)op.R : 0;
  for (int r = threadIdx.x; r < R; r += kThreads) crow[r] = op.A0[((size_t)b * N + pim) * R + r];
  __syncthreads();
  const int j0 = max(s * d.rows, m + 1), j1 = min(N, (s + 1) * d.rows);
  float lerr = 0.f;
  const int n1 = (int)op.R, n2 = (int)op.n2;
  for (int j = j0 + threadIdx.x; j < j1; j += kThreads) {
    const int i = (int)perm[j];
    float rowv;
    if (op.kind == LO_OP_LOWRANK_DIAG) {
      rowv = seq_dot(crow, op.A0 + ((size_t)b * N + i) * R, R);
    } else if (op.kind == LO_OP_DENSE_DIAG) {
      rowv = op.A0[((size_t)b * N + pim) * N + i];
    } else {
      const int p1 = pim / n2, p2 = pim % n2, i1 = i / n2, i2 = i % n2;
      rowv = op.A0[((size_t)b * n1 + p1) * n1 + i1] * op.A1[((size_t)b * n2 + p2) * n2 + i2];
    }
    float v = rowv;
    if (m > 0) {
      float acc = upd[0] * Lb[i];
      for (int jj = 1; jj < m; ++jj) acc = acc + upd[jj] * Lb[(size_t)jj * N + i];  // :83-89
      v = rowv - acc;
    }
    v = v / piv;                                   // :91
    Lb[(size_t)m * N + i] = v;                               // :92
    const float dn = diag[i] - v * v;    // :94-95
    diag[i] = dn;
    lerr += fabsf(dn);
  }
  const float t = block_sum256(lerr, red);
  if (threadIdx.x == 0) d.part_b[b * d.S + s] = t;
}

static void pc_layout(const lo_op_desc* op, int max_rank, Arena& ar, PcDev* d) {
  const int64_t B = op->B, N = op->N;
  Split sp = choose_split(B, N, 1024);
  d->op = *op;
  d->B = B; d->N = N; d->S = sp.S; d->rows = sp.rows; d->max_rank = max_rank;
  d->ctrl = ar.take<PcCtrl>(1);
  d->diag = ar.take<float>((size_t)B * N);
  d->pim = ar.take<long long>(B);
  d->maxval = ar.take<float>(B);
  d->part_a = ar.take<float>((size_t)B * sp.S);
  d->part_b = ar.take<float>((size_t)B * sp.S);
  d->orig = ar.take<float>(B);
  d->errors = ar.take<float>(B);
}

}  // namespace lo

using namespace lo;

extern "C" {

size_t lo_pivoted_cholesky_workspace_bytes(const lo_op_desc* op, int32_t max_rank) {
  if (!op) return 0;
  Arena ar(nullptr, 0);
  PcDev d;
  pc_layout(op, max_rank, ar, &d);
  return ar.off + 1024;
}

int lo_pivoted_cholesky_f32(const lo_op_desc* op, int32_t max_rank, float error_tol, float* L_rows, int64_t* perm,
                            int32_t* rank_out, void* ws, size_t ws_bytes, void* stream) {
  if (!op || !L_rows || !perm || !rank_out || !ws || max_rank < 1) return LO_ERR_BADARG;
  if (op->kind != LO_OP_LOWRANK_DIAG && op->kind != LO_OP_DENSE_DIAG && op->kind != LO_OP_KRON_DIAG)
    return LO_ERR_UNSUPPORTED;
  if (op->N > 0x7ffffff0) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = op->B, N = op->N;
  const int rank = (int)std::min<int64_t>(max_rank, N);  // :33
  Arena ar(ws, ws_bytes);
  PcDev d;
  pc_layout(op, max_rank, ar, &d);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  d.tol = error_tol;
  d.L = L_rows;
  d.perm = (long long*)perm;
  LO_HIP_CHECK(hipMemsetAsync(d.ctrl, 0, sizeof(PcCtrl), st));
  LO_HIP_CHECK(hipMemsetAsync(L_rows, 0, sizeof(float) * (size_t)B * max_rank * N, st));  // L = zeros :36-42
  dim3 grid(d.S, (unsigned)B), block(kThreads);
  hipLaunchKernelGGL(k_pc_init, grid, block, 0, st, d);
  hipLaunchKernelGGL(k_pc_ctrl0, dim3(1), block, 0, st, d);
  LO_LAUNCH_CHECK();
  const size_t R = (op->kind == LO_OP_LOWRANK_DIAG) ? (size_t)op->R : 0;
  for (int m = 0; m < rank; ++m) {
    hipLaunchKernelGGL(k_pc_ctrl, dim3(1), block, 0, st, d, m);
    LO_PROF_BEGIN("pc_argmax", st);
    hipLaunchKernelGGL(k_pc_argmax, dim3((unsigned)B), block, 0, st, d, m);
    LO_PROF_END(st);
    if (m + 1 < N) {  // :77
      LO_PROF_BEGIN("pc_update", st);
      hipLaunchKernelGGL(k_pc_update, grid, block, (m + R) * sizeof(float), st, d, m);
      LO_PROF_END(st);
    }
    LO_LAUNCH_CHECK();
  }
  PcCtrl h;
  LO_HIP_CHECK(hipMemcpyAsync(&h, d.ctrl, sizeof(PcCtrl), hipMemcpyDeviceToHost, st));
  LO_HIP_CHECK(hipStreamSynchronize(st));
  *rank_out = h.m;
  return LO_OK;
}

}  // extern "C"
