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
#include <string.h>

#include "lo_internal.h"

namespace lo {

struct PcCtrl {
  int stop;
  int m;  // pivots taken so far
};

template <typename T>
struct PcDevT {
  lo_op_desc op;
  // the terms whose rows / diagonals are summed left to right (SumLinearOperator._diagonal / _getitem of the
  // reference, sum_linear_operator.py:31-45): one entry for a plain operator
  int nterms;
  lo_op_desc terms[LO_MAX_TERMS];
  int64_t B, N;
  int S, rows;     // position split
  int max_rank;
  T tol;
  T* diag;       // [B,N]
  T* L;          // [B,max_rank,N]
  long long* perm;   // [B,N]
  long long* pim;    // [B]
  T* maxval;     // [B]
  T* part_a;     // [B,S]  (max / error partials)
  T* part_b;     // [B,S]
  T* orig;       // [B]
  T* errors;     // [B]
  T* arg_v;      // [B,S] per-slice argmax partial (value)
  int* arg_j;        // [B,S]                          (position)
  PcCtrl* ctrl;
};

// scalar-type helpers of the templated engine (float: the hot path; double: the reference is dtype-generic,
// functions/_pivoted_cholesky.py:14-105 -- same operation order, so pivots agree with a float64 run of the reference)
template <typename T> __device__ __forceinline__ T pc_sqrt(T x);
template <> __device__ __forceinline__ float pc_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double pc_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T pc_abs(T x) { return x < T(0) ? -x : x; }
template <typename T> __device__ __forceinline__ T pc_max(T a, T b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
template <typename T> __device__ __forceinline__ T pc_ninf() { return -(T)INFINITY; }
template <typename T> __device__ __forceinline__ const T* pc_ptr(const float* p) { return reinterpret_cast<const T*>(p); }
// block reductions over 256 threads through `red` (256 elements of T): fixed tree order, result in every thread
__device__ __forceinline__ float pc_block_sum(float v, float* red) { return block_sum256(v, red); }  // (round 1-3 order)
__device__ __forceinline__ float pc_block_max(float v, float* red) { return block_max256(v, red); }
template <typename T>
__device__ __forceinline__ T pc_block_sum(T v, T* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int h = kThreads / 2; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + h];
    __syncthreads();
  }
  return red[0];
}
template <typename T>
__device__ __forceinline__ T pc_block_max(T v, T* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int h = kThreads / 2; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] = pc_max(red[threadIdx.x], red[threadIdx.x + h]);
    __syncthreads();
  }
  return red[0];
}

template <typename T>
__device__ __forceinline__ T seq_dot(const T* __restrict__ a, const T* __restrict__ b, int R) {
  // acc = a0*b0; acc = acc + a_r*b_r  (sequential, product rounded first)
  T acc = a[0] * b[0];
  for (int r = 1; r < R; ++r) acc = acc + a[r] * b[r];
  return acc;
}

template <typename T>
__device__ __forceinline__ T src_diag(const PcDevT<T>& d, const lo_op_desc& op, int64_t b, int i) {
  const T* A0 = pc_ptr<T>(op.A0);
  const T* A1 = pc_ptr<T>(op.A1);
  if (op.kind == LO_OP_LOWRANK_DIAG) {
    const T* ci = A0 + ((size_t)b * d.N + i) * op.R;
    return seq_dot(ci, ci, (int)op.R);
  } else if (op.kind == LO_OP_DENSE_DIAG) {
    return A0[((size_t)b * d.N + i) * d.N + i];
  } else if (op.kind == LO_OP_CALLBACK) {  // generic operator: A1 = matrix._diagonal(), A0 = the fetched pivot rows
    return A1[(size_t)b * d.N + i];
  } else {
    const int n1 = (int)op.R, n2 = (int)op.n2;
    const int i1 = i / n2, i2 = i % n2;
    return A0[((size_t)b * n1 + i1) * n1 + i1] * A1[((size_t)b * n2 + i2) * n2 + i2];
  }
}

// Stage the C rows of `np` positions (rows i = perm[j0 + t], or i = j0 + t when perm == nullptr) into LDS with row
// stride R + 1 (conflict-free when every thread then walks its own row), coalesced: consecutive lanes read
// consecutive floats (float4 when R % 4 == 0) of consecutive rows.
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ Cb, int R, const long long* __restrict__ perm,
                                           int j0, int np, T* __restrict__ tile) {
  const int ld = R + 1;
  if constexpr (sizeof(T) != 4) {
    for (int e = threadIdx.x; e < np * R; e += kThreads) {
      const int pos = e / R, r = e % R;
      const int i = perm ? (int)perm[j0 + pos] : j0 + pos;
      tile[pos * ld + r] = Cb[(size_t)i * R + r];
    }
  } else if ((R & 3) == 0) {
    const int RQ = R >> 2;
    for (int e = threadIdx.x; e < np * RQ; e += kThreads) {
      const int pos = e / RQ, q = e % RQ;
      const int i = perm ? (int)perm[j0 + pos] : j0 + pos;
      const float4 v = *reinterpret_cast<const float4*>(Cb + (size_t)i * R + 4 * q);
      T* t = tile + pos * ld + 4 * q;
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  } else {
    for (int e = threadIdx.x; e < np * R; e += kThreads) {
      const int pos = e / R, r = e % R;
      const int i = perm ? (int)perm[j0 + pos] : j0 + pos;
      tile[pos * ld + r] = Cb[(size_t)i * R + r];
    }
  }
}

template <typename T>
__device__ __forceinline__ int tile_positions(int R) {
  // LDS budget ~48 KiB for the tile: 256 positions for R <= 47 (float), fewer for fat roots
  const int tp = (int)(49152 / sizeof(T)) / (R + 1);
  return tp >= kThreads ? kThreads : (tp < 1 ? 1 : tp);
}

// per-slice argmax partial: best (value, position) with the FIRST maximal position winning
template <typename T>
__device__ __forceinline__ void slice_argmax(T bv, int bj, T* vbest, int* jbest, T* out_v, int* out_j) {
  vbest[threadIdx.x] = bv;
  jbest[threadIdx.x] = bj;
  __syncthreads();
  for (int h = kThreads / 2; h >= 1; h >>= 1) {
    if (threadIdx.x < h) {
      const T ov = vbest[threadIdx.x + h];
      const int oj = jbest[threadIdx.x + h];
      const T mv = vbest[threadIdx.x];
      const int mj = jbest[threadIdx.x];
      if (oj != 0x7fffffff && (mj == 0x7fffffff || ov > mv || (ov == mv && oj < mj))) {
        vbest[threadIdx.x] = ov;
        jbest[threadIdx.x] = oj;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *out_v = vbest[0];
    *out_j = jbest[0];
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_pc_init(PcDevT<T> d) {
  extern __shared__ unsigned char pc_dyn_smem[];
  T* sh = reinterpret_cast<T*>(pc_dyn_smem);
  __shared__ T red[kThreads];
  __shared__ T vbest[kThreads];
  __shared__ int jbest[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int N = (int)d.N;
  const int j0 = s * d.rows, j1 = min(N, j0 + d.rows);
  // positions are walked in tiles small enough for the fattest low-rank term's LDS tile
  int tp = kThreads;
  for (int it = 0; it < d.nterms; ++it)
    if (d.terms[it].kind == LO_OP_LOWRANK_DIAG) tp = min(tp, tile_positions<T>((int)d.terms[it].R));
  T lmax = pc_ninf<T>(), lsum = T(0);
  T bv = pc_ninf<T>();
  int bj = 0x7fffffff;
  for (int t0 = j0; t0 < j1; t0 += tp) {
    const int np = min(tp, j1 - t0);
    const int i = t0 + threadIdx.x;
    T v = T(0);
    for (int it = 0; it < d.nterms; ++it) {
      const lo_op_desc& op = d.terms[it];
      T tv = T(0);
      if (op.kind == LO_OP_LOWRANK_DIAG) {
        const int R = (int)op.R;
        __syncthreads();
        stage_rows(pc_ptr<T>(op.A0) + (size_t)b * N * R, R, nullptr, t0, np, sh);
        __syncthreads();
        if ((int)threadIdx.x < np) {
          const T* row = sh + threadIdx.x * (R + 1);
          tv = seq_dot(row, row, R);                   // (root ** 2).sum(-1), root_linear_operator.py:22-28
        }
      } else if ((int)threadIdx.x < np) {
        tv = src_diag(d, op, b, i);
      }
      v = (it == 0) ? tv : v + tv;                     // sum(op._diagonal() for op in linear_ops), left to right
    }
    if ((int)threadIdx.x < np) {
      d.diag[(size_t)b * N + i] = v;
      d.perm[(size_t)b * N + i] = i;
      lmax = pc_max(lmax, v);
      lsum += pc_abs(v);
      if (v > bv || bj == 0x7fffffff) {  // positions increase with the loop: ties keep the earlier one
        bv = v;
        bj = i;
      }
    }
  }
  const T m = pc_block_max(lmax, red);
  const T t = pc_block_sum(lsum, red);
  if (threadIdx.x == 0) {
    d.part_a[b * d.S + s] = m;
    d.part_b[b * d.S + s] = t;
  }
  slice_argmax(bv, bj, vbest, jbest, &d.arg_v[b * d.S + s], &d.arg_j[b * d.S + s]);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_pc_ctrl0(PcDevT<T> d) {
  for (int64_t b = threadIdx.x; b < d.B; b += kThreads) {
    T m = pc_ninf<T>(), t = T(0);
    for (int s = 0; s < d.S; ++s) {
      m = pc_max(m, d.part_a[b * d.S + s]);
      t += d.part_b[b * d.S + s];
    }
    d.orig[b] = m;           // orig_error = max(diag)                    :43
    d.errors[b] = t / m;     // ||diag||_1 / orig_error                    :44
  }
}

// Decides whether pivot m is taken (loop condition :57) from the errors of pivot m-1, then finishes the argmax over
// the not-yet-pivoted positions from the per-slice partials (FIRST maximal position wins: torch.max on CPU, :61-63),
// swaps the permutation entries (:67-70) and sets L[m, pi_m] = sqrt(max) (:73-74).
template <typename T>
__global__ __launch_bounds__(kThreads) void k_pc_ctrl(PcDevT<T> d, int m) {
  if (d.ctrl->stop) return;
  __shared__ T red[kThreads];
  const int N = (int)d.N;
  if (m > 0) {
    T lmax = pc_ninf<T>(), lnan = T(0);
    for (int64_t b = threadIdx.x; b < d.B; b += kThreads) {
      T t = T(0);
      for (int s = 0; s < d.S; ++s) t += d.part_b[b * d.S + s];
      const T e = t / d.orig[b];                                    // :99
      d.errors[b] = e;
      if (e != e) lnan = T(1);
      lmax = pc_max(lmax, e);
    }
    const T mx = pc_block_max(lmax, red);
    const T anynan = pc_block_sum(lnan, red);
    // torch.max propagates NaN and (NaN > tol) is False -> the reference stops
    const bool cont = (anynan == T(0)) && (mx > d.tol);
    if (!cont) {
      if (threadIdx.x == 0) d.ctrl->stop = 1;
      return;
    }
  }
  if (threadIdx.x == 0) d.ctrl->m = m + 1;
  for (int64_t b = threadIdx.x; b < d.B; b += kThreads) {
    T bv = pc_ninf<T>();
    int bj = 0x7fffffff;
    for (int s = 0; s < d.S; ++s) {
      const T v = d.arg_v[b * d.S + s];
      const int j = d.arg_j[b * d.S + s];
      if (j != 0x7fffffff && (bj == 0x7fffffff || v > bv || (v == bv && j < bj))) {
        bv = v;
        bj = j;
      }
    }
    long long* perm = d.perm + (size_t)b * N;
    const long long old = perm[m];
    const long long piv = perm[bj];
    perm[m] = piv;
    perm[bj] = old;
    d.pim[b] = piv;
    d.maxval[b] = bv;
    d.L[((size_t)b * d.max_rank + m) * N + piv] = pc_sqrt(bv);
  }
}

// Schur update of row m (:77-99) over the positions j > m of this workgroup's slice, plus the slice's argmax
// partial for pivot m + 1.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_pc_update(PcDevT<T> d, int m) {
  if (d.ctrl->stop) return;
  extern __shared__ unsigned char pc_dyn_smem[];  // [max_rank] L[j][pi_m] | [R] C[pi_m,:] | row tile
  T* sh = reinterpret_cast<T*>(pc_dyn_smem);
  __shared__ T red[kThreads];
  __shared__ T vbest[kThreads];
  __shared__ int jbest[kThreads];
  const int s = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int N = (int)d.N;
  const long long* perm = d.perm + (size_t)b * N;
  T* diag = d.diag + (size_t)b * N;
  T* Lb = d.L + (size_t)b * d.max_rank * N;
  const int pim = (int)d.pim[b];
  const T piv = pc_sqrt(d.maxval[b]);
  // shared memory: [max_rank] L[j][pi_m] | per low-rank term its row C[pi_m,:] | one row tile (reused by the terms)
  T* upd = sh;
  T* crow = sh + d.max_rank;
  int Rtot = 0, tp = kThreads;
  for (int it = 0; it < d.nterms; ++it)
    if (d.terms[it].kind == LO_OP_LOWRANK_DIAG) {
      Rtot += (int)d.terms[it].R;
      tp = min(tp, tile_positions<T>((int)d.terms[it].R));
    }
  T* tile = crow + Rtot;
  for (int j = threadIdx.x; j < m; j += kThreads) upd[j] = Lb[(size_t)j * N + pim];
  {
    int off = 0;
    for (int it = 0; it < d.nterms; ++it) {
      const lo_op_desc& tm = d.terms[it];
      if (tm.kind != LO_OP_LOWRANK_DIAG) continue;
      const int R = (int)tm.R;
      for (int r = threadIdx.x; r < R; r += kThreads) crow[off + r] = pc_ptr<T>(tm.A0)[((size_t)b * N + pim) * R + r];
      off += R;
    }
  }
  __syncthreads();
  const int j0 = max(s * d.rows, m + 1), j1 = min(N, (s + 1) * d.rows);
  T lerr = T(0);
  T bv = pc_ninf<T>();
  int bj = 0x7fffffff;
  for (int t0 = j0; t0 < j1; t0 += tp) {
    const int np = min(tp, j1 - t0);
    const bool live = (int)threadIdx.x < np;
    const int j = t0 + threadIdx.x;
    const int i = live ? (int)perm[j] : 0;
    T rowv = T(0);
    int off = 0;
    for (int it = 0; it < d.nterms; ++it) {   // row pi_m of the sum = sum of the terms' rows, left to right
      const lo_op_desc& tm = d.terms[it];
      T tv = T(0);
      if (tm.kind == LO_OP_LOWRANK_DIAG) {
        const int R = (int)tm.R;
        __syncthreads();
        stage_rows(pc_ptr<T>(tm.A0) + (size_t)b * N * R, R, perm, t0, np, tile);
        __syncthreads();
        if (live) tv = seq_dot(crow + off, tile + threadIdx.x * (R + 1), R);
        off += R;
      } else if (live) {
        if (tm.kind == LO_OP_DENSE_DIAG) {
          tv = pc_ptr<T>(tm.A0)[((size_t)b * N + pim) * N + i];
        } else if (tm.kind == LO_OP_CALLBACK) {
          tv = pc_ptr<T>(tm.A0)[(size_t)b * N + i];  // row pi_m of this member, fetched by the host callback for this pivot
        } else {
          const int n1 = (int)tm.R, n2 = (int)tm.n2;
          const int p1 = pim / n2, p2 = pim % n2, i1 = i / n2, i2 = i % n2;
          tv = pc_ptr<T>(tm.A0)[((size_t)b * n1 + p1) * n1 + i1] * pc_ptr<T>(tm.A1)[((size_t)b * n2 + p2) * n2 + i2];
        }
      }
      rowv = (it == 0) ? tv : rowv + tv;
    }
    if (live) {
      T v = rowv;
      if (m > 0) {
        // :83-89, products and sums in the reference's order; the loads of up to sixteen earlier columns are issued
        // together (a load per trip, each waited for before the next, made the update a chain of HBM latencies)
        T acc = T(0);
        for (int j8 = 0; j8 < m; j8 += 16) {
          T lv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) lv[u] = (j8 + u < m) ? Lb[(size_t)(j8 + u) * N + i] : T(0);
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (j8 + u < m) acc = (j8 + u == 0) ? upd[0] * lv[0] : acc + upd[j8 + u] * lv[u];
        }
        v = rowv - acc;
      }
      v = v / piv;                                   // :91
      Lb[(size_t)m * N + i] = v;                     // :92
      const T dn = diag[i] - v * v;              // :94-95
      diag[i] = dn;
      lerr += pc_abs(dn);
      if (dn > bv || bj == 0x7fffffff) {
        bv = dn;
        bj = j;
      }
    }
  }
  const T t = pc_block_sum(lerr, red);
  if (threadIdx.x == 0) d.part_b[b * d.S + s] = t;
  slice_argmax(bv, bj, vbest, jbest, &d.arg_v[b * d.S + s], &d.arg_j[b * d.S + s]);
}

template <typename T>
static void pc_layout(const lo_op_desc* op, int max_rank, Arena& ar, PcDevT<T>* d) {
  const int64_t B = op->B, N = op->N;
  Split sp = choose_split(B, N, 1024);
  d->op = *op;
  if (op->kind == LO_OP_SUM) {
    d->nterms = op->terms ? std::min<int>(std::max<int>(op->nterms, 0), LO_MAX_TERMS) : 0;
    for (int i = 0; i < d->nterms; ++i) d->terms[i] = op->terms[i];
  } else {
    d->nterms = 1;
    d->terms[0] = *op;
  }
  d->B = B; d->N = N; d->S = sp.S; d->rows = sp.rows; d->max_rank = max_rank;
  d->ctrl = ar.take<PcCtrl>(1);
  d->diag = ar.take<T>((size_t)B * N);
  d->pim = ar.take<long long>(B);
  d->maxval = ar.take<T>(B);
  d->part_a = ar.take<T>((size_t)B * sp.S);
  d->part_b = ar.take<T>((size_t)B * sp.S);
  d->orig = ar.take<T>(B);
  d->errors = ar.take<T>(B);
  d->arg_v = ar.take<T>((size_t)B * sp.S);
  d->arg_j = ar.take<int>((size_t)B * sp.S);
}

typedef int (*pc_rowfetch_any)(void* user, const int64_t* piv, void* rows, int64_t B, int64_t N, void* stream);

// streaming engine shared by the descriptor and the callback entry points, float and double
template <typename T>
static int pc_stream_t(const lo_op_desc* op, const T* diag0, pc_rowfetch_any row_cb, void* row_user, int32_t max_rank,
                       T error_tol, T* L_rows, int64_t* perm, int32_t* rank_out, void* ws, size_t ws_bytes,
                       hipStream_t st) {
  const int64_t B = op->B, N = op->N;
  const int rank = (int)std::min<int64_t>(max_rank, N);  // :33
  Arena ar(ws, ws_bytes);
  PcDevT<T> d;
  pc_layout(op, max_rank, ar, &d);
  if (row_cb) {  // generic operator: one "term" whose rows arrive through the callback
    T* rows = ar.take<T>((size_t)B * N);
    d.terms[0].A0 = reinterpret_cast<const float*>(rows);
    d.terms[0].A1 = reinterpret_cast<const float*>(diag0);
  }
  if (!ar.ok) return LO_ERR_WORKSPACE;
  d.tol = error_tol;
  d.L = L_rows;
  d.perm = (long long*)perm;
  LO_HIP_CHECK(hipMemsetAsync(d.ctrl, 0, sizeof(PcCtrl), st));
  LO_HIP_CHECK(hipMemsetAsync(L_rows, 0, sizeof(T) * (size_t)B * max_rank * N, st));  // L = zeros :36-42
  dim3 grid(d.S, (unsigned)B), block(kThreads);
  // shared memory: the pivot rows of all low-rank terms (R elements in total) + one row tile sized for the walk of the
  // fattest low-rank term (the kernels walk min over the terms of tile_positions(R_i) positions at a time)
  size_t R = 0, tile_elems = 0;
  {
    size_t tpmin = kThreads;
    for (int it = 0; it < d.nterms; ++it)
      if (d.terms[it].kind == LO_OP_LOWRANK_DIAG) {
        const size_t Ri = (size_t)d.terms[it].R;
        R += Ri;
        size_t tp = (49152 / sizeof(T)) / (Ri + 1);
        tp = tp >= (size_t)kThreads ? (size_t)kThreads : (tp < 1 ? 1 : tp);
        tpmin = std::min(tpmin, tp);
      }
    for (int it = 0; it < d.nterms; ++it)
      if (d.terms[it].kind == LO_OP_LOWRANK_DIAG)
        tile_elems = std::max(tile_elems, tpmin * ((size_t)d.terms[it].R + 1));
  }
  if ((tile_elems + R + max_rank) * sizeof(T) > 60000) return LO_ERR_UNSUPPORTED;
  LO_PROF_BEGIN("pc_init", st);
  hipLaunchKernelGGL((k_pc_init<T>), grid, block, tile_elems * sizeof(T), st, d);
  LO_PROF_END(st);
  hipLaunchKernelGGL((k_pc_ctrl0<T>), dim3(1), block, 0, st, d);
  LO_LAUNCH_CHECK();
  for (int m = 0; m < rank; ++m) {
    LO_PROF_BEGIN("pc_ctrl_argmax", st);
    hipLaunchKernelGGL((k_pc_ctrl<T>), dim3(1), block, 0, st, d, m);
    LO_PROF_END(st);
    if (m + 1 < N) {  // :77
      if (row_cb) {  // row = matrix[..., pi_m, :] (:81): the reference's generic __getitem__, as a callback
        if (row_cb(row_user, (const int64_t*)d.pim, const_cast<float*>(d.terms[0].A0), B, N, (void*)st))
          return LO_ERR_LAUNCH;
      }
      LO_PROF_BEGIN("pc_update", st);
      hipLaunchKernelGGL((k_pc_update<T>), grid, block, (max_rank + R + tile_elems) * sizeof(T), st, d, m);
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

int pc_stream(const lo_op_desc* op, const float* diag0, lo_rowfetch_cb row_cb, void* row_user, int32_t max_rank,
              float error_tol, float* L_rows, int64_t* perm, int32_t* rank_out, void* ws, size_t ws_bytes,
              hipStream_t st) {
  return pc_stream_t<float>(op, diag0, reinterpret_cast<pc_rowfetch_any>(row_cb), row_user, max_rank, error_tol, L_rows,
                            perm, rank_out, ws, ws_bytes, st);
}

template <typename T>
static size_t pc_ws_bytes(const lo_op_desc* op, int32_t max_rank, bool cb) {
  Arena ar(nullptr, 0);
  PcDevT<T> d;
  pc_layout(op, max_rank, ar, &d);
  if (cb) ar.take<T>((size_t)op->B * op->N);  // the fetched rows
  return ar.off + 1024;
}

static int pc_check_desc(const lo_op_desc* op) {
  if (op->kind == LO_OP_SUM) {
    if (op->nterms < 2 || op->nterms > LO_MAX_TERMS || !op->terms) return LO_ERR_BADARG;
    for (int i = 0; i < op->nterms; ++i) {
      const lo_op_desc& t = op->terms[i];
      if ((t.kind != LO_OP_LOWRANK_DIAG && t.kind != LO_OP_DENSE_DIAG && t.kind != LO_OP_KRON_DIAG) || t.B != op->B ||
          t.N != op->N)
        return LO_ERR_BADARG;
    }
  } else if (op->kind != LO_OP_LOWRANK_DIAG && op->kind != LO_OP_DENSE_DIAG && op->kind != LO_OP_KRON_DIAG) {
    return LO_ERR_UNSUPPORTED;
  }
  if (op->N > 0x7ffffff0) return LO_ERR_UNSUPPORTED;
  return LO_OK;
}

}  // namespace lo

using namespace lo;

extern "C" {

size_t lo_pivoted_cholesky_workspace_bytes(const lo_op_desc* op, int32_t max_rank) {
  if (!op) return 0;
  size_t need = std::max(pc_ws_bytes<float>(op, max_rank, false),
                         (pc_onchip_eligible(op, max_rank) ? pc_onchip_workspace_bytes(op, max_rank) : (size_t)0));
  if (pc_onchip_rows_eligible(op, max_rank)) need = std::max(need, pc_onchip_rows_workspace_bytes(op, max_rank));
  return need;
}

int lo_pivoted_cholesky_f32(const lo_op_desc* op, int32_t max_rank, float error_tol, float* L_rows, int64_t* perm,
                            int32_t* rank_out, void* ws, size_t ws_bytes, void* stream) {
  if (!op || !L_rows || !perm || !rank_out || !ws || max_rank < 1) return LO_ERR_BADARG;
  if (const int rc = pc_check_desc(op)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int rank = (int)std::min<int64_t>(max_rank, op->N);  // :33
  resident_tick();
  if (pc_onchip_eligible(op, max_rank)) {  // operator-resident fast path (lo_pivchol_onchip.hip), same results
    const int rc = pc_onchip_run(op, rank, max_rank, error_tol, L_rows, (long long*)perm, rank_out, ws, ws_bytes, st);
    if (rc != LO_ERR_LAUNCH) return rc;
    (void)hipGetLastError();  // exchange timed out (co-residency lost): redo with the streaming engine
  } else if (pc_onchip_rows_eligible(op, max_rank)) {  // dense / Kronecker rows, same results (k_pc_onchip_rows)
    const int rc = pc_onchip_rows_run(op, rank, max_rank, error_tol, L_rows, (long long*)perm, rank_out, ws, ws_bytes, st);
    if (rc != LO_ERR_LAUNCH) return rc;
    (void)hipGetLastError();
  }
  return pc_stream(op, nullptr, nullptr, nullptr, max_rank, error_tol, L_rows, perm, rank_out, ws, ws_bytes, st);
}

size_t lo_pivoted_cholesky_cb_workspace_bytes(int64_t B, int64_t N, int32_t max_rank) {
  lo_op_desc op;
  memset(&op, 0, sizeof(op));
  op.kind = LO_OP_CALLBACK; op.B = B; op.N = N;
  return pc_ws_bytes<float>(&op, max_rank, true);
}

int lo_pivoted_cholesky_cb_f32(int64_t B, int64_t N, const float* diag, lo_rowfetch_cb row_cb, void* row_user,
                               int32_t max_rank, float error_tol, float* L_rows, int64_t* perm, int32_t* rank_out,
                               void* ws, size_t ws_bytes, void* stream) {
  if (!diag || !row_cb || !L_rows || !perm || !rank_out || !ws || max_rank < 1 || B < 1 || N < 1) return LO_ERR_BADARG;
  if (N > 0x7ffffff0) return LO_ERR_UNSUPPORTED;
  lo_op_desc op;
  memset(&op, 0, sizeof(op));
  op.kind = LO_OP_CALLBACK; op.B = B; op.N = N;
  return pc_stream(&op, diag, row_cb, row_user, max_rank, error_tol, L_rows, perm, rank_out, ws, ws_bytes,
                   (hipStream_t)stream);
}

// ---- float64 (round 4): the reference's PivotedCholesky.forward is dtype-generic.  Same streaming engine, same
// operation order; the descriptor's pointers (A0, A1; d is ignored) are read as `const double*`.
size_t lo_pivoted_cholesky_f64_workspace_bytes(const lo_op_desc* op, int32_t max_rank) {
  return op ? pc_ws_bytes<double>(op, max_rank, false) : 0;
}

int lo_pivoted_cholesky_f64(const lo_op_desc* op, int32_t max_rank, double error_tol, double* L_rows, int64_t* perm,
                            int32_t* rank_out, void* ws, size_t ws_bytes, void* stream) {
  if (!op || !L_rows || !perm || !rank_out || !ws || max_rank < 1) return LO_ERR_BADARG;
  if (const int rc = pc_check_desc(op)) return rc;
  return pc_stream_t<double>(op, nullptr, nullptr, nullptr, max_rank, error_tol, L_rows, perm, rank_out, ws, ws_bytes,
                             (hipStream_t)stream);
}

size_t lo_pivoted_cholesky_cb_f64_workspace_bytes(int64_t B, int64_t N, int32_t max_rank) {
  lo_op_desc op;
  memset(&op, 0, sizeof(op));
  op.kind = LO_OP_CALLBACK; op.B = B; op.N = N;
  return pc_ws_bytes<double>(&op, max_rank, true);
}

int lo_pivoted_cholesky_cb_f64(int64_t B, int64_t N, const double* diag, lo_rowfetch_cb_f64 row_cb, void* row_user,
                               int32_t max_rank, double error_tol, double* L_rows, int64_t* perm, int32_t* rank_out,
                               void* ws, size_t ws_bytes, void* stream) {
  if (!diag || !row_cb || !L_rows || !perm || !rank_out || !ws || max_rank < 1 || B < 1 || N < 1) return LO_ERR_BADARG;
  if (N > 0x7ffffff0) return LO_ERR_UNSUPPORTED;
  lo_op_desc op;
  memset(&op, 0, sizeof(op));
  op.kind = LO_OP_CALLBACK; op.B = B; op.N = N;
  return pc_stream_t<double>(&op, diag, reinterpret_cast<pc_rowfetch_any>(row_cb), row_user, max_rank, error_tol, L_rows,
                             perm, rank_out, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
