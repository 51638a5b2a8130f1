// lo_probes.hip -- the host-side glue of InvQuadLogdet around the solves, as kernels (round 4).
//
// lo_probe_vectors_f32: the probe vectors of InvQuadLogdet.forward (functions/_inv_quad_logdet.py:91-110, :131) for the
// preconditioner P = L L^T + D of an AddedDiagLinearOperator.  The reference draws them with
// precond_lt.zero_mean_mvn_samples (sum_linear_operator.py:88-91 over root / diag samples: z = L e1 + sqrt(d) o e2),
// takes their column norms, divides, and concatenates them with the inv_quad right-hand side.  With torch that is a
// skinny batched GEMM (rocBLAS: 0.3 - 0.4 ms for 512 x 8192 x 15 x 16), an addcmul, a norm, a division and a cat:
// 1.0 - 1.1 ms of glue in front of a 6 ms forward.  Here: one pass that forms z into the first P columns of the
// right-hand-side block with per-tile partial sums of squares, one pass that divides by the norms and copies the
// inv_quad columns behind them.
//   bound: HBM -- e2 and L in, the [B, N, P + q] block out, then read and written once more.
//
// lo_iql_backward_factors_f32: the element-wise part of InvQuadLogdet.backward (:183-213): the left / right factors of
// the operator's bilinear derivative and of the preconditioner's, from the solves and the preconditioned probes.
#include <algorithm>

#include "lo_device.h"
#include "lo_internal.h"

namespace lo {

constexpr int PV_ROWS = 256;  // rows of a member per workgroup
constexpr int PV_MLP = 8;     // loads a thread keeps in flight in the flat walks (4 measured the same: not latency-bound)

// Flat walk over a tile's [rows, ld] elements by 256 threads: element e = t + 256 u sits at (row, col); both advance
// incrementally (no division in the loop).
struct FlatIdx {
  int row, col, drow, dcol, ld;
  __device__ __forceinline__ FlatIdx(int t, int ld_) : ld(ld_) {
    row = t / ld_;
    col = t - row * ld_;
    drow = kThreads / ld_;
    dcol = kThreads - drow * ld_;
  }
  __device__ __forceinline__ void next() {
    row += drow;
    col += dcol;
    if (col >= ld) {
      col -= ld;
      ++row;
    }
  }
};

// z = L e1 + sqrt(d) o e2 -> out[:, :P]; part[b, s, p] = sum over the tile's rows of z^2.
// The e2 tile comes in with flat coalesced loads and is parked in LDS (row stride P + 1: conflict-free for a thread per
// row); a thread then owns ONE row -- its k values of L stay in registers for all P columns, e1 is read from LDS as
// broadcasts -- and leaves z in the same LDS row; the tile goes out with flat coalesced stores.  (A thread per element
// re-read the row of L for every column: 535 us for 512 x 8192 x 16; a thread per row reading e2 and writing z in
// global memory touched 64 cache lines per wave instruction: 566 us.)
__global__ __launch_bounds__(kThreads) void k_probe_form(const float* __restrict__ L, int64_t l_sb, int64_t l_sn,
                                                         int64_t l_sk, int k, const float* __restrict__ d, int diag_mode,
                                                         const float* __restrict__ e1, const float* __restrict__ e2,
                                                         int N, int P, int ldo, float* __restrict__ out,
                                                         float* __restrict__ part) {
  extern __shared__ float pv_lds[];  // e1 [k, P] | tile [256, P + 1]
  __shared__ float red[4][64];
  float* e1_s = pv_lds;
  float* tile = pv_lds + ((k * P + 3) & ~3);
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r0 = s * PV_ROWS, nr = min(N - r0, PV_ROWS);
  const int ldt = P + 1;
  for (int i = t; i < k * P; i += kThreads) e1_s[i] = e1[(size_t)b * k * P + i];
  {
    const float* src = e2 + ((size_t)b * N + r0) * P;
    const int total = nr * P;
    FlatIdx f(t, P);
    for (int e0 = t; e0 < total; e0 += PV_MLP * kThreads) {  // PV_MLP loads in flight per thread
      float v[PV_MLP];
      int at[PV_MLP];
#pragma unroll
      for (int u = 0; u < PV_MLP; ++u) {
        const int e = e0 + u * kThreads;
        v[u] = (e < total) ? src[e] : 0.f;
        at[u] = f.row * ldt + f.col;
        f.next();
      }
#pragma unroll
      for (int u = 0; u < PV_MLP; ++u)
        if (e0 + u * kThreads < total) tile[at[u]] = v[u];
    }
  }
  const int row = r0 + t;
  const bool ok = t < nr;
  float lr[32];
  const float* Lr = L + (size_t)b * l_sb + (size_t)(ok ? row : 0) * l_sn;
#pragma unroll
  for (int j = 0; j < 32; ++j) lr[j] = (ok && j < k) ? Lr[(size_t)j * l_sk] : 0.f;
  const float sq = ok ? sqrtf((diag_mode == LO_DIAG_FULL) ? d[(size_t)b * N + row] : d[b]) : 0.f;
  __syncthreads();
  float* my = tile + t * ldt;
  for (int p0 = 0; p0 < P; p0 += 16) {
    float z[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = (ok && p0 + i < P) ? sq * my[p0 + i] : 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j < k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = fmaf(lr[j], (p0 + i < P) ? e1_s[j * P + p0 + i] : 0.f, z[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (ok && p0 + i < P) my[p0 + i] = z[i];
      const float tot = wave_sum_fast(z[i] * z[i]);
      if (lane == 0 && p0 + i < P) red[wave][p0 + i] = tot;
    }
  }
  __syncthreads();
  if (t < P) part[((size_t)b * S + s) * P + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  {
    float* dst = out + ((size_t)b * N + r0) * ldo;
    const int total = nr * P;
    FlatIdx f(t, P);
    for (int e = t; e < total; e += kThreads) {
      dst[f.row * ldo + f.col] = tile[f.row * ldt + f.col];
      f.next();
    }
  }
}

// out[:, :P] /= norm (norm = sqrt of the fixed-order sum of the partials); out[:, P:] = inv_quad_rhs; norms out.
// Flat, coalesced walk over the tile's [rows, P + q] elements, PV_MLP loads in flight per thread.
__global__ __launch_bounds__(kThreads) void k_probe_scale(const float* __restrict__ part, int S, int N, int P, int q,
                                                          const float* __restrict__ iq_rhs, float* __restrict__ out,
                                                          float* __restrict__ norms) {
  __shared__ float nrm_s[64];
  const int s = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x;
  if (t < P) {
    float tot = 0.f;
    for (int i = 0; i < S; ++i) tot += part[((size_t)b * S + i) * P + t];
    const float nr = sqrtf(tot);  // torch.linalg.vector_norm(probe_vectors, dim=-2)   :108
    nrm_s[t] = nr;
    if (s == 0) norms[(size_t)b * P + t] = nr;
  }
  __syncthreads();
  const int ldo = P + q;
  const int r0 = s * PV_ROWS, r1 = min(N, r0 + PV_ROWS);
  const int total = (r1 - r0) * ldo;
  float* ob = out + ((size_t)b * N + r0) * ldo;
  const float* ib = iq_rhs ? iq_rhs + ((size_t)b * N + r0) * q : nullptr;
  FlatIdx f(t, ldo);
  for (int e0 = t; e0 < total; e0 += PV_MLP * kThreads) {
    float v[PV_MLP];
    int col[PV_MLP];
#pragma unroll
    for (int u = 0; u < PV_MLP; ++u) {
      const int e = e0 + u * kThreads;
      col[u] = f.col;
      v[u] = 0.f;
      if (e < total) v[u] = (f.col < P) ? ob[e] : ib[f.row * q + (f.col - P)];
      f.next();
    }
#pragma unroll
    for (int u = 0; u < PV_MLP; ++u) {
      const int e = e0 + u * kThreads;
      if (e < total) ob[e] = (col[u] < P) ? v[u] / nrm_s[col[u]] : v[u];  // probe_vectors.div(probe_vector_norms)   :109
    }
  }
}

// InvQuadLogdet.backward (:183-213), element-wise part.  solves [B, N, P + q], pp [B, N, ldp] = P^-1 applied to the
// NORMALISED probes (first P columns used), norms [B, P], g_ld [B] (logdet grad), g_iq [B, q] (inv_quad grad or null).
//   left[:, :P]  = solves[:, :P] * norms * g_ld * coef        right[:, :P] = pp * norms      (= P^-1 of the raw probes)
//   left[:, P:]  = -solves[:, P:] * g_iq                      right[:, P:] = solves[:, P:]
//   pre_left     = -pp * norms * coef                         pre_right    = pp * norms * g_ld
__global__ __launch_bounds__(kThreads) void k_iql_factors(const float* __restrict__ solves, const float* __restrict__ pp,
                                                          int ldp, const float* __restrict__ norms,
                                                          const float* __restrict__ g_ld, const float* __restrict__ g_iq,
                                                          float coef, int N, int P, int q, float* __restrict__ left,
                                                          float* __restrict__ right, float* __restrict__ pre_left,
                                                          float* __restrict__ pre_right) {
  __shared__ float nrm_s[64], giq_s[64];
  const int s = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x;
  if (t < P) nrm_s[t] = norms[(size_t)b * P + t];
  if (t < q) giq_s[t] = g_iq ? g_iq[(size_t)b * q + t] : 0.f;
  __syncthreads();
  const float gl = g_ld[b];
  const int ldo = P + q;
  const int r0 = s * PV_ROWS, r1 = min(N, r0 + PV_ROWS);
  const int total = (r1 - r0) * ldo;
  const size_t base = ((size_t)b * N + r0);
  FlatIdx f(t, ldo);
  for (int e0 = t; e0 < total; e0 += PV_MLP * kThreads) {
    float sv[PV_MLP], pv[PV_MLP];
    int row[PV_MLP], col[PV_MLP];
#pragma unroll
    for (int u = 0; u < PV_MLP; ++u) {
      const int e = e0 + u * kThreads;
      row[u] = f.row;
      col[u] = f.col;
      sv[u] = (e < total) ? solves[base * ldo + e] : 0.f;
      pv[u] = (e < total && f.col < P) ? pp[(base + f.row) * ldp + f.col] : 0.f;
      f.next();
    }
#pragma unroll
    for (int u = 0; u < PV_MLP; ++u) {
      const int e = e0 + u * kThreads;
      if (e >= total) continue;
      if (col[u] < P) {
        const float ppv = pv[u] * nrm_s[col[u]];
        left[base * ldo + e] = sv[u] * (nrm_s[col[u]] * gl * coef);
        right[base * ldo + e] = ppv;
        pre_left[(base + row[u]) * P + col[u]] = -ppv * coef;
        pre_right[(base + row[u]) * P + col[u]] = ppv * gl;
      } else {
        left[base * ldo + e] = -sv[u] * giq_s[col[u] - P];
        right[base * ldo + e] = sv[u];
      }
    }
  }
}

}  // namespace lo

using namespace lo;

extern "C" {

size_t lo_probe_vectors_workspace_bytes(int64_t B, int64_t N, int64_t P) {
  const int64_t S = (N + PV_ROWS - 1) / PV_ROWS;
  return (size_t)B * S * P * sizeof(float) + 256;
}

int lo_probe_vectors_f32(const float* L, int64_t l_sb, int64_t l_sn, int64_t l_sk, int32_t k, const float* d,
                         int32_t diag_mode, const float* e1, const float* e2, const float* inv_quad_rhs, int64_t q,
                         int64_t B, int64_t N, int64_t P, float* rhs_out, float* norms, void* ws, size_t ws_bytes,
                         void* stream) {
  if (!L || !d || !e1 || !e2 || !rhs_out || !norms || !ws || B < 1 || N < 1 || (q > 0 && !inv_quad_rhs) || q < 0)
    return LO_ERR_BADARG;
  if (k < 1 || k > 32 || P < 1 || P > 64 || q > 64 || B > 65535 || N >= (1 << 30) / (P + q)) return LO_ERR_UNSUPPORTED;
  if (diag_mode != LO_DIAG_FULL && diag_mode != LO_DIAG_CONST) return LO_ERR_BADARG;
  if (ws_bytes < lo_probe_vectors_workspace_bytes(B, N, P)) return LO_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int S = (int)((N + PV_ROWS - 1) / PV_ROWS);
  float* part = reinterpret_cast<float*>(ws);
  dim3 grid(S, (unsigned)B), block(kThreads);
  LO_PROF_BEGIN("probe_form", st);
  const size_t lds = (size_t)((((int)k * (int)P + 3) & ~3) + PV_ROWS * ((int)P + 1)) * sizeof(float);
  hipLaunchKernelGGL(k_probe_form, grid, block, lds, st, L, l_sb, l_sn, l_sk, (int)k, d, (int)diag_mode, e1, e2, (int)N,
                     (int)P, (int)(P + q), rhs_out, part);
  LO_PROF_END(st);
  LO_PROF_BEGIN("probe_scale", st);
  hipLaunchKernelGGL(k_probe_scale, grid, block, 0, st, part, S, (int)N, (int)P, (int)q, inv_quad_rhs, rhs_out, norms);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

int lo_iql_backward_factors_f32(const float* solves, const float* pp, int64_t ldp, const float* norms, const float* g_ld,
                                const float* g_iq, float coef, int64_t B, int64_t N, int64_t P, int64_t q, float* left,
                                float* right, float* pre_left, float* pre_right, void* stream) {
  if (!solves || !pp || !norms || !g_ld || !left || !right || !pre_left || !pre_right || B < 1 || N < 1 || q < 0)
    return LO_ERR_BADARG;
  if (P < 1 || P > 64 || q > 64 || ldp < P || B > 65535 || N >= (1 << 30) / (P + q)) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int S = (int)((N + PV_ROWS - 1) / PV_ROWS);
  LO_PROF_BEGIN("iql_factors", st);
  hipLaunchKernelGGL(k_iql_factors, dim3(S, (unsigned)B), dim3(kThreads), 0, st, solves, pp, (int)ldp, norms, g_ld, g_iq,
                     coef, (int)N, (int)P, (int)q, left, right, pre_left, pre_right);
  LO_PROF_END(st);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // extern "C"
