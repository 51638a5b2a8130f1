// lo_solve_fused.hip -- host side of the fused end-to-end solve (kernel: lo_solve_fused_impl.h).
// C ABI: lo_solve_fused_f32 = PivotedCholesky.forward (functions/_pivoted_cholesky.py:14-105) +
// AddedDiagLinearOperator._init_cache (operators/added_diag_linear_operator.py:144-184) + linear_cg
// (utils/linear_cg.py:98-359) of an AddedDiag(LowRankRoot, Diag | ConstantDiag) operator in ONE resident launch.
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lo_solve_fused_impl.h"

namespace lo {

struct FusedCtrl {
  int status;       // LO_FUSED_* below
  int iterations;
  int tol_reached;
  int nan_detected;
  int skipped;
  int oc_err;       // a group exchange timed out
  int early;        // a member's own pivot error reached the tolerance (or NaN) before `rank` pivots
  float mean_resid;
};

// Batch-global decisions after the launch: linear_cg.py:302-308 at the floor, the NaN check (:199-200), the skip rule
// (:207-208); the pivot rule (_pivoted_cholesky.py:57) is exact iff no member raised `early`.
__global__ __launch_bounds__(kThreads) void k_fused_ctrl(FusedCtrl* ctrl, const float* __restrict__ resid_rec,
                                                          const int* __restrict__ init_conv, const int* __restrict__ flags,
                                                          const int* __restrict__ err, int64_t n, int iters, int max_iter,
                                                          float tol) {
  __shared__ float red[kThreads];
  float lsum = 0.f, lnan = 0.f, lnotconv = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const float rn = resid_rec[(size_t)(iters - 1) * n + i];
    const float r0 = resid_rec[i];
    lsum += rn;
    if (r0 != r0 || rn != rn) lnan = 1.f;
    if (!init_conv[i]) lnotconv = 1.f;
  }
  const float mean = block_sum256(lsum, red) / (float)n;
  const float anynan = block_sum256(lnan, red);
  const float notconv = block_sum256(lnotconv, red);
  if (threadIdx.x == 0) {
    const int k = iters - 1;
    ctrl->iterations = iters;
    ctrl->mean_resid = mean;
    ctrl->oc_err = *err;
    ctrl->early = flags[0];
    ctrl->tol_reached = ctrl->nan_detected = ctrl->skipped = 0;
    int status = LO_FUSED_CONTINUE;
    if (*err) status = LO_FUSED_TIMEOUT;
    else if (flags[0]) status = LO_FUSED_EARLY_STOP;
    else if (anynan > 0.f) {
      ctrl->nan_detected = 1;
      status = LO_FUSED_OK;
    } else if (notconv == 0.f) {
      ctrl->skipped = 1;
      ctrl->iterations = 0;
      status = LO_FUSED_OK;
    } else if (k >= min(10, max_iter - 1) && mean < tol) {
      ctrl->tol_reached = 1;
      status = LO_FUSED_OK;
    }
    ctrl->status = status;
  }
}

// swaps [B, rank] -> the reference's permutation [B, N] (int64): identity with the recorded exchanges applied in order
__global__ __launch_bounds__(kThreads) void k_fused_perm(const int* __restrict__ swaps, int rank, int N,
                                                          long long* __restrict__ perm) {
  const int64_t b = blockIdx.x;
  long long* pb = perm + (size_t)b * N;
  for (int i = threadIdx.x; i < N; i += kThreads) pb[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int m = 0; m < rank; ++m) {
      const int j = swaps[(size_t)b * rank + m];
      const long long x = pb[m];
      pb[m] = pb[j];
      pb[j] = x;
    }
  }
}

struct FuLayout {
  unsigned long long *pgbuf, *egbuf, *cgbuf;
  int* ints;  // err, flags, next_member, pad
  size_t zero_bytes;  // [pgbuf, ints + 4): cleared before every launch
  float *err_rec, *orig, *resid_rec;
  int* init_conv;
  FusedCtrl* ctrl;
  long long* dbg;
};

static constexpr int kFuMaxWgs = 1024;  // two workgroups per CU on up to 512 CUs

static FusedCtrl* pinned_ctrl() { return static_cast<FusedCtrl*>(pinned_status_block()); }

static void fu_layout(int64_t B, int rank, int iters, int64_t c, int nwgs, Arena& ar, FuLayout* l) {
  l->pgbuf = ar.take<unsigned long long>((size_t)nwgs * 2 * FU_SLOT);
  l->egbuf = ar.take<unsigned long long>((size_t)nwgs * FU_ESLOT);
  l->cgbuf = ar.take<unsigned long long>((size_t)nwgs * 2 * R4_SLOT);
  l->ints = ar.take<int>(8);
  l->zero_bytes = (size_t)(reinterpret_cast<char*>(l->ints + 8) - reinterpret_cast<char*>(l->pgbuf));
  l->err_rec = ar.take<float>((size_t)rank * B);
  l->orig = ar.take<float>((size_t)B);
  l->resid_rec = ar.take<float>((size_t)iters * B * c);
  l->init_conv = ar.take<int>((size_t)B * c);
  l->ctrl = ar.take<FusedCtrl>(1);
  l->dbg = ar.take<long long>(16);
}

static int fused_group_size(int64_t N) {
  for (int gw = 1; gw < 32; gw *= 2)
    if (N <= (int64_t)gw * R4_ROWS) return gw;
  return 32;
}

static int fused_iters(const lo_cg_params* prm) {
  const int fmi = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter;  // (linear_cg.py:303-305)
  return std::min(10, fmi - 1) + 1;
}

}  // namespace lo

using namespace lo;

extern "C" {

int lo_solve_fused_supported(const lo_op_desc* op, int32_t rank, const lo_cg_params* prm) {
  if (!op || !prm || resident_off() || getenv("LO_NO_FUSED_SOLVE")) return 0;
  if (op->kind != LO_OP_LOWRANK_DIAG || (op->diag_mode != LO_DIAG_FULL && op->diag_mode != LO_DIAG_CONST)) return 0;
  if (!(op->R == 8 || op->R == 16 || op->R == 32)) return 0;
  if (rank < 1 || rank > FU_MAXRANK || rank > op->N) return 0;
  // (groups of 1 .. 16 workgroups: up to 16384 rows; groups of 32 spill at the 256-VGPR budget)
  if (op->N < 256 || op->N > (int64_t)16 * R4_ROWS || op->B < 1 || op->B >= (1 << 24) - 1024) return 0;
  if (op->N > (int64_t)8 * R4_ROWS && getenv("LO_FUSED_NO_GW16")) return 0;
  if (prm->c < 1 || prm->c > 8 || prm->n_tridiag != 0 || prm->stop_reduce) return 0;
  // (debug switch of the w-recurrence: the single-column instantiation of this kernel always carries w by recurrence,
  //  like k_cg_onchip5 MODE 2 -- with the switch set, single-column solves take the three-launch path)
  if (prm->c == 1 && getenv("LO_OC_NO_WREC")) return 0;
  const int fmi = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter;
  if (prm->max_iter < 11 || fmi < 11) return 0;
  const int nwg = onchip_num_workgroups();
  if (nwg < 64 || 2 * nwg > kFuMaxWgs) return 0;
  return 1;
}

size_t lo_solve_fused_workspace_bytes(const lo_op_desc* op, int32_t rank, const lo_cg_params* prm) {
  if (!op || !prm) return 0;
  Arena ar(nullptr, 0);
  FuLayout l;
  fu_layout(op->B, rank, fused_iters(prm), prm->c, kFuMaxWgs, ar, &l);
  return ar.off + 1024;
}

int lo_solve_fused_f32(const lo_op_desc* op, int32_t rank, float error_tol, const lo_cg_params* prm, const float* rhs,
                       float* x, float* F, float* EF, float* E, float* dinv, float* logdet_p, int32_t* swaps,
                       void* ws, size_t ws_bytes, lo_fused_info* info, void* stream) {
  if (!op || !prm || !rhs || !x || !swaps || !ws || !info) return LO_ERR_BADARG;
  if (!lo_solve_fused_supported(op, rank, prm)) return LO_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = op->B, N = op->N;
  const int iters = fused_iters(prm);
  const int nwg = onchip_num_workgroups();
  Arena ar(ws, ws_bytes);
  FuLayout l;
  fu_layout(B, rank, iters, prm->c, 2 * nwg, ar, &l);
  if (!ar.ok) return LO_ERR_WORKSPACE;
  FusedArgs a;
  a.C = op->A0; a.d = op->d; a.d_mode = op->diag_mode;
  a.rhs = rhs; a.xout = x; a.c = (int)prm->c;
  a.B = B; a.N = (int)N;
  a.GW = (getenv("LO_OC_GW8") && N <= 8 * (int64_t)R4_ROWS) ? 8 : fused_group_size(N);
  a.RW = (int)((N + a.GW - 1) / a.GW);
  a.rank = rank; a.pc_tol = error_tol; a.iters = iters;
  a.eps = prm->eps; a.stop_after = prm->stop_updating_after;
  a.F = F; a.EF = EF; a.E = E; a.dinv = dinv; a.logdet_p = logdet_p;
  a.swaps = swaps; a.err_rec = l.err_rec; a.orig = l.orig;
  a.resid_rec = l.resid_rec; a.init_conv = l.init_conv;
  a.err = l.ints; a.flags = l.ints + 1; a.next_member = l.ints + 2;
  a.pgbuf = l.pgbuf; a.egbuf = l.egbuf; a.cgbuf = l.cgbuf;
  a.allow_l2_handoff = onchip_l2_handoff_allowed();
  const bool debug = getenv("LO_FU_DEBUG") != nullptr && B >= 8;
  a.dbg = debug ? l.dbg : nullptr;
  a.dbg_member = debug ? atoi(getenv("LO_FU_DEBUG")) : 0;
  {
    const int zrc = zero_span(l.pgbuf, l.zero_bytes, st);  // (one launch instead of the runtime's fill kernels)
    if (zrc) return zrc;
  }
  if (debug) LO_HIP_CHECK(hipMemsetAsync(l.dbg, 0, 16 * sizeof(long long), st));
  if (getenv("LO_OC_TEST_FALLBACK")) LO_HIP_CHECK(hipMemsetAsync(l.ints, 1, 1, st));  // as if an exchange had timed out
  int rc = LO_ERR_UNSUPPORTED;
  if (op->R == 32) rc = fused_launch_r32(a, nwg, st);
  else if (op->R == 16) rc = fused_launch_r16(a, nwg, st);
  else if (op->R == 8) rc = fused_launch_r8(a, nwg, st);
  if (rc) return rc;
  const int fmi = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter;
  FusedCtrl h;
  FusedCtrl* hp = pinned_ctrl();  // the decision lands in pinned host memory: no device-to-host copy command
  hipLaunchKernelGGL(k_fused_ctrl, dim3(1), dim3(kThreads), 0, st, hp ? hp : l.ctrl, l.resid_rec, l.init_conv, l.ints + 1,
                     l.ints, B * prm->c, iters, fmi, prm->tolerance);
  LO_LAUNCH_CHECK();
  if (hp) {
    LO_HIP_CHECK(hipStreamSynchronize(st));
    h = *hp;
  } else {
    LO_HIP_CHECK(hipMemcpyAsync(&h, l.ctrl, sizeof(h), hipMemcpyDeviceToHost, st));
    LO_HIP_CHECK(hipStreamSynchronize(st));
  }
  if (debug) {
    long long ts[16];
    LO_HIP_CHECK(hipMemcpy(ts, l.dbg, sizeof(ts), hipMemcpyDeviceToHost));
    fprintf(stderr, "solve_fused member %d (100 MHz ticks): load %lld E-partials %lld pivots %lld E-totals+M %lld algebra %lld "
            "(T,G %lld chol+Y %lld F,EF %lld) CG %lld\n", a.dbg_member, ts[1] - ts[0], ts[8] - ts[1], ts[2] - ts[8],
            ts[3] - ts[2], ts[4] - ts[3], ts[11] - ts[3], ts[12] - ts[11], ts[4] - ts[12], ts[5] - ts[4]);
  }
  info->status = h.status;
  info->iterations = h.iterations;
  info->matvecs = h.iterations;
  info->tolerance_reached = h.tol_reached;
  info->nan_detected = h.nan_detected;
  info->skipped = h.skipped;
  info->rank = rank;
  info->mean_residual = h.mean_resid;
  if (h.status == LO_FUSED_TIMEOUT) {
    fprintf(stderr, "liblo_amd: fused solve timed out in a group exchange, the caller falls back to the three-launch path\n");
    g_onchip_fused_timeouts++;
    onchip_note_timeout();
  }
  return LO_OK;
}

// the reference's permutation [B, N] from the swaps the fused solve recorded (tests; probe sampling needs L, not this)
int lo_solve_fused_perm(const int32_t* swaps, int64_t B, int64_t N, int32_t rank, int64_t* perm, void* stream) {
  if (!swaps || !perm) return LO_ERR_BADARG;
  hipLaunchKernelGGL(k_fused_perm, dim3((unsigned)B), dim3(kThreads), 0, (hipStream_t)stream, swaps, rank, (int)N,
                     (long long*)perm);
  LO_LAUNCH_CHECK();
  return LO_OK;
}

}  // extern "C"
