// lo_cg.hip -- batched modified preconditioned conjugate gradients on the device.
// Restates linear_operator/utils/linear_cg.py:98-359 (reference) as a launch sequence with NO host
// synchronisation inside the iteration: every scalar decision of the reference (masked alpha/beta,
// has_converged, the batch-global mean-residual stopping rule :302-308, the tridiagonal recurrence and
// its batch-global freeze :311-332, the NaN check :199-200) is taken by a single-workgroup control
// kernel that writes a sticky `stop` word; all later kernels of the solve read it and exit at once.
// The host launches the iterations the reference is guaranteed to run (the 11 / n_tridiag floors) in
// one go, then polls the control block every few iterations.
//
// per iteration k (vectors [B,N,c], partial sums [B,S,c] reduced in fixed order):
//   update_p   p = z + beta p                                   (linear_cg.py:46; z == r if no precond)
//   matvec     Ap = A p ;  pAp_part = sum_rows p o Ap           (:248, :250-251 fused into the matvec)
//   update_xr  alpha (masked, :254-260) ; r -= alpha Ap (:264) ; x += alpha p (:31) ; rr_part = sum r^2
//   precond    z = P^-1 r ; rz_part = sum r o z                 (:268, :35-36 fused)
//   ctrl       beta (:34-42), residual norms (:298-300), stop rule (:302-308), tridiag (:311-332)
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cmath>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "lo_device.h"
#include "lo_internal.h"
#include "lo_cg_onchip.h"

namespace lo {

bool g_onchip_disabled = false;
int g_onchip_fused_timeouts = 0;
// ---- the gate of the resident kernels (round 5; replaces the process-wide latch of rounds 2 - 4) ----
// A hand-off timeout (co-residency lost: another kernel -- RCCL's, another process' -- holds part of the CUs) sends the
// next `backoff` entry-point calls to the streaming engines, then the resident kernels are tried again ("re-armed").
// A timeout right after a re-arm doubles the cool-down (16, 32, ... 4096 calls: a persistent loss costs one ~0.5 s spin
// per cool-down, amortised to nothing); a clean resident solve resets it.  Everything is reported (lo_resident_status).
constexpr int kResidentBackoff0 = 16, kResidentBackoffMax = 4096;
static std::atomic<int> g_res_timeouts{0}, g_res_cooldown{0}, g_res_backoff{kResidentBackoff0}, g_res_rearms{0},
    g_res_inject{0};
bool resident_off() { return g_onchip_disabled || g_res_cooldown.load(std::memory_order_relaxed) > 0; }
void resident_tick() {  // once per entry-point call that could use a resident kernel
  int c = g_res_cooldown.load(std::memory_order_relaxed);
  while (c > 0 && !g_res_cooldown.compare_exchange_weak(c, c - 1, std::memory_order_relaxed)) {
  }
  if (c == 1) {
    g_res_rearms.fetch_add(1, std::memory_order_relaxed);
    fprintf(stderr, "liblo_amd: resident kernels re-armed after their cool-down\n");
  }
}
void resident_note_ok() { g_res_backoff.store(kResidentBackoff0, std::memory_order_relaxed); }
bool resident_take_injection() {
  int n = g_res_inject.load(std::memory_order_relaxed);
  while (n > 0 && !g_res_inject.compare_exchange_weak(n, n - 1, std::memory_order_relaxed)) {
  }
  return n > 0;
}
void* pinned_status_block() {
  static thread_local void* p = nullptr;
  if (!p) {
    if (hipHostMalloc(&p, 256, hipHostMallocCoherent) != hipSuccess) p = nullptr;
    else memset(p, 0, 256);
  }
  return p;
}
// Wait for the kernel that mirrors its decision into the pinned block: the block's last word carries a per-call ticket
// written AFTER the payload (system-scope fence in between), so the host can spin on it for a few microseconds instead
// of paying the wake-up latency of hipStreamSynchronize (~15 us on this stack); falls back to the stream
// synchronisation after ~200 us (long solves, other work queued on the stream in front of ours).
static unsigned next_ticket() {
  static thread_local unsigned t = 0;
  return ++t == 0 ? ++t : t;
}
static int wait_ticket(volatile unsigned* word, unsigned ticket, hipStream_t st) {
  // (time-bounded: ~2 ms of spinning covers every resident launch of the BASELINE shapes; the iteration count of a fixed
  //  spin loop depends on what `pause` costs on the host CPU)
  const auto t0 = std::chrono::steady_clock::now();
  for (int spin = 0;; ++spin) {
    if (*word == ticket) return LO_OK;
    __builtin_ia32_pause();
    if ((spin & 1023) == 1023 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3)
      break;
  }
  LO_HIP_CHECK(hipStreamSynchronize(st));
  return (*word == ticket) ? LO_OK : LO_ERR_LAUNCH;
}
void onchip_note_timeout() {
  if (getenv("LO_OC_TEST_FALLBACK")) return;
  g_res_timeouts.fetch_add(1, std::memory_order_relaxed);
  const int bo = g_res_backoff.load(std::memory_order_relaxed);
  g_res_cooldown.store(bo, std::memory_order_relaxed);
  g_res_backoff.store(std::min(2 * bo, kResidentBackoffMax), std::memory_order_relaxed);
  fprintf(stderr, "liblo_amd: a resident kernel lost its co-residency (hand-off timeout): the next %d calls run on the "
                  "streaming engines, then the resident kernels are tried again (lo_resident_status)\n", bo);
}
// The result-only ("lean") first pass is speculative: when the stop rule does not hold at the floor the launches are
// repeated with the continuation state.  An operator that missed is likely to miss again on its next solve (same tensors,
// same settings): remember up to 16 such (operator, shape, tolerance) signatures per thread and start those solves with
// the state-writing pass right away; a solve that then DOES stop at the floor clears its entry.
struct LeanMiss {
  const void* a0;
  const void* d;
  int64_t B, N, c;
  float tol;
  int gen;  // lo_precond_desc.generation of the cache the miss was seen with (0: none given): an address the allocator
            // hands out again for OTHER tensors does not inherit the entry
  int valid;
};
static thread_local LeanMiss tls_lean_miss[16] = {};
static thread_local int tls_lean_miss_next = 0;
static int lean_miss_find(const lo_op_desc* op, const lo_cg_params* prm, int gen) {
  for (int i = 0; i < 16; ++i) {
    const LeanMiss& m = tls_lean_miss[i];
    if (m.valid && m.a0 == op->A0 && m.d == op->d && m.B == op->B && m.N == op->N && m.c == prm->c &&
        m.tol == prm->tolerance && m.gen == gen)
      return i;
  }
  return -1;
}
static void lean_miss_note(const lo_op_desc* op, const lo_cg_params* prm, int gen) {
  if (lean_miss_find(op, prm, gen) >= 0) return;
  tls_lean_miss[tls_lean_miss_next] = LeanMiss{op->A0, op->d, op->B, op->N, prm->c, prm->tolerance, gen, 1};
  tls_lean_miss_next = (tls_lean_miss_next + 1) % 16;
}
static thread_local lo_cg_plan tls_last_exec = {};    // what the last lo_cg_solve_f32 of this thread actually launched
static thread_local bool tls_no_fused_precond = false;  // set while a solve is redone after a timed-out hand-off

// hipGraph of one CG iteration: captured on a private side stream, replayed on the caller's stream.
struct GraphReplay {
  hipStream_t side = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool capturing = false;
  bool begin() {
    if (!side && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) {
      side = nullptr;
      (void)hipGetLastError();
      return false;
    }
    if (hipStreamBeginCapture(side, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    capturing = true;
    return true;
  }
  bool end() {
    capturing = false;
    if (hipStreamEndCapture(side, &graph) != hipSuccess || !graph) {
      (void)hipGetLastError();
      graph = nullptr;
      return false;
    }
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      exec = nullptr;
      return false;
    }
    return true;
  }
  void reset() {
    if (capturing) {
      hipGraph_t g2 = nullptr;
      (void)hipStreamEndCapture(side, &g2);
      if (g2) (void)hipGraphDestroy(g2);
      capturing = false;
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    exec = nullptr;
    graph = nullptr;
  }
  ~GraphReplay() {
    reset();
    if (side) (void)hipStreamDestroy(side);
  }
};
// A chunk of columns goes to the column-lockstep kernel (16 at a time on the matrix cores) when it has at least this
// many live columns; fewer are cheaper one after the other on the second-generation kernel.
constexpr int kLockstepMinCols = 4;

struct CgDev {
  int64_t B, N;
  int c, S, S_dot, S_rz;
  int n_tridiag, max_iter, n_tridiag_iter, T;  // max_iter: the value the stop-rule floors are taken from
  float tol, eps, stop_after;
  int check_nan_first;
  // vectors
  float *x, *r, *p, *z, *Ap;
  // partials
  float *pAp_part, *rr_part, *rz_part;
  // per (b,col) scalars
  float *rhs_norm, *rz, *alpha, *beta, *resid_norm, *prev_ar, *prev_beta;
  int *rhs_is_zero, *has_conv;
  float* t_mat;
  CgCtrl* ctrl;
  float* ctrl_part;  // [3, kCtrlMaxG] per-workgroup partials of the control step
  unsigned long long* oc_gbuf;
  unsigned long long* oc_close; // [B] closing granules of k_cg_onchip5 + one counter word behind them (right after oc_gbuf:
                                // cleared with the control block by the same launch)
  unsigned long long* ls_gbuf;  // granules of the column-lockstep kernel (lo_cg_lockstep.hip) or nullptr
  unsigned long long* pf_gbuf;  // granules of the fused preconditioner apply (lo_precond_fused.hip) or nullptr
  int* pf_ctr;                  // one member hand-out counter per launch of that kernel (max_iter + 1 ints)
  unsigned long long* pf_gran;  // [B] tagged residual norms of the fused control step
  unsigned long long* sc_gbuf;  // granules of the multi-column iteration step (lo_cg_step_cols.hip) or nullptr
  int* sc_ctr;                  // hand-out | done counters of that kernel (max_iter + 1 ints each)
  unsigned long long* sc_gran;  // [3, B] tagged per-member aggregates of its folded control step
  int* oc_err;
  float* oc_resid;
  int* oc_init_conv;
  float* oc_zero_q;  // unpreconditioned resident path: an all-zero Q [B,N,4] and 1/d = 1 make z = r
  float* oc_ones;
  float* oc_ab;      // [iters, B, c, 2] alpha / beta record of the resident kernel (n_tridiag > 0)
  int* oc_maxoff;    // [T + 1] per-iteration max off-diagonal (fp32 bit patterns, non-negative values)
  long long* oc_dbg;
  double* rs_ws;      // partials / coefficients of the three-launch R-space form (lo_rspace.hip) or nullptr
};

// ---- init ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_cg_init_scal(CgDev d, const float* __restrict__ part) {
  const int64_t n = d.B * d.c;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const int64_t b = i / d.c;
    const int col = (int)(i % d.c);
    float acc = 0.f;
    for (int s = 0; s < d.S; ++s) acc += part[(b * d.S + s) * d.c + col];
    float nrm = sqrtf(acc);                      // rhs.norm(2, dim=-2)            linear_cg.py:177
    const int zero = nrm < d.eps;                // rhs_norm.lt(eps)               :178
    d.rhs_is_zero[i] = zero;
    d.rhs_norm[i] = zero ? 1.0f : nrm;           // masked_fill_(rhs_is_zero, 1)   :179
  }
}

__global__ __launch_bounds__(kThreads) void k_cg_init_vec(CgDev d, const float* __restrict__ rhs,
                                                           const float* __restrict__ x0, int rows_per) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int c = d.c, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float nrm = d.rhs_norm[(size_t)b * c + col];
  const size_t base = (size_t)b * N * c + col;
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    d.r[i] = rhs[i] / nrm;                       // rhs.div(rhs_norm)              :182
    d.x[i] = x0 ? x0[i] / nrm : 0.f;             // initial_guess.div(rhs_norm)    :183
  }
}

__global__ __launch_bounds__(kThreads) void k_cg_sub(float* __restrict__ r, const float* __restrict__ y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) r[i] -= y[i];  // residual = rhs - A x0   :186
}

__global__ __launch_bounds__(kThreads) void k_cg_ctrl_init(CgDev d) {
  __shared__ float red[kThreads];
  const int64_t n = d.B * d.c;
  float lnan = 0.f, lnotconv = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const int64_t b = i / d.c;
    const int col = (int)(i % d.c);
    float rr = 0.f, rz = 0.f;
    for (int s = 0; s < d.S; ++s) rr += d.rr_part[(b * d.S + s) * d.c + col];
    for (int s = 0; s < d.S_rz; ++s) rz += d.rz_part[(b * d.S_rz + s) * d.c + col];
    const float rn = sqrtf(rr);                  // residual.norm(2, dim=-2)       :204
    const int conv = rn < d.stop_after;          // :205
    d.resid_norm[i] = rn;
    d.has_conv[i] = conv;
    d.rz[i] = rz;                                // residual_inner_prod            :215
    if (rr != rr || rz != rz) lnan = 1.f;
    if (!conv) lnotconv = 1.f;
  }
  const float anynan = block_sum256(lnan, red);
  const float notconv = block_sum256(lnotconv, red);
  if (threadIdx.x == 0) {
    if (anynan > 0.f) {                          // torch.equal(residual, residual) :199-200
      d.ctrl->nan_detected = 1;
      d.ctrl->stop = 1;
    } else if (notconv == 0.f && d.n_tridiag == 0) {  // has_converged.all() and not n_tridiag :207-208
      d.ctrl->skipped = 1;
      d.ctrl->stop = 1;
    }
  }
}

// ---- iteration -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_cg_update_p(CgDev d, const float* __restrict__ z, int first,
                                                           int rows_per) {
  if (d.ctrl->stop) return;
  const int s = blockIdx.x, b = blockIdx.y;
  const int c = d.c, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const size_t base = (size_t)b * N * c + col;
  if (first) {
    for (int row = r0 + slot; row < r1; row += nrs) d.p[base + (size_t)row * c] = z[base + (size_t)row * c];  // :214
  } else {
    const float beta = d.beta[(size_t)b * c + col];
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      d.p[i] = fmaf(d.p[i], beta, z[i]);         // curr_conjugate_vec.mul_(beta).add_(precond_residual) :46
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_cg_update_xr(CgDev d, int rows_per) {
  if (d.ctrl->stop) return;
  __shared__ float red[kThreads];
  __shared__ float alpha_s[kMaxCols];
  const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
  const int c = d.c, N = (int)d.N;
  if (threadIdx.x < c) {
    const int col = threadIdx.x;
    float pAp = 0.f;
    for (int ss = 0; ss < d.S_dot; ++ss) pAp += d.pAp_part[((size_t)b * d.S_dot + ss) * c + col];  // :250-251
    const float rz = d.rz[(size_t)b * c + col];
    float a = (pAp < d.eps) ? 0.f : rz / pAp;    // safe division :254-257 (negative curvature zeroed too)
    if (d.has_conv[(size_t)b * c + col]) a = 0.f;  // alpha.masked_fill_(has_converged, 0) :260
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
    const size_t base = (size_t)b * N * c + col;
    for (int row = r0 + slot; row < r1; row += nrs) {
      const size_t i = base + (size_t)row * c;
      const float rn = fmaf(-a, d.Ap[i], d.r[i]);  // residual - alpha * mvms  :264 / :78
      d.r[i] = rn;
      d.x[i] = fmaf(a, d.p[i], d.x[i]);            // result + alpha * p       :31
      acc = fmaf(rn, rn, acc);
    }
  }
  const float tot = block_colsum(slot < nrs ? acc : 0.f, c, nrs, red);
  if (threadIdx.x < c) d.rr_part[((size_t)b * S + s) * c + col] = tot;
}

// Control step of iteration k, split in two so that it scales with B * c:
//  k_cg_scal (many workgroups): per (member, column) scalars -- beta (:34-42), residual norm (:298-299),
//    has_converged (:300) -- the tridiagonal recurrence (:311-332) of that column, and per-workgroup partials of the
//    residual-norm sum / NaN flag / max off-diagonal.
//  k_cg_ctrl (one workgroup): reduces the partials in fixed order and takes the batch-global decisions: stop rule
//    (:302-308), tridiag freeze (:326-327), last_tridiag_iter (:329).
// The recurrence entries of iteration k are written before the stop decision is known; the reference breaks BEFORE
// its tridiag update (:307-308), but a stop can only happen at k >= n_tridiag_iter (or the last iteration), where no
// entry inside the returned [: last_tridiag_iter + 1]^2 block is touched.
constexpr int kCtrlMaxG = 256;

__device__ __forceinline__ void cg_scal_body(const CgDev& d, int k, float* red) {
  const int64_t n = d.B * d.c;
  const bool tri = d.n_tridiag && k < d.n_tridiag_iter && !d.ctrl->tri_disabled;
  const int T = d.T;
  float lsum = 0.f, lnan = 0.f, lmax = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / d.c;
    const int col = (int)(i % d.c);
    float rr = 0.f, rzn = 0.f;
    for (int s = 0; s < d.S; ++s) rr += d.rr_part[(b * d.S + s) * d.c + col];
    for (int s = 0; s < d.S_rz; ++s) rzn += d.rz_part[(b * d.S_rz + s) * d.c + col];
    const float rzo = d.rz[i];                   // beta <- old residual_inner_prod   :34
    const float beta = (rzo < d.eps) ? 0.f : rzn / rzo;  // :39-42
    d.rz[i] = rzn;
    d.beta[i] = beta;
    float rn = sqrtf(rr);                        // vector_norm(residual)            :298
    if (d.rhs_is_zero[i]) rn = 0.f;              // :299
    d.resid_norm[i] = rn;
    d.has_conv[i] = rn < d.stop_after;           // :300
    lsum += rn;
    if (rr != rr || rzn != rzn) lnan = 1.f;
    if (tri && col < d.n_tridiag) {
      const int64_t it = b * d.n_tridiag + col;
      const float alpha = d.alpha[i];
      const float ar = 1.0f / ((alpha == 0.f) ? 1.0f : alpha);  // :314-317
      float* t = d.t_mat + ((size_t)col * d.B + b) * T * T;
      if (k == 0) {
        t[0] = ar;                               // :320
      } else {
        const float pb = d.prev_beta[it], par = d.prev_ar[it];
        t[k * T + k] = fmaf(pb, par, ar);        // addcmul(alpha_reciprocal, prev_beta, prev_alpha_reciprocal) :322
        const float off = sqrtf(pb) * par;       // :323
        t[k * T + k - 1] = off;
        t[(k - 1) * T + k] = off;                // :324
        lmax = fmaxf(lmax, off);
      }
      d.prev_ar[it] = ar;                        // :331-332
      d.prev_beta[it] = beta;
    }
  }
  const float bs = block_sum256(lsum, red);
  const float bn = block_sum256(lnan, red);
  const float bm = block_max256(lmax, red);
  if (threadIdx.x == 0) {
    d.ctrl_part[blockIdx.x] = bs;
    d.ctrl_part[kCtrlMaxG + blockIdx.x] = bn;
    d.ctrl_part[2 * kCtrlMaxG + blockIdx.x] = bm;
  }
}

// k < 0: the iteration index is read from the control block (graph replay: the previous control step left it there)
__global__ __launch_bounds__(kThreads) void k_cg_scal(CgDev d, int k) {
  if (d.ctrl->stop) return;
  __shared__ float red[kThreads];
  if (k < 0) k = d.ctrl->iterations;
  cg_scal_body(d, k, red);
}

__device__ __forceinline__ void cg_ctrl_body(const CgDev& d, int k, int G, float* red) {
  const int t = threadIdx.x;
  const float ls = (t < G) ? d.ctrl_part[t] : 0.f;
  const float ln = (t < G) ? d.ctrl_part[kCtrlMaxG + t] : 0.f;
  const float lm = (t < G) ? d.ctrl_part[2 * kCtrlMaxG + t] : -INFINITY;
  const int64_t n = d.B * d.c;
  const float mean = block_sum256(ls, red) / (float)n;
  const float anynan = block_sum256(ln, red);
  const float mx = block_max256(lm, red);
  const int kfloor = min(10, d.max_iter - 1);
  const bool stopnow = (k >= kfloor) && (mean < d.tol) &&
                       !(d.n_tridiag && k < min(d.n_tridiag_iter, d.max_iter - 1));  // :302-306
  if (t == 0) {
    d.ctrl->iterations = k + 1;
    d.ctrl->mean_resid = mean;
    if (k == 0 && d.check_nan_first && anynan > 0.f) {  // NaN matvec detected on the first product
      d.ctrl->nan_detected = 1;
      d.ctrl->stop = 1;
    }
    if (stopnow) {
      d.ctrl->tol_reached = 1;                   // :307
      d.ctrl->stop = 1;
    } else if (d.n_tridiag && k < d.n_tridiag_iter && !d.ctrl->tri_disabled) {
      if (k > 0 && mx < 1e-6f) d.ctrl->tri_disabled = 1;  // :326-327
      d.ctrl->last_tridiag_iter = k;                       // :329
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_cg_ctrl(CgDev d, int k, int G) {
  if (d.ctrl->stop) return;
  __shared__ float red[kThreads];
  if (k < 0) k = d.ctrl->iterations;
  cg_ctrl_body(d, k, G, red);
}

// B c <= 256 (one workgroup of partials): both halves of the control step in ONE launch
__global__ __launch_bounds__(kThreads) void k_cg_scal_ctrl(CgDev d, int k) {
  if (d.ctrl->stop) return;
  __shared__ float red[kThreads];
  if (k < 0) k = d.ctrl->iterations;
  cg_scal_body(d, k, red);
  __threadfence_block();
  __syncthreads();
  cg_ctrl_body(d, k, 1, red);
}

// ---- after the operator-resident kernel ran iterations 0 .. iters-1 for every column ----
// The tridiagonal recurrence (:311-332) is replayed from the recorded (masked) alpha / beta of each iteration:
//  k_oc_maxoff: per-iteration maximum off-diagonal over the whole batch (the freeze rule :326-327 is batch-global)
//  k_cg_ctrl_onchip: stop rule of iteration k = iters-1, NaN check, skip rule (:207-208), freeze iteration
//  k_oc_tridiag: writes the entries of the iterations that ran before the freeze
__device__ __forceinline__ float oc_recip(float alpha) { return 1.0f / ((alpha == 0.f) ? 1.0f : alpha); }  // :314-317

__global__ __launch_bounds__(kThreads) void k_oc_maxoff(CgDev d, const float* __restrict__ ab, int ktri) {
  const int64_t n = d.B * d.n_tridiag;
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / d.n_tridiag;
  const int col = (int)(i % d.n_tridiag);
  const size_t bc = (size_t)b * d.c + col, stride = (size_t)d.B * d.c;
  for (int k = 1; k < ktri; ++k) {
    const float pa = ab[2 * ((size_t)(k - 1) * stride + bc)], pb = ab[2 * ((size_t)(k - 1) * stride + bc) + 1];
    const float off = sqrtf(pb) * oc_recip(pa);  // :323
    if (off > 0.f) atomicMax(d.oc_maxoff + k, __float_as_int(off));
  }
}

__global__ __launch_bounds__(kThreads) void k_oc_tridiag(CgDev d, const float* __restrict__ ab, int ktri) {
  const int64_t n = d.B * d.n_tridiag;
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / d.n_tridiag;
  const int col = (int)(i % d.n_tridiag);
  const size_t bc = (size_t)b * d.c + col, stride = (size_t)d.B * d.c;
  const int T = d.T;
  const int klast = min(ktri - 1, d.ctrl->last_tridiag_iter);
  float* t = d.t_mat + ((size_t)col * d.B + b) * T * T;
  float par = 0.f, pb = 0.f;
  for (int k = 0; k <= klast; ++k) {
    const float ar = oc_recip(ab[2 * ((size_t)k * stride + bc)]);
    if (k == 0) {
      t[0] = ar;                                 // :320
    } else {
      t[k * T + k] = fmaf(pb, par, ar);          // :322
      const float off = sqrtf(pb) * par;         // :323
      t[k * T + k - 1] = off;
      t[(k - 1) * T + k] = off;                  // :324
    }
    par = ar;                                    // :331-332
    pb = ab[2 * ((size_t)k * stride + bc) + 1];
  }
}

__global__ __launch_bounds__(kThreads) void k_cg_ctrl_onchip(CgDev d, const float* __restrict__ resid_rec,
                                                              const int* __restrict__ init_conv, int iters, int ktri,
                                                              CgCtrl* __restrict__ mirror, unsigned ticket) {
  __shared__ float red[kThreads];
  float lsum = 0.f, lnan = 0.f, lnotconv = 0.f;
  const int64_t n = d.B * d.c;
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
    d.ctrl->iterations = iters;
    d.ctrl->mean_resid = mean;
    if (ktri > 0) {  // freeze rule: the first k > 0 whose largest off-diagonal is below 1e-6 is the last one recorded
      int last = ktri - 1;
      for (int kk = 1; kk < ktri; ++kk)
        if (__int_as_float(d.oc_maxoff[kk]) < 1e-6f) {
          last = kk;
          d.ctrl->tri_disabled = 1;
          break;
        }
      d.ctrl->last_tridiag_iter = last;          // :329
    }
    if (anynan > 0.f) {
      d.ctrl->nan_detected = 1;
      d.ctrl->stop = 1;
    } else if (notconv == 0.f && d.n_tridiag == 0) {  // every column converged before the first iteration (:207-208)
      d.ctrl->skipped = 1;
      d.ctrl->iterations = 0;
      d.ctrl->stop = 1;
    } else if (k >= min(10, d.max_iter - 1) && mean < d.tol) {  // (k >= min(n_tridiag_iter, max_iter-1) by construction)
      d.ctrl->tol_reached = 1;
      d.ctrl->stop = 1;
    }
    // the host reads the decision from pinned memory right after the stream drains: no device-to-host copy command
    if (mirror) {
      *mirror = *d.ctrl;
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<unsigned*>(mirror) + 63, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_cg_final(CgDev d, float* __restrict__ xout, int rows_per) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int c = d.c, N = (int)d.N;
  const int nrs = kThreads / c;
  const int col = threadIdx.x % c, slot = threadIdx.x / c;
  if (slot >= nrs) return;
  const int r0 = s * rows_per, r1 = min(N, r0 + rows_per);
  const float nrm = d.rhs_norm[(size_t)b * c + col];
  const size_t base = (size_t)b * N * c + col;
  for (int row = r0 + slot; row < r1; row += nrs) {
    const size_t i = base + (size_t)row * c;
    xout[i] = d.x[i] * nrm;                      // result.mul(rhs_norm)  :335
  }
}

// ---- host engine -----------------------------------------------------------------------------------
struct CgLayout {
  Split sp;
  int S_dot;
  bool precond;
  int pre_R4;
  bool pre_pad;
};

static int padded_rank_k(int k) {
  int rq = (k + 3) / 4, p = 1;
  while (p < rq) p <<= 1;
  return 4 * p;
}

// ---- engine selection: a PURE function of the shapes, the pointers' null-ness, the parameters and the number of
// workgroup slots (no device memory is read, nothing is launched).  lo_cg_solve_f32 executes this plan; the only
// decisions left to run time are the fall-backs after an occupancy query refuses a kernel or a hand-off times out.
// Exported as lo_cg_plan_f32 so that the selection matrix has a table-driven test (tests/test_host_api.py).
static int padded_rank_c(int64_t R) {  // floats per row of the root as the skinny kernels read it (lo_matvec.hip)
  int64_t rq = (R + 3) / 4, p = 1;
  while (p < rq) p <<= 1;
  return (int)(4 * p);
}

struct CgShape {  // what cg_layout allocates for the resident paths (it sizes the workspace from the same predicates)
  bool oc_shape, has_ab, has_ls_gbuf, has_zero_q, pf_shape, sc_shape, sc_alloc, rs_cols;
};

static CgShape cg_shape(const lo_op_desc* op, const lo_precond_desc* pre, bool pre_cb, const lo_cg_params* prm) {
  const int64_t B = op->B, N = op->N, c = prm->c;
  CgShape s;
  s.oc_shape = op->kind == LO_OP_LOWRANK_DIAG && c <= 64 && N >= 256 && N <= 65536;
  s.has_ab = s.oc_shape && prm->n_tridiag > 0;
  s.has_ls_gbuf = s.oc_shape && c >= kLockstepMinCols && N <= 8192;
  s.has_zero_q = !pre && !pre_cb && s.oc_shape;
  s.rs_cols = s.oc_shape && pre && pre->RS && rspace_cols_eligible(padded_rank_c(op->R), N, c);
  const Split sp = choose_split(B, N, 256);
  s.pf_shape = pre && precond_fused_eligible(B, N, c, padded_rank_k(pre->k), sp.S);
  // every other streaming shape of up to 32 columns and 16384 rows: the whole step behind the product in one launch
  const int ldq = pre ? padded_rank_k(pre->k) : 0;
  s.sc_alloc = !s.pf_shape && ldq <= 16 && cg_step_cols_eligible(B, N, c, ldq);  // (the sizing pass assumes a closure)
  s.sc_shape = s.sc_alloc && !pre_cb;
  return s;
}

static void cg_plan(const lo_op_desc* op, const lo_precond_desc* pre, bool pre_cb, bool has_x0, const lo_cg_params* prm,
                    int oc_nwg, lo_cg_plan* out) {
  memset(out, 0, sizeof(*out));
  const int64_t B = op->B, N = op->N;
  const int c = (int)prm->c;
  const CgShape sh = cg_shape(op, pre, pre_cb, prm);
  const int fmi = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter;  // (linear_cg.py:303-305)
  int kfloor0 = std::min(10, fmi - 1);
  if (prm->n_tridiag) kfloor0 = std::max(kfloor0, std::min(prm->max_tridiag_iter, fmi - 1));  // first possible stop
  out->first_stop_iteration = kfloor0;
  const bool global_rule = prm->stop_reduce != nullptr;
  const bool pre_root = pre && pre->F && pre->EF && pre->rf_ld > 0;
  const int RC = op->kind == LO_OP_LOWRANK_DIAG ? padded_rank_c(op->R) : 0;
  const int preR4 = pre ? padded_rank_k(pre->k) : 0;
  // no preconditioner (N < min_preconditioning_size in the host API): the resident kernels run with Q = 0 and
  // 1/d = 1, i.e. z = r and r.z = ||r||^2, which is the reference's unpreconditioned update (linear_cg.py:49-95)
  const bool oc_nopre = !pre && !pre_cb && sh.has_zero_q;
  const int ocR4 = oc_nopre ? 4 : preR4;
  // (the second generation loops over the columns with the Q form; the third advances 16 columns together on the matrix
  // cores; the root-form kernel needs F / EF of the operator's own root.  The first generation went in round 6.)
  const bool oc_base = (op->kind == LO_OP_LOWRANK_DIAG) && RC <= kMaxRank && (pre || oc_nopre) && !pre_cb && !has_x0 &&
                       prm->max_iter >= 11 && prm->max_iter > kfloor0 && oc_nwg >= 64 && !resident_off() &&
                       (prm->n_tridiag == 0 || sh.has_ab) && B < (1 << 24) - 1024;
  // column split: full chunks of 16 (and a last chunk of at least kLockstepMinCols) -> lockstep kernel, the rest serial
  int ls_cols = 0;
  if (oc_base && sh.has_ls_gbuf && !getenv("LO_OC_NO_LOCKSTEP") && (!pre || pre->Q) &&
      lockstep_eligible(RC, pre ? preR4 : 0, pre != nullptr, N, c)) {
    const int full = (c / 16) * 16, rem = c - full;
    ls_cols = full + (rem >= kLockstepMinCols ? rem : 0);
  }
  const bool root_match = oc_nopre || (pre_root && pre->rf_ld == RC);
  const bool oc_root_ok = ls_cols < c && !getenv("LO_OC_GEN2") && onchip5_eligible(RC, N, c - ls_cols) && root_match;
  const bool gen2_ok = ls_cols < c && onchip4_eligible(RC, ocR4, N, c - ls_cols) && (!pre || pre->Q);
  const bool oc_ok = oc_base && (ls_cols == c || oc_root_ok || gen2_ok);
  out->streaming_precond = pre_cb ? LO_STREAM_PRE_CLOSURE : (pre ? LO_STREAM_PRE_TWO_PASS : LO_STREAM_PRE_NONE);
  if (pre && pre->Q && sh.pf_shape && !tls_no_fused_precond && oc_nwg >= 64) {
    const bool kron = op->kind == LO_OP_KRON_DIAG && op->diag_mode == LO_DIAG_CONST && pre->constant_diag && pre->kron_a &&
                      pre->kron_b && pre->kron_F && pre->k <= 16 && !getenv("LO_NO_KRON_ROOT");
    out->streaming_precond = kron ? LO_STREAM_PRE_FUSED_KRON : LO_STREAM_PRE_FUSED_Q;
  }
  if ((out->streaming_precond == LO_STREAM_PRE_TWO_PASS || out->streaming_precond == LO_STREAM_PRE_NONE) && sh.sc_shape &&
      (!pre || pre->Q) && !tls_no_fused_precond && std::min(512, 2 * oc_nwg) / 8 >= cg_step_cols_group(N) &&
      cg_step_cols_worthwhile(B, N, c, pre != nullptr))
    out->streaming_precond = pre ? LO_STREAM_PRE_FUSED_COLS : LO_STREAM_NOPRE_FUSED_COLS;
  const bool opaque = (op->kind == LO_OP_CALLBACK) || pre_cb;
  out->poll_chunk = (opaque || global_rule) ? 1 : 4;
  // a root-form-only preconditioner cannot feed the streaming engine: without a resident kernel the caller is told to
  // build the Q form (LO_ERR_UNSUPPORTED)
  out->needs_q = (pre && !pre->Q && !(oc_ok && (ls_cols == c || oc_root_ok))) ? 1 : 0;
  if (!oc_ok) return;
  out->resident = 1;
  out->resident_iterations = kfloor0 + 1;
  out->lockstep_cols = ls_cols;
  out->lockstep_group = ls_cols ? lockstep_group_size(N) : 0;
  if (ls_cols < c) {
    if (oc_root_ok) {
      out->serial_engine = LO_ENGINE_RESIDENT_ROOT;
      out->serial_group = (getenv("LO_OC_GW8") && N <= 32768) ? onchip4_group_size(N) : onchip5_group_size(N);
    } else {
      out->serial_engine = LO_ENGINE_RESIDENT_GEN2;
      out->serial_group = onchip4_group_size(N);
    }
  }
  // Result-only first pass ("lean"): the resident kernels that take it (root-form serial columns, lockstep) write the
  // scaled result but not x / r / p / z of a possible continuation -- 4 of the 5 vectors they used to store.  All
  // BASELINE shapes stop at the floor; when the rule does not hold there, the same launches are repeated with the state
  // (deterministic kernels: the continuation starts from the very numbers the first pass computed).
  out->lean = (!global_rule && !getenv("LO_OC_KEEP_STATE") && (ls_cols == c || out->serial_engine == LO_ENGINE_RESIDENT_ROOT))
                  ? 1 : 0;
  // R-space forms of the result-only pass (lo_rspace.hip), with the fp64 Gram matrices of the root at hand
  const bool rs_base = pre && pre->RS && pre_root && pre->rf_ld == RC && !global_rule && !getenv("LO_OC_KEEP_STATE");
  if (rs_base && sh.rs_cols && (c >= 2 || prm->n_tridiag > 0) && !getenv("LO_NO_RSPACE_COLS")) {
    out->rspace = 1;  // all columns, three streaming launches; the engines above are the repeat with the state
    out->lean = 1;
  } else if (rs_base && out->lean && ls_cols == 0 && c == 1 && prm->n_tridiag == 0 && pre->E &&
             out->serial_engine == LO_ENGINE_RESIDENT_ROOT && rspace_eligible(RC, N, c) && !getenv("LO_OC_NO_RSPACE") &&
             !getenv("LO_OC_NO_WREC")) {
    out->rspace = 2;  // the single column inside the resident launch
  }
}


static size_t cg_layout(const lo_op_desc* op, const lo_precond_desc* pre, bool pre_cb, const lo_cg_params* prm,
                        void* ws, size_t ws_bytes, CgDev* d, MatvecPlan* pl, lo_matvec_cb mv_cb, void* mv_user,
                        const float** Qpad, float** upart, hipStream_t st, int* rc_out, bool init) {
  const int64_t B = op->B, N = op->N, c = prm->c;
  Split sp = choose_split(B, N, 256);
  Arena ar(ws, ws_bytes);
  const size_t nv = (size_t)B * N * c;
  const bool precond = pre != nullptr || pre_cb;
  CgDev dd;
  dd.B = B; dd.N = N; dd.c = (int)c; dd.S = sp.S;
  dd.ctrl = ar.take<CgCtrl>(1);
  // (granule buffer of the serial resident kernels right behind the control block: ONE memset clears both)
  // (k_cg_rspace launches 2 x onchip_num_workgroups() workgroups and indexes its granules by group: sized from that count,
  //  not from the 256-CU part's 528 -- ADVICE r5)
  dd.oc_gbuf = ar.take<unsigned long long>(std::max(onchip_gbuf_bytes(66), rspace_gbuf_bytes(std::max(66 * 8, 2 * onchip_num_workgroups()))) / sizeof(unsigned long long));
  dd.oc_close = ar.take<unsigned long long>((size_t)B + 2);
  dd.x = ar.take<float>(nv);
  dd.r = ar.take<float>(nv);
  dd.p = ar.take<float>(nv);
  dd.Ap = ar.take<float>(nv);
  dd.z = precond ? ar.take<float>(nv) : nullptr;
  int S_dot = sp.S;
  if (op->kind == LO_OP_DENSE_DIAG) {
    S_dot = dense_S_dot(B, N, c);
  } else if (op->kind == LO_OP_KRON_DIAG) {
    S_dot = kron_S_dot((int)op->R, (int)op->n2, c, sp.S);
  }
  dd.S_dot = S_dot;
  dd.pAp_part = ar.take<float>((size_t)B * std::max(S_dot, sp.S) * c);
  dd.rr_part = ar.take<float>((size_t)B * sp.S * c);
  dd.rz_part = precond ? ar.take<float>((size_t)B * sp.S * c) : dd.rr_part;
  dd.S_rz = sp.S;
  const size_t ns = (size_t)B * c;
  dd.rhs_norm = ar.take<float>(ns);
  dd.rz = ar.take<float>(ns);
  dd.alpha = ar.take<float>(ns);
  dd.beta = ar.take<float>(ns);
  dd.resid_norm = ar.take<float>(ns);
  dd.rhs_is_zero = ar.take<int>(ns);
  dd.has_conv = ar.take<int>(ns);
  const size_t nt = (size_t)B * std::max(1, (int)prm->n_tridiag);
  dd.prev_ar = ar.take<float>(nt);
  dd.prev_beta = ar.take<float>(nt);
  dd.ctrl_part = ar.take<float>(3 * 256);
  // operator-resident fast path scratch (c == 1): granule buffer, error word, per-iteration residuals
  dd.oc_err = reinterpret_cast<int*>(reinterpret_cast<char*>(dd.ctrl) + offsetof(CgCtrl, oc_err));  // (+ oc_next)
  const int fmi0 = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter;
  int oc_iters = std::min(10, fmi0 - 1);
  if (prm->n_tridiag) oc_iters = std::max(oc_iters, std::min(prm->max_tridiag_iter, fmi0 - 1));
  oc_iters = std::max(1, oc_iters + 1);
  const CgShape shp = cg_shape(op, pre, pre_cb, prm);  // (the predicates cg_plan decides on)
  const bool oc_shape = shp.oc_shape;
  const size_t oc_n = oc_shape ? (size_t)B * c : 1;
  dd.oc_resid = ar.take<float>(oc_n * oc_iters);
  dd.oc_init_conv = ar.take<int>(oc_n);
  dd.oc_ab = shp.has_ab ? ar.take<float>(2 * oc_n * oc_iters) : nullptr;
  dd.oc_maxoff = ar.take<int>((size_t)std::max(1, (int)prm->max_tridiag_iter) + 1);
  dd.oc_dbg = ar.take<long long>(16);
  dd.rs_ws = shp.rs_cols ? ar.take<double>(rspace_cols_ws_doubles(B, N, padded_rank_c(op->R), (int)c)) : nullptr;
  dd.ls_gbuf = shp.has_ls_gbuf
                   ? ar.take<unsigned long long>(lockstep_gbuf_bytes(32, 16) / sizeof(unsigned long long))
                   : nullptr;
  const bool pf_shape = shp.pf_shape;
  dd.pf_gbuf = pf_shape ? ar.take<unsigned long long>(precond_fused_gbuf_bytes() / sizeof(unsigned long long)) : nullptr;
  // (decided by the shape, not by the pointer: the sizing pass runs on a null arena)
  dd.pf_ctr = pf_shape ? ar.take<int>(2 * ((size_t)std::max(1, (int)prm->max_iter) + 1)) : nullptr;  // hand-out | done
  dd.pf_gran = pf_shape ? ar.take<unsigned long long>((size_t)B) : nullptr;
  dd.sc_gbuf = shp.sc_alloc ? ar.take<unsigned long long>(cg_step_cols_gbuf_bytes() / sizeof(unsigned long long)) : nullptr;
  dd.sc_ctr = shp.sc_alloc ? ar.take<int>(2 * ((size_t)std::max(1, (int)prm->max_iter) + 1)) : nullptr;
  dd.sc_gran = shp.sc_alloc ? ar.take<unsigned long long>(3 * (size_t)B) : nullptr;
  dd.oc_zero_q = nullptr;
  dd.oc_ones = nullptr;
  if (shp.has_zero_q) {
    dd.oc_zero_q = ar.take<float>((size_t)B * N * 4);
    dd.oc_ones = ar.take<float>((size_t)B);
  }
  // preconditioner staging
  if (pre) {
    const int R4 = padded_rank_k(pre->k);
    float* up = ar.take<float>((size_t)B * sp.S * R4 * c);
    if (upart) *upart = up;
    if (pre->Q && pre->ldq != R4) {
      float* qp = ar.take<float>((size_t)B * N * R4);
      if (init && ar.ok && ws) {
        if (pre->ldq != pre->k) { if (rc_out) *rc_out = LO_ERR_BADARG; }
        else {
          int rc = pad_rows(pre->Q, pre->k, qp, R4, B * N, st);
          if (rc && rc_out) *rc_out = rc;
        }
      }
      if (Qpad) *Qpad = qp;
    } else if (Qpad) {
      *Qpad = pre->Q;
    }
  }
  // matvec plan
  size_t before = ar.off;
  (void)before;
  if (init) {
    int rc = matvec_plan_init(pl, op, mv_cb, mv_user, c, sp, &ar, st);
    if (rc && rc_out) *rc_out = rc;
  } else {
    ar.off += matvec_plan_bytes(op, c, sp);
  }
  if (d) *d = dd;
  if (init && !ar.ok && rc_out && *rc_out == LO_OK) *rc_out = LO_ERR_WORKSPACE;
  return ar.off + 1024;
}

}  // namespace lo

using namespace lo;

extern "C" {

size_t lo_cg_workspace_bytes(const lo_op_desc* op, const lo_precond_desc* pre, const lo_cg_params* prm) {
  if (!op || !prm) return 0;
  // worst case: assume an opaque preconditioner closure may be used as well (needs z)
  lo_precond_desc dummy;
  const lo_precond_desc* p = pre;
  if (!p) {
    dummy.k = 4; dummy.ldq = 4; dummy.constant_diag = 0; dummy.reserved = 0; dummy.Q = nullptr; dummy.dinv = nullptr;
    p = &dummy;
  }
  size_t need = cg_layout(op, p, true, prm, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, false);
  if (!pre)  // the unpreconditioned resident path stages an all-zero Q instead of z
    need = std::max(need, cg_layout(op, nullptr, false, prm, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, false));
  return need;
}

int lo_cg_solve_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const lo_precond_desc* pre,
                    lo_matvec_cb precond_cb, void* precond_user, const lo_cg_params* prm, const float* rhs,
                    const float* x0, float* x, float* t_mat, void* ws, size_t ws_bytes, lo_cg_info* info,
                    void* stream) {
  if (!op || !prm || !rhs || !x || !ws || !info) return LO_ERR_BADARG;
  if (prm->c < 1 || prm->c > kMaxCols) return LO_ERR_UNSUPPORTED;
  if (prm->n_tridiag < 0 || prm->n_tridiag > prm->c) return LO_ERR_BADARG;
  if (prm->n_tridiag && !t_mat) return LO_ERR_BADARG;
  if (pre && precond_cb) return LO_ERR_BADARG;
  const bool pre_root = pre && pre->F && pre->EF && pre->rf_ld > 0;  // root form of the preconditioner available
  if (pre && (pre->k < 1 || pre->k > kMaxRank || !pre->dinv || (!pre->Q && !pre_root))) return LO_ERR_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = op->B, N = op->N;
  const int c = (int)prm->c;
  resident_tick();

  CgDev d;
  MatvecPlan pl;
  PlanGuard pl_guard(&pl);
  const float* Qp = nullptr;
  float* upart = nullptr;
  int rc = LO_OK;
  cg_layout(op, pre, precond_cb != nullptr, prm, ws, ws_bytes, &d, &pl, matvec, matvec_user, &Qp, &upart, st, &rc, true);
  if (rc) return rc;
  const Split sp = pl.sp;
  d.n_tridiag = prm->n_tridiag;
  const int fmi = prm->floor_max_iter > 0 ? prm->floor_max_iter : prm->max_iter;  // (linear_cg.py:303-305)
  d.max_iter = fmi;
  d.n_tridiag_iter = prm->max_tridiag_iter;
  d.T = prm->max_tridiag_iter;
  // batch-global stopping rule over several ranks (lo_stop_reduce_cb): the device never decides the tolerance stop
  // itself (tol = -1 can not be undercut by a mean of norms); the host evaluates the rule on the all-reduced statistic
  const bool global_rule = prm->stop_reduce != nullptr;
  d.tol = global_rule ? -1.0f : prm->tolerance;
  d.eps = prm->eps;
  d.stop_after = prm->stop_updating_after;
  d.check_nan_first = (x0 == nullptr);
  d.t_mat = t_mat;
  const bool precond = (pre != nullptr) || (precond_cb != nullptr);
  const int preR4 = pre ? padded_rank_k(pre->k) : 0;
  const int* stop = &d.ctrl->stop;
  dim3 gridv(sp.S, (unsigned)B), block(kThreads);

  {  // control block (+ the granule buffer behind it when a resident kernel may run)
    const bool oc_possible = op->kind == LO_OP_LOWRANK_DIAG && !resident_off();
    const size_t span = oc_possible ? (size_t)(reinterpret_cast<char*>(d.oc_close + B + 2) - reinterpret_cast<char*>(d.ctrl))
                                    : sizeof(CgCtrl);
    rc = zero_span(d.ctrl, span, st);
    if (rc) return rc;
  }
  if (prm->n_tridiag)
    LO_HIP_CHECK(hipMemsetAsync(t_mat, 0, sizeof(float) * (size_t)prm->n_tridiag * B * d.T * d.T, st));

  auto apply_precond = [&](const float* r, float* z, float* dotp) -> int {
    if (pre) {
      int e = skinny_tn(Qp, preR4, preR4, r, c, upart, B, N, sp, stop, st);
      if (e) return e;
      return skinny_nn(Qp, preR4, preR4, upart, pre->dinv, pre->constant_diag ? LO_DIAG_CONST : LO_DIAG_FULL, -1.0f, r,
                       c, z, dotp, B, N, sp, stop, st);
    }
    int e = precond_cb(precond_user, r, z, B, N, c, (void*)st);
    if (e) return LO_ERR_LAUNCH;
    return vec_dot_part(r, z, c, dotp, B, N, sp, stop, st);
  };

  // ---- operator-resident fast path for the guaranteed iterations (lo_cg_onchip.hip) ----
  int k_start = 0;
  bool x_written = false;  // the resident kernel already wrote result * rhs_norm
  CgCtrl h;
  memset(&h, 0, sizeof(h));
  // linear_cg.py:302-308 on the statistic of ALL ranks; called exactly once per completed iteration k >= first stop
  auto global_check = [&]() -> int {
    if (!global_rule) return LO_OK;
    const int kdone = h.iterations - 1;
    double vals[3] = {(double)h.mean_resid * (double)(B * c), (double)(B * c), h.stop ? 1.0 : 0.0};
    if (prm->stop_reduce(prm->stop_reduce_user, vals)) return LO_ERR_LAUNCH;
    const float gmean = (float)(vals[0] / vals[1]);
    h.mean_resid = gmean;
    if (vals[2] > 0.0) {  // some rank stopped on its own (NaN in a product): everybody stops
      h.stop = 1;
      return LO_OK;
    }
    const int kfl = std::min(10, fmi - 1);
    const bool stopnow = kdone >= kfl && gmean < prm->tolerance &&
                         !(prm->n_tridiag && kdone < std::min(prm->max_tridiag_iter, fmi - 1));
    if (stopnow) {
      h.tol_reached = 1;
      h.stop = 1;
    }
    return LO_OK;
  };
  const int oc_nwg = onchip_num_workgroups();
  // ---- the plan (pure: cg_plan above; lo_cg_plan_f32 reports the same decisions to the tests) ----
  lo_cg_plan plan;
  cg_plan(op, pre, precond_cb != nullptr, x0 != nullptr, prm, oc_nwg, &plan);
  const int kfloor0 = plan.first_stop_iteration;
  const bool oc_nopre = !pre && !precond_cb && d.oc_zero_q != nullptr;
  const int ocR4 = oc_nopre ? 4 : preR4;
  const bool oc_ok = plan.resident != 0;
  int ls_cols = plan.lockstep_cols;
  bool lean = plan.lean != 0;
  const int miss_slot = lean ? lean_miss_find(op, prm, pre ? pre->generation : 0) : -1;
  const bool lean_skipped = miss_slot >= 0 && pre && pre->Q && !getenv("LO_OC_NO_LEAN_MEMO");
  if (lean_skipped) lean = false;  // this operator missed the floor last time: write the state in the first pass
  lo_cg_plan exec = plan;  // the plan as executed: run-time fall-backs are recorded here (lo_cg_last_executed)
  exec.resident = 0;
  exec.serial_engine = LO_ENGINE_NONE;
  exec.rspace = 0;
  tls_rspace_resident_ran = false;
  bool rs_force_dense = false;  // the diagonal form asked for the dense one (CgCtrl::rs_redo): one more result-only pass
  for (int oc_pass = 0; oc_pass < 3; ++oc_pass) {
  bool oc_redo = false;
  if (oc_ok) {
    OnchipArgs a;
    a.C = pl.Apad; a.d = op->d;
    a.d_mode = op->diag_mode;
    if (oc_nopre) {
      if (ls_cols < c && (getenv("LO_OC_GEN2") || !onchip5_eligible(pl.R4, N, c - ls_cols))) {  // (second generation only)
        LO_HIP_CHECK(hipMemsetAsync(d.oc_zero_q, 0, sizeof(float) * (size_t)B * N * 4, st));
        LO_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)d.oc_ones, 0x3f800000, (size_t)B, st));  // 1.0f
      }
      a.Q = d.oc_zero_q; a.dinv = d.oc_ones; a.dinv_mode = LO_DIAG_CONST;
    } else {
      a.Q = Qp; a.dinv = pre->dinv; a.dinv_mode = pre->constant_diag ? LO_DIAG_CONST : LO_DIAG_FULL;
    }
    a.rhs = rhs; a.B = B; a.N = (int)N;
    a.c = c; a.ab_rec = prm->n_tridiag ? d.oc_ab : nullptr;
    a.col0 = 0; a.ncols = c; a.RK = pre ? preR4 : 0; a.RCg = pl.R4;
    a.F = nullptr; a.EF = nullptr; a.E = nullptr; a.RS = nullptr; a.RSD = nullptr;
    a.close_gran = nullptr; a.close_count = nullptr; a.close_ctrl = nullptr; a.close_mirror = nullptr;
    a.close_ticket = 0; a.close_tol = 0.f; a.close_floor_ok = 0;
    bool close_in_kernel = false;
    unsigned close_ticket = 0;
    a.iters = kfloor0 + 1;
    a.eps = prm->eps; a.stop_after = prm->stop_updating_after;
    a.x = d.x; a.r = d.r; a.p = d.p; a.z = d.z;
    auto lean_state = [&](bool on) {  // (kernels without the result-only mode get the state pointers back)
      a.x = on ? nullptr : d.x; a.r = on ? nullptr : d.r; a.p = on ? nullptr : d.p; a.z = on ? nullptr : d.z;
    };
    a.rhs_norm = d.rhs_norm; a.rz = d.rz; a.alpha = d.alpha; a.beta = d.beta; a.resid_norm = d.resid_norm;
    a.rhs_is_zero = d.rhs_is_zero; a.has_conv = d.has_conv;
    a.resid_rec = d.oc_resid; a.init_conv = d.oc_init_conv; a.err = d.oc_err;
    a.allow_l2_handoff = onchip_l2_handoff_allowed();
    a.prefetch = getenv("LO_OC_NO_PREFETCH") ? 0 : 1;
    // LO_OC_DEBUG=<member index to time> (serial-column kernels), LO_LS_DEBUG=<member> (lockstep kernel)
    const bool ls_dbg = getenv("LO_LS_DEBUG") != nullptr && B >= 8;
    const bool oc_dbg = !ls_dbg && getenv("LO_OC_DEBUG") != nullptr && B >= 8;
    a.dbg = nullptr;
    a.dbg_member = oc_dbg ? atoi(getenv("LO_OC_DEBUG")) : (ls_dbg ? atoi(getenv("LO_LS_DEBUG")) : 0);
    // LO_OC_TEST_FALLBACK: start with the error word set, as if a hand-off had timed out (exercises the host fallback)
    // (error word and member counters live in the control block: cleared with it, copied back with it)
    // lo_resident_inject_timeouts(n): the same through the real bookkeeping (cool-down, re-arm) -- the multi-rank tests
    if (getenv("LO_OC_TEST_FALLBACK") || (oc_pass == 0 && resident_take_injection()))
      LO_HIP_CHECK(hipMemsetAsync(d.oc_err, 1, 1, st));
    if (oc_dbg || ls_dbg) LO_HIP_CHECK(hipMemsetAsync(d.oc_dbg, 0, 16 * sizeof(long long), st));
    rc = LO_OK;
    bool xout_ok = true;  // every launched kernel wrote result * rhs_norm itself
    bool serial_done = false;
    bool rs_cols_ran = false;
    if (plan.rspace == 1 && lean && d.rs_ws) {  // all columns on R + 1 coordinates: three streaming launches
      a.xout = x;
      lean_state(true);
      a.F = pre->F; a.EF = pre->EF; a.E = pre->E; a.RS = pre->RS; a.RSD = pre->RSD;
      rc = rspace_cols_launch(pl.R4, a, d.rs_ws, st);
      if (rc == LO_OK) {
        serial_done = true;
        rs_cols_ran = true;
      } else if (rc == LO_ERR_UNSUPPORTED) {
        rc = LO_OK;
      }
    }
    const int ls_plan = ls_cols;
    if (rs_cols_ran) ls_cols = 0;
    if (ls_cols) {  // third generation: columns [0, ls_cols)
      a.ncols = ls_cols;
      a.xout = x;
      lean_state(lean);
      a.gbuf = d.ls_gbuf; a.next_member = d.oc_err + 2;
      a.dbg = ls_dbg ? d.oc_dbg : nullptr;
      LO_HIP_CHECK(hipMemsetAsync(d.ls_gbuf, 0, lockstep_gbuf_bytes(32, 16), st));
      a.GW = lockstep_group_size(N);
      a.RW = (int)((N + a.GW - 1) / a.GW);
      rc = lockstep_launch(pl.R4, pre != nullptr, a, std::min(oc_nwg, 256), st);
      if (rc == LO_ERR_UNSUPPORTED && a.GW == 16) {  // (two workgroups per CU do not fit: one 1024-row workgroup)
        a.GW = 8;
        a.RW = (int)((N + 7) / 8);
        rc = lockstep_launch(pl.R4, pre != nullptr, a, std::min(oc_nwg, 256), st);
      }
      if (rc == LO_OK) exec.lockstep_group = a.GW;
      if (rc == LO_ERR_UNSUPPORTED) {  // (does not fit this device: all columns go to the serial kernels)
        ls_cols = 0;
        rc = onchip4_eligible(pl.R4, ocR4, N, c) ? LO_OK : LO_ERR_UNSUPPORTED;
      }
    }
    // (the plan chose the root-form kernel for the serial columns -- or the lockstep kernel did not fit this device and
    // its columns come here as well)
    if (rc == LO_OK && !serial_done && ls_cols < c && !getenv("LO_OC_GEN2") && onchip5_eligible(pl.R4, N, c - ls_cols) &&
        (oc_nopre || (pre_root && pre->rf_ld == pl.R4))) {
      // root-form serial-column kernel (one all-reduce per iteration): columns [ls_cols, c)
      a.GW = (getenv("LO_OC_GW8") && N <= 32768) ? onchip4_group_size(N) : onchip5_group_size(N);
      a.RW = (int)((N + a.GW - 1) / a.GW);
      a.col0 = ls_cols; a.ncols = c - ls_cols;
      a.xout = x;
      lean_state(lean);
      a.F = oc_nopre ? nullptr : pre->F;
      a.EF = oc_nopre ? nullptr : pre->EF;
      a.E = oc_nopre ? nullptr : pre->E;
      a.RS = (oc_nopre || plan.rspace != 2) ? nullptr : pre->RS;
      a.RSD = (a.RS && !rs_force_dense) ? pre->RSD : nullptr;
      tls_rspace_resident_ran = false;
      a.gbuf = d.oc_gbuf; a.next_member = d.oc_err + 1;
      a.dbg = oc_dbg ? d.oc_dbg : nullptr;
      // one column, no tridiagonals, no lockstep launch in front: the kernel closes the solve itself (stop rule, NaN /
      // skip conditions, mirror + ticket) -- no separate control launch
      close_in_kernel = c == 1 && ls_cols == 0 && prm->n_tridiag == 0 && !oc_dbg && !getenv("LO_OC_NO_INKERNEL_CLOSE") &&
                        pinned_status_block() != nullptr;
      if (close_in_kernel) {
        a.close_gran = d.oc_close;
        a.close_count = reinterpret_cast<int*>(d.oc_close + B);
        a.close_ctrl = d.ctrl;
        a.close_mirror = static_cast<CgCtrl*>(pinned_status_block());
        a.close_ticket = close_ticket = next_ticket();
        a.close_tol = d.tol;
        a.close_floor_ok = (a.iters - 1 >= std::min(10, d.max_iter - 1)) ? 1 : 0;
      }
      rc = onchip5_launch(pl.R4, a, oc_nwg, st);  // (d.oc_gbuf was cleared together with the control block)
      if (rc != LO_OK) {
        close_in_kernel = false;
        a.close_gran = nullptr;
      }
      if (rc == LO_OK) {
        serial_done = true;
        exec.serial_engine = LO_ENGINE_RESIDENT_ROOT;
        exec.serial_group = a.GW;
      }
      else if (rc == LO_ERR_UNSUPPORTED) rc = LO_OK;  // (does not fit: the Q-form kernels below)
    }
    if (rc == LO_OK && ls_cols < c && !serial_done && pre && !pre->Q) rc = LO_ERR_UNSUPPORTED;  // (root form only)
    if (rc == LO_OK && ls_cols < c && !serial_done) {  // second (first) generation: columns [ls_cols, c)
      if (lean) {  // (the root-form kernel did not fit after all; these kernels have no result-only mode: start over)
        lean = false;
        oc_redo = true;
      }
    }
    if (rc == LO_OK && ls_cols < c && !serial_done && !oc_redo) {
      lean_state(false);
      const bool gen2 = onchip4_eligible(pl.R4, ocR4, N, c - ls_cols);
      a.GW = gen2 ? onchip4_group_size(N) : 8;
      a.RW = (int)((N + a.GW - 1) / a.GW);
      a.col0 = ls_cols; a.ncols = c - ls_cols;
      a.xout = gen2 ? x : nullptr;
      a.gbuf = d.oc_gbuf; a.next_member = d.oc_err + 1;
      a.dbg = oc_dbg ? d.oc_dbg : nullptr;
      LO_HIP_CHECK(hipMemsetAsync(d.oc_gbuf, 0, onchip_gbuf_bytes(66), st));  // whole allocation
      rc = LO_ERR_UNSUPPORTED;
      if (gen2) rc = onchip4_launch(pl.R4, ocR4, a, oc_nwg, st);
      if (gen2 && rc == LO_OK) {
        exec.serial_engine = LO_ENGINE_RESIDENT_GEN2;
        exec.serial_group = a.GW;
      }
      xout_ok = gen2 && rc == LO_OK;
    }
    if (rc && rc != LO_ERR_UNSUPPORTED) return rc;
    if (rc == LO_OK && !oc_redo) {  // (LO_ERR_UNSUPPORTED: no resident kernel fits this device / shape -> streaming engine)
      const int ktri = prm->n_tridiag ? std::min(a.iters, (int)prm->max_tridiag_iter) : 0;
      const unsigned tri_grid = (unsigned)(((size_t)B * std::max(1, (int)prm->n_tridiag) + kThreads - 1) / kThreads);
      if (ktri) {
        LO_HIP_CHECK(hipMemsetAsync(d.oc_maxoff, 0, sizeof(int) * (prm->max_tridiag_iter + 1), st));
        hipLaunchKernelGGL(k_oc_maxoff, dim3(tri_grid), block, 0, st, d, d.oc_ab, ktri);
      }
      // (with tridiagonals k_oc_tridiag follows and does not touch the control block: the mirror is final either way)
      CgCtrl* mirror = static_cast<CgCtrl*>(pinned_status_block());
      const unsigned ticket = close_in_kernel ? close_ticket : next_ticket();
      if (!close_in_kernel)
        hipLaunchKernelGGL(k_cg_ctrl_onchip, dim3(1), block, 0, st, d, d.oc_resid, d.oc_init_conv, a.iters, ktri, mirror,
                           ticket);
      if (ktri) hipLaunchKernelGGL(k_oc_tridiag, dim3(tri_grid), block, 0, st, d, d.oc_ab, ktri);
      LO_LAUNCH_CHECK();
      if (mirror) {
        // (with tridiagonals k_oc_tridiag is still queued behind the control kernel: drain the stream as before)
        if (ktri) LO_HIP_CHECK(hipStreamSynchronize(st));
        else if (wait_ticket(reinterpret_cast<volatile unsigned*>(mirror) + 63, ticket, st)) return LO_ERR_LAUNCH;
        memcpy(&h, mirror, sizeof(CgCtrl));
      } else {
        LO_HIP_CHECK(hipMemcpyAsync(&h, d.ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, st));
        LO_HIP_CHECK(hipStreamSynchronize(st));
      }
      const int oc_err = h.oc_err;
      if (oc_err == 0) {
        rc = global_check();
        if (rc) return rc;
      }
      if (ls_dbg) {
        long long ts[10];
        LO_HIP_CHECK(hipMemcpy(ts, d.oc_dbg, sizeof(ts), hipMemcpyDeviceToHost));
        fprintf(stderr, "lockstep item (100 MHz ticks): load+H %lld init %lld iters %lld store %lld | direction %lld "
                "alpha+x+r %lld reduce-mfma %lld cross-wave %lld group-sum %lld\n",
                ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5], ts[6], ts[7], ts[8], ts[9]);
      }
      if (oc_dbg) {
        long long ts[12];
        LO_HIP_CHECK(hipMemcpy(ts, d.oc_dbg, sizeof(ts), hipMemcpyDeviceToHost));
        if (ts[8] || ts[9])
          fprintf(stderr, "  root-form kernel phases: updates %lld, w partials + all-reduce %lld, first-wave algebra %lld\n",
                  ts[8], ts[9], ts[10]);
        fprintf(stderr, "onchip member0 (100 MHz ticks): load %lld init %lld iters %lld store %lld | wg-wait %lld publish %lld poll %lld\n",
                ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5], ts[6], ts[7]);
      }
      if (oc_err == 0 && lean && h.rs_redo && tls_rspace_diag_ran && !rs_force_dense) {
        rs_force_dense = true;  // (the result-only pass again, on the dense R-space form)
        oc_redo = true;
      } else if (oc_err == 0 && lean && !h.stop) {
        // the stop rule does not hold at the floor and the state was not written: the same launches once more, in full
        // (a root-form-only preconditioner cannot continue on the streaming engine anyway: the caller builds Q first)
        lean_miss_note(op, prm, pre ? pre->generation : 0);
        if (pre && !pre->Q) return LO_ERR_UNSUPPORTED;
        lean = false;
        oc_redo = true;
        if (rs_cols_ran) ls_cols = ls_plan;  // (the repeat runs the engines of the plan)
      } else if (oc_err == 0) {
        k_start = a.iters;
        resident_note_ok();
        if (lean_skipped && h.stop) tls_lean_miss[miss_slot].valid = 0;  // (it stops at the floor now: speculate again)
        exec.resident = 1;
        exec.lockstep_cols = ls_cols;
        exec.lean = lean ? 1 : 0;
        exec.rspace = rs_cols_ran ? 1 : ((lean && tls_rspace_resident_ran) ? 2 : 0);
        exec.reserved2 = (exec.rspace == 2 && tls_rspace_diag_ran) ? 1 : 0;  // 1: the diagonal form of the R-space iteration ran
        x_written = xout_ok;
      } else {  // a group hand-off timed out: redo everything with the streaming engine
        fprintf(stderr, "liblo_amd: operator-resident CG timed out, falling back to the streaming engine\n");
        onchip_note_timeout();
        LO_HIP_CHECK(hipMemsetAsync(d.ctrl, 0, sizeof(CgCtrl), st));
        memset(&h, 0, sizeof(h));
      }
    }
  }
  if (!oc_redo) break;
  {  // second pass: control block and granules as at the start of the solve
    const size_t span = (size_t)(reinterpret_cast<char*>(d.oc_close + B + 2) - reinterpret_cast<char*>(d.ctrl));
    rc = zero_span(d.ctrl, span, st);
    if (rc) return rc;
    memset(&h, 0, sizeof(h));
  }
  }  // oc_pass
  // a root-form-only preconditioner cannot feed the streaming engine (initial run, redo after a timeout, or the
  // continuation beyond the resident iterations): the caller builds the Q form and calls again
  if (pre && !pre->Q && (k_start == 0 || !h.stop)) return LO_ERR_UNSUPPORTED;
  int matvecs = 0;
  if (k_start == 0) {
    // ---- initialisation (linear_cg.py:177-215) ----
    rc = vec_dot_part(rhs, rhs, c, d.pAp_part, B, N, sp, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cg_init_scal, dim3(1), block, 0, st, d, d.pAp_part);
    hipLaunchKernelGGL(k_cg_init_vec, gridv, block, 0, st, d, rhs, x0, sp.rows);
    LO_LAUNCH_CHECK();
    if (x0) {
      rc = matvec_run(&pl, d.x, d.Ap, nullptr, nullptr, st);
      if (rc) return rc;
      ++matvecs;
      const size_t nv = (size_t)B * N * c;
      hipLaunchKernelGGL(k_cg_sub, dim3((unsigned)std::min<size_t>((nv + 255) / 256, 8192)), block, 0, st, d.r, d.Ap, nv);
      LO_LAUNCH_CHECK();
    }
    rc = vec_dot_part(d.r, d.r, c, d.rr_part, B, N, sp, nullptr, st);
    if (rc) return rc;
    if (precond) {
      rc = apply_precond(d.r, d.z, d.rz_part);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_cg_ctrl_init, dim3(1), block, 0, st, d);
    LO_LAUNCH_CHECK();
  }  // k_start == 0

  // ---- iterations ----
  const float* zsrc = precond ? d.z : d.r;
  const int kfloor = std::min(10, fmi - 1);
  int first_poll = kfloor;
  if (prm->n_tridiag) first_poll = std::max(first_poll, std::min(prm->max_tridiag_iter, fmi - 1));
  const bool opaque = (op->kind == LO_OP_CALLBACK) || (precond_cb != nullptr);
  const int chunk = plan.poll_chunk;
  auto poll = [&]() -> int {
    void* hp = pinned_status_block();
    LO_HIP_CHECK(hipMemcpyAsync(hp ? hp : &h, d.ctrl, sizeof(CgCtrl), hipMemcpyDeviceToHost, st));
    LO_HIP_CHECK(hipStreamSynchronize(st));
    if (hp) memcpy(&h, hp, sizeof(CgCtrl));
    return LO_OK;
  };
  if (opaque || prm->max_iter == 0) {  // closures cannot see the stop word: look before the first product
    rc = poll();
    if (rc) return rc;
  }
  const int ctrl_G = (int)std::min<int64_t>(kCtrlMaxG, ((int64_t)B * c + kThreads - 1) / kThreads);
  int k = k_start;
  int launched = k_start;
  // large-N single-column solves: the preconditioner apply fused with the r / x update, Q read once per iteration
  bool pf_on = d.pf_gbuf && (plan.streaming_precond == LO_STREAM_PRE_FUSED_Q ||
                             plan.streaming_precond == LO_STREAM_PRE_FUSED_KRON);
  // Kronecker operator with a constant diagonal: the rows of the preconditioner's tall matrix are formed on the fly from
  // the pivot rows of the two factors (lo_precond_desc.kron_*): 7 instead of 23 floats of traffic per row and iteration
  PfKron kron{nullptr, nullptr, nullptr, 0, 0};
  const bool pf_kron = pf_on && plan.streaming_precond == LO_STREAM_PRE_FUSED_KRON;
  if (pf_kron) kron = PfKron{pre->kron_a, pre->kron_b, pre->kron_F, (int)op->R, (int)op->n2};
  bool p_done = false;  // the fused apply of the previous iteration already wrote this iteration's p
  // up to 32 columns, up to 16384 rows: alpha, r / x, the preconditioner, beta, p and the control step in one launch
  bool sc_on = d.sc_gbuf && (plan.streaming_precond == LO_STREAM_PRE_FUSED_COLS ||
                            plan.streaming_precond == LO_STREAM_NOPRE_FUSED_COLS);
  const bool sc_dbg = sc_on && getenv("LO_SC_DEBUG") != nullptr;
  if (sc_dbg) LO_HIP_CHECK(hipMemsetAsync(d.oc_dbg, 0, 16 * sizeof(long long), st));
  // (a solve the resident phase has already closed launches no streaming iteration: its three clears -- 3 hipMemsetAsync =
  //  4 fill kernels, ~15 us of stream time behind every headline solve -- are skipped)
  const bool will_stream = k_start < prm->max_iter && !h.stop;
  if (sc_on && will_stream) {  // cleared once per solve (tags are unique per launch)
    LO_HIP_CHECK(hipMemsetAsync(d.sc_gbuf, 0, cg_step_cols_gbuf_bytes(), st));
    LO_HIP_CHECK(hipMemsetAsync(d.sc_ctr, 0, sizeof(int) * 2 * ((size_t)std::max(1, (int)prm->max_iter) + 1), st));
    LO_HIP_CHECK(hipMemsetAsync(d.sc_gran, 0, sizeof(unsigned long long) * 3 * (size_t)B, st));
  }
  if (pf_on && will_stream) {  // granules and hand-out counters of the fused apply: cleared once per solve (tags are unique per launch)
    LO_HIP_CHECK(hipMemsetAsync(d.pf_gbuf, 0, precond_fused_gbuf_bytes(), st));
    LO_HIP_CHECK(hipMemsetAsync(d.pf_ctr, 0, sizeof(int) * 2 * ((size_t)std::max(1, (int)prm->max_iter) + 1), st));
    LO_HIP_CHECK(hipMemsetAsync(d.pf_gran, 0, sizeof(unsigned long long) * (size_t)B, st));
  }
  // One iteration's launches on stream `ls`.  dyn: the kernels that need the iteration index read it from the control
  // block (the same launch sequence is then valid for every later iteration: it is captured once and replayed as a
  // hipGraph -- the iterations of small / single operators are launch-bound: ~100 us of kernels, ~50 us of gaps).
  auto issue = [&](int kk, bool dyn, hipStream_t ls) -> int {
    int rcb;
    if (p_done) {
      rcb = matvec_run(&pl, d.p, d.Ap, d.pAp_part, stop, ls);
      if (rcb) return rcb;
    } else if (matvec_can_fuse_pupdate(&pl)) {
      rcb = matvec_run_pupdate(&pl, d.p, zsrc, d.beta, kk == 0 ? 1 : 0, d.Ap, d.pAp_part, stop, ls);
      if (rcb) return rcb;
    } else {
      LO_PROF_BEGIN("cg_update_p", ls);
      hipLaunchKernelGGL(k_cg_update_p, gridv, block, 0, ls, d, zsrc, kk == 0 ? 1 : 0, sp.rows);
      LO_PROF_END(ls);
      LO_LAUNCH_CHECK();
      rcb = matvec_run(&pl, d.p, d.Ap, d.pAp_part, stop, ls);
      if (rcb) return rcb;
    }
    bool pre_done = false;
    bool ctrl_done = false;  // the fused apply also took the iteration's control step
    if (pre && pf_on) {
      PfCtrl cf;
      cf.on = (d.n_tridiag == 0 && kk > 0 && !getenv("LO_NO_FUSED_CTRL")) ? 1 : 0;
      cf.rhs_is_zero = d.rhs_is_zero; cf.rz = d.rz; cf.beta = d.beta; cf.resid_norm = d.resid_norm;
      cf.has_conv = d.has_conv; cf.stop_after = d.stop_after; cf.tol = d.tol;
      cf.kfloor = std::min(10, d.max_iter - 1);
      cf.done = d.pf_ctr + (std::max(1, (int)prm->max_iter) + 1);
      cf.ctrl = d.ctrl;
      cf.gran = d.pf_gran;
      // single pass over Q: r / x update, Q^T r, group all-reduce, z = r/d - Q u, p = z + beta p (lo_precond_fused.hip)
      rcb = precond_fused_rupdate(Qp, pre->dinv, pre->constant_diag ? LO_DIAG_CONST : LO_DIAG_FULL, d.r, d.Ap, d.p, d.x,
                                  d.z, d.pAp_part, d.S_dot, d.rz, d.has_conv, d.eps, d.alpha, d.rr_part, d.rz_part, sp.S,
                                  B, N, d.pf_gbuf, d.oc_err, d.pf_ctr, kk, dyn ? &d.ctrl->iterations : nullptr,
                                  (int)prm->max_iter, stop, oc_nwg, &cf, pf_kron ? &kron : nullptr, ls);
      if (rcb == LO_ERR_UNSUPPORTED) {  // (does not fit this device: the two-launch path from now on)
        pf_on = false;
        exec.streaming_precond = LO_STREAM_PRE_TWO_PASS;
      }
      else if (rcb) return rcb;
      else {
        pre_done = p_done = true;
        ctrl_done = cf.on != 0;
      }
    }
    if (!pre_done && sc_on) {
      if (dyn) return LO_ERR_UNSUPPORTED;  // (not replayed as a graph node)
      ScCtrl cf;
      cf.on = getenv("LO_NO_FUSED_CTRL") ? 0 : 1;
      cf.rhs_is_zero = d.rhs_is_zero; cf.beta = d.beta; cf.resid_norm = d.resid_norm;
      cf.stop_after = d.stop_after; cf.tol = d.tol; cf.max_iter = d.max_iter;
      cf.n_tridiag = d.n_tridiag; cf.n_tridiag_iter = d.n_tridiag_iter; cf.T = d.T; cf.t_mat = d.t_mat;
      cf.prev_ar = d.prev_ar; cf.prev_beta = d.prev_beta; cf.check_nan_first = d.check_nan_first;
      cf.done = d.sc_ctr + (std::max(1, (int)prm->max_iter) + 1);
      cf.gran = d.sc_gran;
      cf.ctrl = d.ctrl;
      rcb = cg_step_cols(pre ? Qp : nullptr, preR4, pre ? pre->dinv : nullptr,
                         (pre && pre->constant_diag) ? LO_DIAG_CONST : LO_DIAG_FULL, d.r, d.Ap, d.p, d.x, c, d.pAp_part,
                         d.S_dot, d.rz, d.has_conv, d.eps, d.alpha, d.rr_part, d.rz_part, sp.S, B, N, d.sc_gbuf, d.oc_err,
                         d.sc_ctr, kk, (int)prm->max_iter, stop, oc_nwg, &cf, sc_dbg ? d.oc_dbg : nullptr, ls);
      if (rcb == LO_ERR_UNSUPPORTED) {  // (does not fit this device: the multi-launch path from now on)
        sc_on = false;
        exec.streaming_precond = pre ? LO_STREAM_PRE_TWO_PASS : LO_STREAM_PRE_NONE;
      } else if (rcb) {
        return rcb;
      } else {
        pre_done = p_done = true;
        ctrl_done = cf.on != 0;
        // LO_SC_TEST_FALLBACK=<k>: the error word set behind launch k, as if a hand-off had timed out (host redo path)
        if (const char* e = getenv("LO_SC_TEST_FALLBACK"))
          if (atoi(e) == kk) LO_HIP_CHECK(hipMemsetAsync(d.oc_err, 1, 1, ls));
      }
    }
    if (!pre_done) p_done = false;  // (the two-launch path below leaves the p update to the next iteration's first step)
    if (pre_done) {
    } else if (pre) {
      // r-update, x-update and the residual norm ride on the first pass over Q
      rcb = skinny_tn_rupdate(Qp, preR4, preR4, d.r, d.Ap, d.p, d.x, d.pAp_part, d.S_dot, d.rz, d.has_conv, d.eps,
                              d.alpha, d.rr_part, c, upart, B, N, sp, stop, ls);
      if (rcb) return rcb;
      rcb = skinny_nn(Qp, preR4, preR4, upart, pre->dinv, pre->constant_diag ? LO_DIAG_CONST : LO_DIAG_FULL, -1.0f, d.r,
                      c, d.z, d.rz_part, B, N, sp, stop, ls);
      if (rcb) return rcb;
    } else {
      LO_PROF_BEGIN("cg_update_xr", ls);
      hipLaunchKernelGGL(k_cg_update_xr, gridv, block, 0, ls, d, sp.rows);
      LO_PROF_END(ls);
      LO_LAUNCH_CHECK();
      if (precond) {
        if (dyn) return LO_ERR_UNSUPPORTED;  // (closure preconditioners are not replayed)
        rcb = apply_precond(d.r, d.z, d.rz_part);
        if (rcb) return rcb;
      }
    }
    if (ctrl_done) return LO_OK;
    const int karg = dyn ? -1 : kk;
    LO_PROF_BEGIN("cg_ctrl", ls);
    if (ctrl_G == 1) {
      hipLaunchKernelGGL(k_cg_scal_ctrl, dim3(1), block, 0, ls, d, karg);
    } else {
      hipLaunchKernelGGL(k_cg_scal, dim3(ctrl_G), block, 0, ls, d, karg);
      hipLaunchKernelGGL(k_cg_ctrl, dim3(1), block, 0, ls, d, karg, ctrl_G);
    }
    LO_PROF_END(ls);
    LO_LAUNCH_CHECK();
    return LO_OK;
  };
  GraphReplay gr;  // (destroys the graph objects on every exit path)
  // MEASURED SLOWER on ROCm 7.0 / MI355X (cfg4 shard 39.7 vs 38.2 ms, one dense GP of N = 4000 10.4 vs 9.8 ms: the
  // replayed nodes start no closer together than individually launched kernels and the instantiation costs ~0.3 ms):
  // opt-in only (LO_CG_GRAPH=1).
  bool graph_ok = getenv("LO_CG_GRAPH") && !g_prof_on && !opaque && !precond_cb && op->kind != LO_OP_CALLBACK;
  while (k < prm->max_iter && !h.stop) {
    bool ran = false;
    if (gr.exec) {
      ran = hipGraphLaunch(gr.exec, st) == hipSuccess;
      if (!ran) {
        (void)hipGetLastError();
        gr.reset();
        graph_ok = false;
      }
    } else if (graph_ok && k >= k_start + 2 && prm->max_iter - k >= 8) {
      // steady state reached (first-iteration flags gone, p_done settled): capture this iteration on a side stream
      // (the caller's may be the legacy default stream, which cannot be captured) and replay it on the caller's
      const bool pf0 = pf_on, pd0 = p_done;
      if (gr.begin()) {
        tls_graph_capture = true;
        const int rcb = issue(k, true, gr.side);
        tls_graph_capture = false;
        const bool ok = gr.end() && rcb == LO_OK && pf_on == pf0 && p_done == pd0;
        if (ok && hipGraphLaunch(gr.exec, st) == hipSuccess) {
          ran = true;
        } else {
          (void)hipGetLastError();
          gr.reset();
          graph_ok = false;
          pf_on = pf0;
          p_done = pd0;
        }
      } else {
        graph_ok = false;
      }
    }
    if (!ran) {
      rc = issue(k, false, st);
      if (rc) return rc;
    }
    ++launched;
    const bool at_poll = (k >= first_poll) && (((k - first_poll) % chunk) == 0);
    if (at_poll || k == prm->max_iter - 1 || (opaque && k == 0)) {
      rc = poll();
      if (rc) return rc;
      if ((pf_on || sc_on) && h.oc_err) break;  // a hand-off of the fused apply timed out: redo below
      if (k >= first_poll) {  // (one collective per iteration from the first possible stop on, on every rank)
        rc = global_check();
        if (rc) return rc;
      }
    }
    ++k;
  }
  if (launched == 0 || !h.stop) {
    rc = poll();
    if (rc) return rc;
  }
  if ((pf_on || sc_on) && h.oc_err) {  // (co-residency lost -- never seen on a dedicated GPU): the two-launch path for the whole solve
    fprintf(stderr, "liblo_amd: fused preconditioner apply timed out, redoing the solve with the two-pass kernels\n");
    tls_no_fused_precond = true;
    const int rc2 = lo_cg_solve_f32(op, matvec, matvec_user, pre, precond_cb, precond_user, prm, rhs, x0, x, t_mat, ws,
                                    ws_bytes, info, stream);
    tls_no_fused_precond = false;
    return rc2;
  }
  if (!(x_written && launched == k_start)) {  // (no streaming iteration after the resident phase: x is final already)
    hipLaunchKernelGGL(k_cg_final, gridv, block, 0, st, d, x, sp.rows);
    LO_LAUNCH_CHECK();
    LO_HIP_CHECK(hipStreamSynchronize(st));
  }

  if (sc_dbg) {
    long long ts[8];
    LO_HIP_CHECK(hipMemcpy(ts, d.oc_dbg, sizeof(ts), hipMemcpyDeviceToHost));
    if (ts[5])
      fprintf(stderr, "cg_step_cols member 0, first workgroup (100 MHz ticks per launch): loads+alpha %.1f update+mfma %.1f "
              "all-reduce %.1f beta+control %.1f p %.1f\n", (double)ts[0] / ts[5], (double)ts[1] / ts[5], (double)ts[2] / ts[5],
              (double)ts[3] / ts[5], (double)ts[4] / ts[5]);
  }
  info->iterations = h.iterations;
  info->matvecs = matvecs + h.iterations;
  info->tolerance_reached = h.tol_reached;
  info->nan_detected = h.nan_detected;
  info->skipped = h.skipped;
  info->last_tridiag_iter = h.last_tridiag_iter;
  info->mean_residual = h.mean_resid;
  info->reserved = 0.f;
  if (!exec.resident) {
    exec.resident_iterations = exec.lockstep_cols = exec.lockstep_group = exec.serial_group = exec.lean = 0;
    exec.serial_engine = LO_ENGINE_NONE;
  }
  exec.reserved = launched - k_start;  // streaming iterations enqueued after the resident phase
  tls_last_exec = exec;
  return LO_OK;
}

// The engine selection of lo_cg_solve_f32 for these arguments (pure; see cg_plan).  cus <= 0: the current device (with
// LO_OC_RESERVE_CUS applied), else plan for a device with that many compute units.
int lo_cg_plan_f32(const lo_op_desc* op, const lo_precond_desc* pre, int has_precond_cb, int has_x0,
                   const lo_cg_params* prm, int cus, lo_cg_plan* plan) {
  if (!op || !prm || !plan) return LO_ERR_BADARG;
  if (prm->c < 1 || prm->c > kMaxCols) return LO_ERR_UNSUPPORTED;
  if (pre && has_precond_cb) return LO_ERR_BADARG;
  const int nwg = cus > 0 ? (cus / 32) * 32 : onchip_num_workgroups();
  cg_plan(op, pre, has_precond_cb != 0, has_x0 != 0, prm, nwg, plan);
  return LO_OK;
}

// What the calling thread's last successful lo_cg_solve_f32 launched: the plan after its run-time fall-backs
// (`reserved` = number of streaming iterations enqueued after the resident phase).
int lo_cg_last_executed(lo_cg_plan* out) {
  if (!out) return LO_ERR_BADARG;
  *out = tls_last_exec;
  return LO_OK;
}

// development / test switch: force the streaming engine (0) or allow the operator-resident fast path (1)
int lo_cg_set_onchip(int enable) {
  g_onchip_disabled = (enable == 0);
  if (enable) {  // (also ends a cool-down)
    g_res_cooldown.store(0, std::memory_order_relaxed);
    g_res_backoff.store(kResidentBackoff0, std::memory_order_relaxed);
  }
  for (int i = 0; i < 16; ++i) tls_lean_miss[i].valid = 0;  // (also forgets which operators missed their result-only pass)
  return LO_OK;
}

// The gate of the resident kernels (see the top of this file).
int lo_resident_status_get(lo_resident_status* out) {
  if (!out) return LO_ERR_BADARG;
  out->user_disabled = g_onchip_disabled ? 1 : 0;
  out->timeouts = g_res_timeouts.load(std::memory_order_relaxed);
  out->cooldown = g_res_cooldown.load(std::memory_order_relaxed);
  out->backoff = g_res_backoff.load(std::memory_order_relaxed);
  out->rearms = g_res_rearms.load(std::memory_order_relaxed);
  out->fused_timeouts = g_onchip_fused_timeouts;
  return LO_OK;
}
int lo_resident_inject_timeouts(int32_t n) {
  g_res_inject.store(n < 0 ? 0 : n, std::memory_order_relaxed);
  return LO_OK;
}

// z = P^{-1} r as a standalone call (precondition_closure, added_diag_linear_operator.py:135-140)
size_t lo_precond_apply_workspace_bytes(int64_t B, int64_t N, int32_t k, int64_t c) {
  Split sp = choose_split(B, N, 256);
  const int R4 = padded_rank_k(k);
  return align_up((size_t)B * sp.S * R4 * c * sizeof(float), 256) + align_up((size_t)B * N * R4 * sizeof(float), 256) +
         1024;
}

int lo_precond_apply_f32(const lo_precond_desc* pre, const float* r, float* z, int64_t B, int64_t N, int64_t c, void* ws,
                         size_t ws_bytes, void* stream) {
  if (!pre || !r || !z || !ws) return LO_ERR_BADARG;
  hipStream_t st = (hipStream_t)stream;
  Split sp = choose_split(B, N, 256);
  const int R4 = padded_rank_k(pre->k);
  Arena ar(ws, ws_bytes);
  float* upart = ar.take<float>((size_t)B * sp.S * R4 * c);
  const float* Qp = pre->Q;
  if (pre->ldq != R4) {
    if (pre->ldq != pre->k) return LO_ERR_BADARG;
    float* qp = ar.take<float>((size_t)B * N * R4);
    if (!ar.ok) return LO_ERR_WORKSPACE;
    int rc = pad_rows(pre->Q, pre->k, qp, R4, B * N, st);
    if (rc) return rc;
    Qp = qp;
  }
  if (!ar.ok) return LO_ERR_WORKSPACE;
  int rc = skinny_tn(Qp, R4, R4, r, c, upart, B, N, sp, nullptr, st);
  if (rc) return rc;
  return skinny_nn(Qp, R4, R4, upart, pre->dinv, pre->constant_diag ? LO_DIAG_CONST : LO_DIAG_FULL, -1.0f, r, c, z,
                   nullptr, B, N, sp, nullptr, st);
}

}  // extern "C"
