// lo_internal.h -- shared declarations of liblo_amd's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lo_amd.h"

namespace lo {

constexpr int kThreads = 256;  // 4 wave64 per workgroup everywhere (vector / skinny kernels)
constexpr int kMaxCols = 256;  // a workgroup's thread->column map needs c <= 256
constexpr int kMaxRank = 256;  // padded root rank (floats per row) of a skinny operand

#define LO_HIP_CHECK(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      fprintf(stderr, "liblo_amd: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return LO_ERR_LAUNCH;                                                                 \
    }                                                                                       \
  } while (0)

#define LO_LAUNCH_CHECK() LO_HIP_CHECK(hipGetLastError())

// opt-in event timing of a launch (lo_prof.hip)
extern bool g_prof_on;
void prof_start(const char* name, hipStream_t st);
void prof_stop(hipStream_t st);
#define LO_PROF_BEGIN(name, st) \
  do {                          \
    if (lo::g_prof_on) lo::prof_start(name, st); \
  } while (0)
#define LO_PROF_END(st)                 \
  do {                                  \
    if (lo::g_prof_on) lo::prof_stop(st); \
  } while (0)

// Row split of one batch member over S workgroups (rows multiple of 4 except the tail).
struct Split {
  int S;
  int rows;
};
// (max_split: 256 unless a caller's kernels hold the partials of a member in a fixed-size buffer)
Split choose_split(int64_t B, int64_t N, int min_rows, int max_split = 256);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
  char* base;
  size_t cap, off;
  bool ok;
  Arena(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0), ok(true) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    size_t bytes = n * sizeof(T);
    T* r = (T*)(base ? base + off : nullptr);
    off += bytes;
    if (base && off > cap) ok = false;
    return r;
  }
};

// ---------------------------------------------------------------------------------------------
// device control block of one CG solve (lives in the workspace; host polls it)
struct CgCtrl {
  int stop;               // sticky: every kernel of the solve exits at once when set
  int iterations;         // loop bodies executed
  int tol_reached;
  int nan_detected;
  int skipped;
  int last_tridiag_iter;
  int tri_disabled;       // update_tridiag == False (linear_cg.py:326-327)
  float mean_resid;
  int oc_err;             // operator-resident kernels: a group exchange timed out (host falls back to streaming)
  int oc_next;            // operator-resident kernels: shared counter of the dynamic member hand-out
  int oc_next_ls;         // the same for the column-lockstep kernel (both may run in one solve)
  int pf_next;            // the same for the fused preconditioner apply of the streaming loop
  int rs_redo;            // diagonal form of the R-space iteration: a member's right-hand side lies (almost) inside span(C) --
                          // the host repeats the launch with the dense form (lo_rspace.hip)
};

// fused-update arguments of the skinny tn kernels (VMODE 1: p-update, VMODE 2: r/x-update; see lo_skinny.hip)
struct TnFuse {
  // VMODE 1
  const float* z;
  const float* beta;  // [B, ldv]
  int first;
  // VMODE 2
  const float* Ap;
  const float* p;
  float* x;
  const float* pAp_part;  // [B, S_dot, ldv]
  int S_dot;
  const float* rz;        // [B, ldv]
  const int* has_conv;    // [B, ldv]
  float eps;
  float* alpha_out;       // [B, ldv]
  float* rr_part;         // [B, S, ldv]
};

// matrix-core versions for 8 < c <= 32 (lo_skinny_mfma.hip); tpart layout [B,S,R4,c]
bool skinny_mfma_ok(int R4, int64_t c);
int skinny_tn_mfma(int vmode, const float* A, int R4, float* v, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
                   const TnFuse& f, const int* stop, hipStream_t st);
int skinny_nn_mfma(const float* A, int R4, const float* tpart, const float* dd, int dd_mode, float sgn, const float* v,
                   int64_t c, float* y, float* dot_part, int64_t B, int64_t N, Split sp, const int* stop,
                   hipStream_t st);

// ---- skinny operand kernels (lo_skinny.hip): A [B,N,lda] with R4 = 4*RQ padded columns --------
// tpart[B,S,R4,c] = A[rows_s]^T v[rows_s]
int skinny_tn(const float* A, int lda, int R4, const float* v, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
              const int* stop, hipStream_t st);
// y = sgn * A (sum_s tpart) + dd o v ; optional dot_part[B,S,c] = sum_rows v o y
int skinny_nn(const float* A, int lda, int R4, const float* tpart, const float* dd, int dd_mode, float sgn,
              const float* v, int64_t c, float* y, float* dot_part, int64_t B, int64_t N, Split sp, const int* stop,
              hipStream_t st);
// fused CG variants of skinny_tn (see lo_skinny.hip): the input vector is built on the fly
//   pupdate: p = z + beta p (first: p = z), written back, then tpart = A^T p
//   rupdate: masked alpha from pAp partials; r -= alpha Ap (written back); x += alpha p; rr_part = sum r^2; tpart = A^T r
int skinny_tn_pupdate(const float* A, int lda, int R4, float* p, const float* z, const float* beta, int first, int64_t c,
                      float* tpart, int64_t B, int64_t N, Split sp, const int* stop, hipStream_t st);
int skinny_tn_rupdate(const float* A, int lda, int R4, float* r, const float* Ap, const float* p, float* x,
                      const float* pAp_part, int S_dot, const float* rz, const int* has_conv, float eps,
                      float* alpha_out, float* rr_part, int64_t c, float* tpart, int64_t B, int64_t N, Split sp,
                      const int* stop, hipStream_t st);
// copy rows [B,N,R] -> zero padded [B,N,R4]
int pad_rows(const float* src, int R, float* dst, int R4, int64_t rows, hipStream_t st);

// ---- vector kernels (lo_vec.hip): vectors [B,N,c] ------------------------------------------------
int vec_dot_part(const float* a, const float* b, int64_t c, float* part, int64_t B, int64_t N, Split sp, const int* stop,
                 hipStream_t st);
// y = dd o v added onto y (dense / kron matvec epilogue): y += dd o v
int vec_add_diag(const float* dd, int dd_mode, const float* v, float* y, int64_t c, int64_t B, int64_t N, Split sp,
                 const int* stop, hipStream_t st);

// ---- operator matvec dispatch (lo_matvec.hip) ---------------------------------------------------
struct MatvecPlan {
  lo_op_desc op;
  int64_t c;
  Split sp;           // row split used by the skinny kernels / dot partials
  int S_dot;          // number of partials per (b,col) the plan writes into dot_part
  const float* Apad;  // padded copy of C when R % 4 != 0 (else op.A0)
  int lda, R4;
  float* tpart;       // [B,S,R4,c]
  bool mv_resident;   // shape the one-pass resident matvec takes (lo_lowrank_mv.hip); else the two-pass kernels
  float* kron_tmp;    // [B,N,c]
  float* dense_part;  // split-K partials of the dense matvec (small batches) or nullptr
  lo_matvec_cb cb;
  void* cb_user;
  // LO_OP_SUM: one sub-plan per term (host heap, released by matvec_plan_free) and the buffer a term beyond the
  // first is computed into before it is added onto y
  int nterms;
  MatvecPlan* sub;
  float* ytmp;        // [B,N,c]
};
void matvec_plan_free(MatvecPlan* pl);
// releases a plan's sub-plans on every exit path of the function that owns it
struct PlanGuard {
  MatvecPlan* p;
  explicit PlanGuard(MatvecPlan* q) : p(q) {
    p->sub = nullptr;
    p->nterms = 0;
  }
  ~PlanGuard() { matvec_plan_free(p); }
};
// y += a (elementwise, n floats)
int vec_axpy1(float* y, const float* a, size_t n, const int* stop, hipStream_t st);
// clears `bytes` at p (16-byte aligned) with one kernel launch (control blocks / hand-off granules of a resident launch)
int zero_span(void* p, size_t bytes, hipStream_t st);
size_t matvec_plan_bytes(const lo_op_desc* op, int64_t c, Split sp);
int matvec_plan_init(MatvecPlan* pl, const lo_op_desc* op, lo_matvec_cb cb, void* cb_user, int64_t c, Split sp,
                     Arena* ar, hipStream_t st);
// y = A v (+ optional dot partials sum_rows v o y, S_dot per (b,col))
int matvec_run(const MatvecPlan* pl, const float* v, float* y, float* dot_part, const int* stop, hipStream_t st);
// true if the plan can fuse the CG search-direction update into its first pass (low-rank operators)
bool matvec_can_fuse_pupdate(const MatvecPlan* pl);
// p = z + beta p (first: p = z); y = A p; dot partials
int matvec_run_pupdate(const MatvecPlan* pl, float* p, const float* z, const float* beta, int first, float* y,
                       float* dot_part, const int* stop, hipStream_t st);

// dense / kron kernels
int dense_matvec(const float* K, const float* d, int dd_mode, const float* v, float* y, float* dot_part, int64_t B,
                 int64_t N, int64_t c, int rows_per_wg, float* ypart, const int* stop, hipStream_t st);
int dense_rows_per_wg(int64_t B, int64_t N);
// number of dot partials per (member, column) the dense matvec writes for this shape (VALU vs MFMA tiling)
int dense_S_dot(int64_t B, int64_t N, int64_t c);
bool dense_mfma_ok(int64_t N, int64_t c);
int dense_mfma_tiles(int64_t N);
// ypart: dense_mfma_slices(B, N, c) * B * N * c floats for the split-K partials of small batches, or nullptr (no split)
int dense_mfma_slices(int64_t B, int64_t N, int64_t c);
int dense_matvec_mfma(const float* K, const float* d, int dd_mode, const float* v, float* y, float* dot_part, int64_t B,
                      int64_t N, int64_t c, float* ypart, const int* stop, hipStream_t st);
int kron_matvec(const float* K1, const float* K2, const float* v, float* tmp, float* y, int64_t B, int n1, int n2,
                int64_t c, const int* stop, hipStream_t st);
// matrix-core engine (c == 1, factors multiples of 128): diagonal term and CG dot partials fused in the epilogue
int kron_bilinear(const float* K1, const float* K2, const float* U, const float* V, float* tmp, float* dK1, float* dK2,
                  int64_t B, int n1, int n2, int64_t D, hipStream_t st);
bool kron_mfma_ok(int n1, int n2, int64_t c);
bool kron_mfma_cols_ok(int n1, int n2, int64_t c);  // c > 1 on the matrix-core engine (columns moved to the front)
int kron_matvec_mfma_cols(const float* K1, const float* K2, const float* diag, int diag_mode, const float* v,
                          float* buf_a, float* buf_b, float* y, int64_t B, int n1, int n2, int64_t c, const int* stop,
                          hipStream_t st);
int kron_S_dot(int n1, int n2, int64_t c, int S_default);
int kron_matvec_mfma(const float* K1, const float* K2, const float* diag, int diag_mode, const float* v, float* tmp,
                     float* y, float* dot_part, int64_t B, int n1, int n2, const int* stop, hipStream_t st);

// ---- one-pass operator-resident matvec of low-rank + diagonal members (lo_lowrank_mv.hip) ------------------------
// y = C (C^T v) + d o v with C read from HBM once (rows wait in registers for the group all-reduce of t = C^T v).
// R4 = padded rank 8 / 16 / 32, c <= 4 columns, N <= 32768.  LO_ERR_UNSUPPORTED: the caller runs the two-pass kernels.
bool lowrank_mv_eligible(int R4, int64_t N, int64_t c);
int lowrank_mv_run(const float* C, int R4, const float* d, int d_mode, const float* v, float* y, int64_t B, int64_t N,
                   int64_t c, const int* stop, hipStream_t st);

// ---- operator-resident CG (lo_cg_onchip.hip) -----------------------------------------------------
struct OnchipArgs;
size_t onchip_gbuf_bytes(int ngroups);
// fp64 Gram partials of a root on the fp64 matrix cores (lo_rspace.hip): E = C^T diag(dinv) C (dinv_full != nullptr) and C^T C
void rs_gram64_launch(const float* C, const float* dinv_full, int64_t B, int64_t N, int R, Split sp, double* gpartE,
                      double* gpart2, hipStream_t st);
int onchip_num_workgroups();
int onchip_l2_handoff_allowed();  // same-XCD plain-store hand-off: verified architectures only (lo_cg_onchip.hip)
bool onchip4_eligible(int RC, int RK, int64_t N, int64_t c);  // lo_cg_onchip4.hip
int onchip4_group_size(int64_t N);
int onchip5_group_size(int64_t N);
// Kernels whose workgroups wait for each other need ALL of them resident: two such kernels on two streams of one
// process would each take part of the CUs and spin until the hand-off timeout.  Construct a ResidentLaunch right before
// such a launch (same scope): if the previous resident kernel of this process went to a DIFFERENT stream, the new stream
// is made to wait for everything submitted to that stream so far; host threads are serialised for the duration of the
// launch call.  No cost when all solves use one stream.
struct ResidentLaunch {
  explicit ResidentLaunch(hipStream_t st);
  ~ResidentLaunch();
  ResidentLaunch(const ResidentLaunch&) = delete;
  ResidentLaunch& operator=(const ResidentLaunch&) = delete;
};
extern thread_local bool tls_graph_capture;  // a CG iteration is being captured: no cross-stream event traffic
extern bool g_onchip_disabled;  // lo_cg_set_onchip(0): streaming engines only (tests compare the two)
extern int g_onchip_fused_timeouts;  // group exchanges of the fused solve that timed out in this process
// A REAL hand-off timeout (co-residency lost: another process / kernel holds part of the CUs) sends the next calls to
// the streaming engines for a cool-down (16 calls, doubling up to 4096 while the timeouts repeat), after which the
// resident kernels are tried again; lo_cg_set_onchip(1) ends a cool-down at once (lo_cg.hip, lo_resident_status).
// The injected timeout of LO_OC_TEST_FALLBACK does not start a cool-down; lo_resident_inject_timeouts does.
void onchip_note_timeout();
bool resident_off();            // user switch or cool-down: no resident kernel for this call
void resident_tick();           // entry points: one call of the cool-down served
void resident_note_ok();        // a resident solve completed: the next cool-down is short again
bool resident_take_injection(); // lo_resident_inject_timeouts: treat this resident launch as timed out
// Pinned host landing zone for the small status read-backs of the synchronous entry points (a device-to-host copy into
// pageable memory is staged through an internal buffer and costs tens of microseconds more): one 256-byte block per
// host thread, allocated on first use; nullptr if pinned memory is unavailable (callers then copy to their stack).
void* pinned_status_block();
// hipOccupancyMaxActiveBlocksPerMultiprocessor costs microseconds per call and its answer for a given kernel, block
// size and dynamic LDS size never changes within a process: asked once per call site (one site per instantiation)
#define LO_OCCUPANCY_CACHED(out_int, kernel, tpb, dyn_lds)                                                   \
  ([&]() -> hipError_t {                                                                                     \
    static int cached = -1;                                                                                  \
    static hipError_t cached_err = hipSuccess;                                                               \
    if (cached < 0) {                                                                                        \
      int v = 0;                                                                                             \
      cached_err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, tpb, dyn_lds);                   \
      cached = (cached_err == hipSuccess) ? v : 0;                                                           \
    }                                                                                                        \
    (out_int) = cached;                                                                                      \
    return cached_err;                                                                                       \
  }())

// ---- fp64 helpers (lo_cg_f64.hip), shared by the fp64 CG and MINRES engines ---------------------------------------
// out1[b, j] = sum_i a[b, i, j] b1[b, i, j] (and out2 from (a2, b2) when a2 != nullptr)
int f64_dots(const double* a, const double* b1, double* out1, const double* a2, const double* b2, double* out2,
             int64_t B, int64_t N, int64_t c, hipStream_t st);
int f64_dense_mv(const double* A, const double* d, const double* v, double* y, int64_t B, int64_t N, int64_t c,
                 hipStream_t st);
int f64_copy(const double* a, double* o, size_t total, hipStream_t st);

// ---- single-pass Woodbury apply fused with the CG r / x update (lo_precond_fused.hip) --------------
bool precond_fused_eligible(int64_t B, int64_t N, int64_t c, int ldq, int S);
bool precond_fused_kron_eligible(int64_t n1, int64_t n2);  // factor sizes the Kronecker root-form kernel takes
size_t precond_fused_gbuf_bytes();
// the iteration's control step folded into the fused apply (single column, no tridiagonal, k >= 1)
struct PfCtrl {
  int on;
  const int* rhs_is_zero;  // [B]
  float* rz;               // [B] residual_inner_prod out
  float* beta;             // [B]
  float* resid_norm;       // [B]
  int* has_conv;           // [B]
  float stop_after, tol;
  int kfloor;              // min(10, max_iter - 1)
  int* done;               // base of one zeroed counter per launch of the solve
  unsigned long long* gran; // [B] tagged residual norms of the current launch (zeroed once per solve)
  CgCtrl* ctrl;
};
// Kronecker root form of the preconditioner (lo_precond_desc.kron_*): rows of "Q" formed on the fly (nullptr = Q form)
struct PfKron {
  const float* a;
  const float* b;
  const float* F;
  int n1, n2;
};
// (also writes the next iteration's p = z + beta p; z itself is not stored)
int precond_fused_rupdate(const float* Q, const float* dinv, int dinv_mode, float* r, const float* Ap, float* p,
                          float* x, float* z, const float* pAp_part, int S_dot, const float* rz, const int* has_conv,
                          float eps, float* alpha_out, float* rr_part, float* rz_part, int S, int64_t B, int64_t N,
                          unsigned long long* gbuf, int* err, int* next_member, int launch, const int* iter_ptr,
                          int max_launch, const int* stop, int ncu, const PfCtrl* cf, const PfKron* kr,
                          hipStream_t st);

// ---- the whole post-product step of a streaming CG iteration for up to 32 columns (lo_cg_step_cols.hip) ----------
// the iteration's control step (cg_scal_body + cg_ctrl_body of lo_cg.hip) folded into that launch
struct ScCtrl {
  int on;
  const int* rhs_is_zero;  // [B, c]
  float* beta;             // [B, c]
  float* resid_norm;       // [B, c]
  float stop_after, tol;
  int max_iter;            // the value the stop-rule floors are taken from
  int n_tridiag, n_tridiag_iter, T;
  float* t_mat;            // [n_tridiag, B, T, T]
  float* prev_ar;          // [B, n_tridiag]
  float* prev_beta;
  int check_nan_first;
  int* done;               // base of one zeroed counter per launch of the solve
  unsigned long long* gran; // [3, B] tagged per-member aggregates of the current launch (zeroed once per solve)
  CgCtrl* ctrl;
};
bool cg_step_cols_eligible(int64_t B, int64_t N, int64_t c, int ldq);  // ldq = 0: no preconditioner
size_t cg_step_cols_gbuf_bytes();
int cg_step_cols_group(int64_t N);
bool cg_step_cols_worthwhile(int64_t B, int64_t N, int64_t c, bool has_pre);
int cg_step_cols(const float* Q, int ldq, const float* dinv, int dinv_mode, float* r, const float* Ap, float* p, float* x,
                 int64_t c, const float* pAp_part, int S_dot, float* rz, int* has_conv, float eps, float* alpha_out,
                 float* rr_part, float* rz_part, int S, int64_t B, int64_t N, unsigned long long* gbuf, int* err,
                 int* next_member, int launch, int max_launch, const int* stop, int ncu, const ScCtrl* cf,
                 long long* dbg, hipStream_t st);

// ---- operator-resident pivoted Cholesky (lo_pivchol_onchip.hip) ----------------------------------
bool pc_onchip_eligible(const lo_op_desc* op, int max_rank);
size_t pc_onchip_workspace_bytes(const lo_op_desc* op, int max_rank);
// returns LO_ERR_LAUNCH when the group exchange timed out (caller falls back to the streaming engine)
int pc_onchip_run(const lo_op_desc* op, int rank, int max_rank, float tol, float* L_rows, long long* perm,
                  int32_t* rank_out, void* ws, size_t ws_bytes, hipStream_t st);

// the same for operators whose rows are fetched (dense, Kronecker of two dense factors): k_pc_onchip_rows
bool pc_onchip_rows_eligible(const lo_op_desc* op, int max_rank);
size_t pc_onchip_rows_workspace_bytes(const lo_op_desc* op, int max_rank);
int pc_onchip_rows_run(const lo_op_desc* op, int rank, int max_rank, float tol, float* L_rows, long long* perm,
                       int32_t* rank_out, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace lo
