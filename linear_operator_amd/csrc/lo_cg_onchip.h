// lo_cg_onchip.h -- argument block of the operator-resident CG kernel (lo_cg_onchip.hip)
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace lo {

struct CgCtrl;

struct OnchipArgs {
  const float* C;     // [B, N, RC]   (padded rank RC)
  const float* Q;     // [B, N, RK]
  const float* d;     // [B, N] or [B]
  const float* dinv;  // [B, N] or [B]
  int d_mode, dinv_mode;
  const float* rhs;   // [B, N, c]
  int c;              // right-hand-side columns of the vectors (row stride; first generation: 1)
  int col0, ncols;    // columns [col0, col0 + ncols) are solved by this launch (second / third generation)
  int RK;             // floats per row of Q (third generation; the others take it as a template parameter)
  int RCg;            // floats per row of C in HBM (third generation: 8, 16 or 32)
  const float* F;     // root-form preconditioner (lo_precond_desc.F / EF), [B, RC, RC] each, or nullptr
  const float* EF;
  const float* E;     // C^T D^-1 C [B, RC, RC] (lo_precond_desc.E) or nullptr: enables the w-recurrence mode of k_cg_onchip5
  const double* RS;   // fp64 [B, 6, RC, RC]: E | F E | E F E | G2 = C^T C | F | E F (lo_precond_desc.RS) or nullptr: enables k_cg_rspace
  const double* RSD;  // fp64 [B, 6, RC, RC]: TinT | E^+ | TuT | G2 | Tin | lam (lo_precond_desc.RSD, lo_eigform.hip) or nullptr: the diagonal chain of k_cg_rspace
  float* ab_rec;      // [iters, B, c, 2] masked alpha / beta per iteration (second generation, n_tridiag > 0) or nullptr
  int64_t B;
  int N, RW;          // rows per workgroup
  int GW;             // workgroups per member (group size): 8 (first generation); 8, 16 or 32 (second)
  int iters;          // iterations to run (k = 0 .. iters-1)
  float eps, stop_after;
  // state out (streaming engine layout, c == 1)
  float *x, *r, *p, *z;
  float* xout;        // second generation: result.mul(rhs_norm) (linear_cg.py:335) written along with the state, or nullptr
  float *rhs_norm, *rz, *alpha, *beta, *resid_norm;
  int *rhs_is_zero, *has_conv;
  float* resid_rec;   // [iters, B, c] residual norm after each iteration (for the stop rule / NaN check)
  int* init_conv;     // [B, c] has_converged before the first iteration (linear_cg.py:205-208)
  unsigned long long* gbuf;  // [ngroups][2][8][40] granules
  int* err;
  int* next_member;   // second generation: shared counter of the dynamic member hand-out (zeroed by the host)
  int prefetch;          // first generation only: pull the next member's rows into L2 while iterating
  int allow_l2_handoff;  // 1: use the verified same-XCD L2 hand-off when the placement check passes
  // In-kernel closing step of k_cg_onchip5 (one column, no tridiagonals, this launch is the whole resident phase): every
  // member leaves {final residual norm | tag + flags} as one 8-byte granule, the workgroup that finishes last evaluates the
  // stop rule / NaN / skip conditions of k_cg_ctrl_onchip over them and mirrors the control block to the host -- the
  // separate control launch (4 us + a dependent-launch gap) disappears.  close_gran == nullptr: the host launches it.
  unsigned long long* close_gran;  // [B], zeroed together with the control block
  int* close_count;                // groups that have finished, zeroed together with the control block
  struct CgCtrl* close_ctrl;       // device control block
  struct CgCtrl* close_mirror;     // pinned host copy (or nullptr)
  unsigned close_ticket;
  float close_tol;                 // tolerance (< 0: the host decides -- batch-global rule over ranks)
  int close_floor_ok;              // iters - 1 >= min(10, max_iter - 1)
  long long* dbg;  // optional timestamps (wall_clock64) of member dbg_member / its workgroup 0, or nullptr
  int dbg_member;
};

int onchip4_launch(int RC, int RK, const OnchipArgs& a, int nwg, hipStream_t st);
// root-form serial-column kernel (k_cg_onchip5, lo_cg_onchip4.hip): one all-reduce per iteration
int onchip5_launch(int RC, const OnchipArgs& a, int nwg, hipStream_t st);
bool onchip5_eligible(int RC, int64_t N, int64_t c);
// the whole iteration on R + 1 coordinates (k_cg_rspace, lo_rspace.hip): one column, result only, one all-reduce per member
int rspace_launch(int RC, const OnchipArgs& a, int nwg, hipStream_t st);
// ... its diagonal form in the chunk-per-lane register layout (k_cg_rspace3, lo_rspace3.hip); called by rspace_launch
int rspace3_launch(int RC, const OnchipArgs& a, int nwg, hipStream_t st);
bool rspace_eligible(int RC, int64_t N, int64_t c);
size_t rspace_gbuf_bytes(int nworkgroups);
extern thread_local bool tls_rspace_diag_ran;      // ... and it ran the diagonal form (lo_precond_desc.RSD)
extern thread_local bool tls_rspace_resident_ran;  // set by rspace_launch when k_cg_rspace was launched (lo_cg_last_executed)
// all columns of a result-only solve in three streaming launches (k_rs_part / k_rs_iter / k_rs_apply, lo_rspace.hip)
int rspace_cols_launch(int RC, const OnchipArgs& a, double* ws, hipStream_t st);
bool rspace_cols_eligible(int RC, int64_t N, int64_t c);
size_t rspace_cols_ws_doubles(int64_t B, int64_t N, int RC, int c);
// third generation (lo_cg_lockstep.hip): 16 columns of a member advance together on the matrix cores
int lockstep_launch(int RC, bool pre, const OnchipArgs& a, int nwg, hipStream_t st);
bool lockstep_eligible(int RC, int RK, bool pre, int64_t N, int64_t ncols);
size_t lockstep_gbuf_bytes(int ngroups, int GW);
int lockstep_group_size(int64_t N);

}  // namespace lo
