/* lo_amd.h -- C ABI of liblo_amd.so: MI355X (gfx950) kernels for linear_operator's
 * iterative solve / logdet hot path.
 *
 * The reference (cornellius-gp/linear_operator) is pure Python and has no FFI; the seams this
 * library sits behind are the reference's own plug-in points:
 *   - linear_operator.utils.linear_cg          (linear_operator/utils/linear_cg.py:98-109, looked up
 *                                                at call time in operators/_linear_operator.py:796)
 *   - LinearOperator._matmul                    (operators/_linear_operator.py:169-190)
 *   - AddedDiagLinearOperator._preconditioner   (operators/added_diag_linear_operator.py:95-142)
 *   - PivotedCholesky.forward                   (functions/_pivoted_cholesky.py:14-105)
 *   - lanczos_tridiag / lanczos_tridiag_to_diag (utils/lanczos.py:9-189), StochasticLQ.to_dense
 *                                                (utils/stochastic_lq.py:45-82)
 * Each entry point below cites the reference interface it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - all tensors are DEVICE pointers, fp32 (`float`), row-major contiguous, in the reference's
 *     layouts: vectors [B, N, c] with the column index c innermost; matrices [B, rows, cols];
 *     permutations int64 [B, N].  B is the flattened batch (product of the reference's *batch dims).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous
 *     unless stated; the library never allocates device memory: the caller provides workspaces
 *     whose sizes come from the matching *_workspace_bytes query (host side: torch.empty).
 *   - return value: 0 = ok, <0 = LO_ERR_* (bad arguments, launch failure).  Numerical conditions
 *     (NaN in a matvec, non-convergence) are reported through lo_cg_info, and are mapped by the
 *     host shim to the reference's RuntimeError / NumericalWarning (linear_cg.py:199-200,337-347).
 *   - pointers are borrowed for the duration of the call only (asynchronous calls: until the
 *     stream has drained); inputs are never written (reference: linear_cg.py:182-190 allocates).
 */
#ifndef LO_AMD_H
#define LO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LO_OK 0
#define LO_ERR_BADARG (-1)
#define LO_ERR_LAUNCH (-2)
#define LO_ERR_WORKSPACE (-3)
#define LO_ERR_UNSUPPORTED (-4)

/* operator kinds: the `_matmul`s that feed CG */
#define LO_OP_LOWRANK_DIAG 0 /* AddedDiag(Root/LowRankRoot(C), Diag(d)):  y = C (C^T v) + d o v       */
#define LO_OP_DENSE_DIAG 1   /* AddedDiag(Dense(K), Diag(d)):             y = K v + d o v             */
#define LO_OP_KRON_DIAG 2    /* AddedDiag(Kron(K1,K2), Diag(d)):          y = (K1 (x) K2) v + d o v   */
#define LO_OP_CALLBACK 3     /* opaque closure (the reference's matmul_closure argument)              */
#define LO_OP_SUM 4          /* SumLinearOperator / PsdSum of up to LO_MAX_TERMS structured terms (+ one diagonal):
                              *   y = sum_i A_i v + d o v   (sum_linear_operator.py:47-51)              */
#define LO_MAX_TERMS 4

/* diagonal storage */
#define LO_DIAG_NONE 0  /* no diagonal term (plain Root / Dense / Kron operator)                    */
#define LO_DIAG_FULL 1  /* DiagLinearOperator._diag [B, N]         (diag_linear_operator.py:25)      */
#define LO_DIAG_CONST 2 /* ConstantDiagLinearOperator.diag_values [B] (diag_linear_operator.py:313)  */

/* Operator descriptor ("op-tree lowering" of an AddedDiag/Sum tree). */
typedef struct lo_op_desc {
  int32_t kind;      /* LO_OP_*                                                                    */
  int32_t diag_mode; /* LO_DIAG_*                                                                  */
  int64_t B;         /* batch members                                                              */
  int64_t N;         /* matrix size (rows == cols)                                                 */
  int64_t R;         /* LOWRANK: root rank R (C is [B,N,R]);  KRON: n1;  DENSE: unused             */
  int64_t n2;        /* KRON: n2 (N == n1*n2); others unused                                       */
  const float* A0;   /* LOWRANK: C [B,N,R];  DENSE: K [B,N,N];  KRON: K1 [B,n1,n1]                 */
  const float* A1;   /* KRON: K2 [B,n2,n2]; others NULL                                            */
  const float* d;    /* diagonal, layout per diag_mode (NULL if LO_DIAG_NONE)                      */
  int32_t nterms;    /* SUM: number of terms (2 .. LO_MAX_TERMS); others 0                         */
  int32_t reserved;
  const struct lo_op_desc* terms; /* SUM: HOST array of nterms descriptors of kind LOWRANK / DENSE / KRON with
                      * diag_mode LO_DIAG_NONE and the same B, N (summed left to right, like the reference's
                      * Python sum()); the SUM's own (diag_mode, d) is the one diagonal of the tree */
} lo_op_desc;

/* Callbacks for LO_OP_CALLBACK and for a user preconditioner closure.
 * v, y: device pointers [B, N, c] fp32 contiguous.  Must enqueue on `stream`.  Return 0 on success. */
typedef int (*lo_matvec_cb)(void* user, const float* v, float* y, int64_t B, int64_t N, int64_t c, void* stream);

/* Pivoted-Cholesky/QR Woodbury preconditioner  P = L L^T + D  (added_diag_linear_operator.py:95-184)
 * in the form the reference caches it: z = r/d - Q (Q^T r)   (non-constant diag, :140)
 *                                     z = (r - Q Q^T r)/sigma (constant diag, :137-139)          */
typedef struct lo_precond_desc {
  int32_t k;             /* rank of Q (<= 256)                                                    */
  int32_t ldq;           /* row stride of Q in floats: k, or the zero-padded stride (4 * pow2) that   *
                          * lo_precond_build_f32 emits (no staging copy in that case)               */
  int32_t constant_diag; /* 1: `dinv` holds 1/sigma per member [B]; 0: `dinv` holds 1/d [B,N]      */
  int32_t reserved;
  const float* Q;        /* [B, N, ldq]  (_q_cache; for constant diag it carries the 1/sqrt(sigma)  *
                          * factor so that z = r*dinv - Q (Q^T r) in both cases)                    */
  const float* dinv;     /* reciprocal noise                                                       */
  /* Optional ROOT FORM of the same preconditioner (lo_precond_root_form_f32), valid when the operator is
   * LO_OP_LOWRANK_DIAG and the factor L is the pivoted Cholesky factor of ITS root C: every column of L lies in the
   * column space of C (L = C M), hence  P^-1 r = (r - C F (C^T (r o dinv))) o dinv  with the R x R matrix
   * F = M (I + M^T E M)^-1 M^T, E = C^T D^-1 C.  The operator-resident CG kernels then need no second tall matrix and
   * one group all-reduce per iteration.  NULL = not available (Q is used).                                        */
  const float* F;        /* [B, rf_ld, rf_ld], zero padded, symmetric                               */
  const float* EF;       /* [B, rf_ld, rf_ld] = E F                                                 */
  const float* E;        /* [B, rf_ld, rf_ld] = C^T D^-1 C                                          */
  int32_t rf_ld;         /* row stride of F / EF / E = padded root rank (8, 16 or 32); 0 = absent   */
  int32_t generation;    /* (was reserved2) optional: a number the host changes whenever it rebuilds this cache; part of
                          * the key under which lo_cg_solve_f32 remembers that a solve missed the stop rule in its
                          * result-only pass, so that an ADDRESS the allocator recycles does not inherit the memo.  0 = none.
                          * Results are bit-reproducible per cache state (form of the cache x memo), not across them.      */
  /* Optional KRONECKER ROOT FORM of the same preconditioner (lo_precond_kron_root_f32), valid when the operator is
   * LO_OP_KRON_DIAG with a constant diagonal and L is the pivoted Cholesky factor of K1 (x) K2 with k <= 16 pivots.
   * Row pi of K1 (x) K2 is the Kronecker product of row pi / n2 of K1 and row pi % n2 of K2, and every column of L
   * is a combination of the pivot rows (L = KP M, KP[(i1, i2), m] = kron_a[i1][m] * kron_b[i2][m]), hence
   *   P^-1 r = (r - KP F KP^T (r o dinv)) o dinv,   F = (KP[pivots, :] + KP^T D^-1 KP)^-1  (k x k).
   * The single-column CG iteration of large N (csrc/lo_precond_fused.hip) then forms the rows of KP on the fly from
   * 16 (n1 + n2) floats per member instead of streaming the 16 N floats of Q.  NULL = not available (Q is used).  */
  const float* kron_a;   /* [B, n1, 16]: kron_a[i1][m] = K1[pi_m / n2, i1], columns >= k zero       */
  const float* kron_b;   /* [B, n2, 16]: kron_b[i2][m] = K2[pi_m % n2, i2]                          */
  const float* kron_F;   /* [B, 16, 16], symmetric, zero padded                                     */
  /* Optional R-SPACE FORM next to the root form (lo_precond_root_form_rs_f32; round 5, ABI 11): the fp64 matrices
   * E = C^T D^-1 C | F E | E F E | G2 = C^T C | F | E F, [B, 6, rf_ld, rf_ld], zero padded.  With A P^-1 = I + C (I - F - E F) C^T D^-1
   * every vector of linear_cg (linear_cg.py:245-332) is a combination of the right-hand side and the columns of C, so
   * the iterations of a single-column, result-only solve run on R + 1 coordinates (csrc/lo_rspace.hip): the rows of C
   * are touched twice per solve and a member costs one group all-reduce.  The Gram matrices have to be fp64-accurate
   * (tests/proto/proto_rspace.py); D^-1 is `dinv` as stored (FULL) / 1.0 / (double)sigma (CONST).  NULL = not available. */
  const double* RS;
  /* Optional DIAGONAL FORM of the R-space iteration (lo_precond_eigform_f32; round 5, ABI 13): fp64 [B, 6, rf_ld, rf_ld] =
   * TinT | E^+ | TuT | G2 = C^T C | Tin | {row 0: lam, row 1: status, sweeps, sweeps, rank}.  In the basis that diagonalises the
   * preconditioned member on span(C) (two Jacobi eigendecompositions per member, csrc/lo_eigform.hip) linear_cg is the
   * CG of a diagonal matrix: one reduction of three values per iteration and no R x R product on the dependent chain
   * (k_cg_rspace<.., true>).  Worth building when the cache serves more than one solve (it costs two R x R
   * eigendecompositions per member); same iterates as the RS form to fp64 rounding (tests/proto/proto_eigform.py).
   * NULL = not available (the RS form is used).  Needs RS. */
  const double* RSD;
} lo_precond_desc;

/* Batch-sharded solves (one process per GPU, SURVEY.md section 8(e) "option A"): the reference's stopping rule is the
 * mean residual over the WHOLE batch (linear_cg.py:302-308).  When the batch is split over ranks the engine hands the
 * local statistics to this hook at every stopping-rule evaluation; the hook all-reduces (SUM) the three values in
 * place over the ranks (8-byte-class collective: torch.distributed / RCCL on the host side) and every rank takes the
 * same decision.  vals = {sum of residual norms, number of (member, column) pairs, abort request (NaN seen)}.
 * All ranks call it the same number of times (once per iteration from the first possible stop on).  Returns 0 on success. */
typedef int (*lo_stop_reduce_cb)(void* user, double* vals /* [3], in/out */);

/* linear_cg arguments that are scalars in the reference signature (linear_cg.py:98-109). */
typedef struct lo_cg_params {
  int64_t c;                /* number of right-hand-side columns                                  */
  int32_t n_tridiag;        /* tridiagonalise the first n_tridiag columns (0 = none)              */
  int32_t max_iter;         /* n_iter (already min'ed with N if terminate_cg_by_size, :170)        */
  int32_t max_tridiag_iter; /* n_tridiag_iter = min(max_tridiag_iter, N) (:171)                    */
  int32_t floor_max_iter;   /* the caller's ORIGINAL max_iter for the stop-rule floors min(10, max_iter-1) and
                             * min(n_tridiag_iter, max_iter-1) (:303-305), which the reference evaluates with the
                             * unclipped value; 0 = same as max_iter                                */
  float tolerance;          /* settings.cg_tolerance (:150-151)                                    */
  float eps;                /* 1e-10 (:104)                                                        */
  float stop_updating_after;/* 1e-10 (:105)                                                        */
  float pad;
  lo_stop_reduce_cb stop_reduce; /* NULL: the stopping rule sees this call's batch only (single process, or
                                  * "option B": per-shard rule); else the batch-global rule over all ranks */
  void* stop_reduce_user;
} lo_cg_params;

/* What the reference reports through its warning text / exception (host-readable after the call). */
typedef struct lo_cg_info {
  int32_t iterations;        /* loop bodies executed (k+1 of linear_cg.py:343)                     */
  int32_t matvecs;           /* operator applications actually executed                            */
  int32_t tolerance_reached; /* linear_cg.py:307                                                   */
  int32_t nan_detected;      /* linear_cg.py:199-200                                               */
  int32_t skipped;           /* all columns converged before the first iteration (:207-208)        */
  int32_t last_tridiag_iter; /* t_mat is valid on [0..last_tridiag_iter]^2 (:353)                  */
  float mean_residual;       /* residual_norm.mean() at exit (:343)                                */
  float reserved;
} lo_cg_info;

/* ---- library ------------------------------------------------------------------------------- */
/* ABI version (bumped on any signature change) and the gfx arch string the code objects target. */
int lo_abi_version(void);
const char* lo_target_arch(void);

/* ---- structured matvecs: LinearOperator._matmul for the hot-path operator classes ----------- */
/* y[B,N,c] = A v.  Replaces AddedDiagLinearOperator._matmul (added_diag_linear_operator.py:72-76)
 * over RootLinearOperator._matmul (root_linear_operator.py:68-72), DenseLinearOperator._matmul
 * (dense_linear_operator.py:60-64), KroneckerProductLinearOperator._matmul
 * (kronecker_product_linear_operator.py:272-284), Diag/ConstantDiag (diag_linear_operator.py:203-230). */
size_t lo_matvec_workspace_bytes(const lo_op_desc* op, int64_t c);
int lo_matvec_f32(const lo_op_desc* op, const float* v, float* y, int64_t c, void* ws, size_t ws_bytes, void* stream);

/* ---- linear_cg (linear_operator/utils/linear_cg.py:98-359) ----------------------------------- */
/* Solves A X = rhs for all B members and c columns with the reference's modified preconditioned CG
 * (column normalisation, masked alpha/beta, batch-global stopping rule, CG-coefficient tridiagonals).
 *   op        structured operator, or kind LO_OP_CALLBACK with (matvec, matvec_user)
 *   pre       NULL | Woodbury-QR preconditioner;  precond_cb non-NULL = opaque preconditioner closure
 *   rhs       [B,N,c]      x0  [B,N,c] or NULL (zeros)        x  [B,N,c] out
 *   t_mat     [n_tridiag, B, T, T] out, T = max_tridiag_iter, zero-filled by the call; the caller crops
 *             to [: last_tridiag_iter+1]^2 (linear_cg.py:353-357)
 * SYNCHRONOUS for the caller's purposes: returns with `info` filled and x / t_mat complete on `stream`.  When the solve
 * ends inside the resident launches, the status block arrives through pinned host memory (a ticket the control kernel
 * writes last) and the call returns without a hipStreamSynchronize: work the caller queued on `stream` BEFORE the call
 * is complete, but the stream's tail event may not have retired yet -- order later work by the stream, not by the
 * host.  Otherwise polls the device stop flag between launch chunks.                                       */
size_t lo_cg_workspace_bytes(const lo_op_desc* op, const lo_precond_desc* pre, const lo_cg_params* prm);
int lo_cg_solve_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const lo_precond_desc* pre,
                    lo_matvec_cb precond_cb, void* precond_user, const lo_cg_params* prm, const float* rhs,
                    const float* x0, float* x, float* t_mat, void* ws, size_t ws_bytes, lo_cg_info* info,
                    void* stream);

/* ---- prototype: the gather of SURVEY 8(e) as peer writes (round 6) ------------------------------------------------ */
/* north_star: "independent batch elements shard across the 8 GPUs of one node with an RCCL all-gather over xGMI at the
 * end".  Instead of a collective kernel that has to share the CUs with the spin-waiting resident groups, the x pass of the
 * resident single-column solve (k_cg_rspace3) can store every solution value into up to seven more buffers: the IPC-mapped
 * gather buffers of the peers, bufs[p] [B_total, N] fp32, this rank's members starting at `member_offset`.  The pointers
 * are process-wide and stay installed until replaced (n = 0 removes them); only the diagonal-form resident solve honours
 * them.  Emulated on one GPU with local buffers by tools/mb_peer_gather.py; no reference counterpart.                 */
int lo_peer_gather_set(float* const* bufs, int n, long long member_offset);

/* ---- engine selection of lo_cg_solve_f32 as a PURE function ------------------------------------------------------ */
/* Which kernels lo_cg_solve_f32 will run for these arguments: decided from the shapes, the null-ness of the pointers in
 * `pre`, the parameters and the number of compute units -- no device memory is read and nothing is launched, so the
 * selection matrix is testable on a machine without a GPU (tests/test_host_api.py holds the table).  lo_cg_solve_f32
 * executes exactly this plan; what is left to run time are the fall-backs after an occupancy query refuses a kernel
 * (next engine down) or a group hand-off times out (streaming engine).
 *   cus  compute units to plan for; <= 0: the current device (LO_OC_RESERVE_CUS applied)                             */
#define LO_ENGINE_NONE 0          /* no serial resident columns                                                 */
#define LO_ENGINE_RESIDENT_GEN1 1 /* (k_cg_onchip of rounds 1 - 5: removed in round 6, never reported any more)          */
#define LO_ENGINE_RESIDENT_GEN2 2 /* k_cg_onchip4 (four rows per thread, Q form, columns one after the other)    */
#define LO_ENGINE_RESIDENT_ROOT 3 /* k_cg_onchip5 (root-form preconditioner, one all-reduce per iteration)       */
#define LO_STREAM_PRE_NONE 0       /* unpreconditioned update                                                    */
#define LO_STREAM_PRE_TWO_PASS 1   /* Q^T r, then z = r/d - Q u (two passes over Q)                              */
#define LO_STREAM_PRE_CLOSURE 2    /* opaque preconditioner closure                                              */
#define LO_STREAM_PRE_FUSED_Q 3    /* k_precond_fused: one pass over Q, r / x / p updates fused                  */
#define LO_STREAM_PRE_FUSED_KRON 4 /* k_precond_fused_kron: Kronecker root form, no Q traffic                    */
#define LO_STREAM_PRE_FUSED_COLS 5 /* k_cg_step_cols: up to 32 columns, the whole step behind the product (alpha, r / x,
                                    * Q form of the preconditioner, beta, p, control step) in one launch              */
#define LO_STREAM_NOPRE_FUSED_COLS 6 /* the same launch without a preconditioner (z = r)                              */
typedef struct lo_cg_plan {
  int32_t resident;             /* 1: the iterations up to the first possible stop run in operator-resident launches */
  int32_t resident_iterations;  /* how many (first_stop_iteration + 1), 0 if not resident                            */
  int32_t lockstep_cols;        /* columns [0, lockstep_cols) advance 16 at a time on k_cg_lockstep                  */
  int32_t lockstep_group;       /* workgroups per member of that kernel                                              */
  int32_t serial_engine;        /* LO_ENGINE_*: the kernel of the columns [lockstep_cols, c)                         */
  int32_t serial_group;         /* workgroups per member of that kernel                                              */
  int32_t lean;                 /* 1: result-only first pass (state written by a repeat only if CG has to continue)  */
  int32_t needs_q;              /* 1: `pre` carries the root form only and no resident kernel takes the solve:
                                 * lo_cg_solve_f32 returns LO_ERR_UNSUPPORTED, the caller builds the Q form          */
  int32_t streaming_precond;    /* LO_STREAM_PRE_*: the preconditioner step of iterations beyond the resident ones
                                 * (all iterations when resident == 0)                                               */
  int32_t poll_chunk;           /* streaming iterations enqueued between two reads of the control block              */
  int32_t first_stop_iteration; /* min(10, max_iter-1), raised to min(max_tridiag_iter, max_iter-1) with tridiagonals
                                 * (linear_cg.py:302-308)                                                            */
  int32_t reserved;
  int32_t rspace;               /* round 5 (needs lo_precond_desc.RS): the result-only first pass runs the iterations on
                                 * R + 1 coordinates (csrc/lo_rspace.hip): 2 = the single column inside the resident
                                 * launch of `serial_engine` (one all-reduce per member), 1 = ALL columns in three
                                 * streaming launches (k_rs_part / k_rs_iter / k_rs_apply) -- lockstep_cols /
                                 * serial_engine then name the engines of the repeat with the state                  */
  int32_t reserved2;            /* lo_cg_last_executed: 1 = the rspace == 2 launch ran the diagonal form (RSD)           */
} lo_cg_plan;
int lo_cg_plan_f32(const lo_op_desc* op, const lo_precond_desc* pre, int has_precond_cb, int has_x0,
                   const lo_cg_params* prm, int cus, lo_cg_plan* plan);
/* The plan as the calling thread's last successful lo_cg_solve_f32 EXECUTED it (after run-time fall-backs; `reserved`
 * = streaming iterations enqueued after the resident phase): the GPU tests compare it with lo_cg_plan_f32.          */
int lo_cg_last_executed(lo_cg_plan* plan);

/* ---- fused end-to-end solve: ONE resident launch ------------------------------------------------- */
/* A.solve(rhs) of AddedDiag(LowRankRoot(C), Diag | ConstantDiag) end to end, the operator read from HBM once:
 *   PivotedCholesky.forward   (functions/_pivoted_cholesky.py:14-105; `rank` pivots of C C^T)
 *   -> AddedDiagLinearOperator._init_cache (operators/added_diag_linear_operator.py:144-184; in ROOT FORM, see
 *      lo_precond_desc.F / EF: L = C M is never written to memory)
 *   -> linear_cg              (utils/linear_cg.py:98-359; the iterations of the reference's floor, :302-308)
 * per member inside one kernel (csrc/lo_solve_fused_impl.h).  The two BATCH-GLOBAL decisions of the reference cannot
 * be taken inside (members are in flight at different times); they are checked after the launch and reported in
 * `info->status`:
 *   LO_FUSED_OK          x holds the reference's result (every member took all `rank` pivots on its own error and the
 *                        stopping rule held at the floor) -- also when nan_detected / skipped are set
 *   LO_FUSED_EARLY_STOP  some member's own pivot error reached error_tol (or NaN) before `rank` pivots: the shared
 *                        pivot count (_pivoted_cholesky.py:57) needs the three-launch path
 *   LO_FUSED_CONTINUE    the tolerance was not met at the floor: CG has to continue (three-launch path)
 *   LO_FUSED_TIMEOUT     a group exchange timed out (co-residency lost)
 * In the last three cases x is NOT valid and the caller redoes the solve with lo_pivoted_cholesky_f32 +
 * lo_precond_root_form_f32 / lo_precond_build_f32 + lo_cg_solve_f32.
 *   op       LO_OP_LOWRANK_DIAG, R in {8, 16, 32}, 256 <= N <= 16384, diag FULL or CONST
 *   rank     pivots (settings.max_preconditioner_size, 1..16), error_tol = settings.preconditioner_tolerance
 *   prm      as lo_cg_solve_f32 (n_tridiag == 0, no stop_reduce, c <= 8)
 *   x        [B, N, c] out
 *   F, EF, E [B, R, R] out, dinv [B, N] | [B] out, logdet_p [B] out: the root-form preconditioner for later solves
 *            (any may be NULL); swaps [B, rank] int32 out: position exchanged with position m at pivot m -- the
 *            reference's permutation is the identity with these exchanges applied in order (lo_solve_fused_perm)
 * SYNCHRONOUS (one read-back of the status block).  lo_solve_fused_supported: 1 if the shape is taken.            */
#define LO_FUSED_OK 0
#define LO_FUSED_EARLY_STOP 1
#define LO_FUSED_CONTINUE 2
#define LO_FUSED_TIMEOUT 3
typedef struct lo_fused_info {
  int32_t status;            /* LO_FUSED_*                                                         */
  int32_t iterations;        /* as lo_cg_info                                                      */
  int32_t matvecs;
  int32_t tolerance_reached;
  int32_t nan_detected;
  int32_t skipped;
  int32_t rank;              /* pivots taken (= rank when status is LO_FUSED_OK)                   */
  float mean_residual;
} lo_fused_info;
int lo_solve_fused_supported(const lo_op_desc* op, int32_t rank, const lo_cg_params* prm);
size_t lo_solve_fused_workspace_bytes(const lo_op_desc* op, int32_t rank, const lo_cg_params* prm);
int lo_solve_fused_f32(const lo_op_desc* op, int32_t rank, float error_tol, const lo_cg_params* prm, const float* rhs,
                       float* x, float* F, float* EF, float* E, float* dinv, float* logdet_p, int32_t* swaps,
                       void* ws, size_t ws_bytes, lo_fused_info* info, void* stream);
int lo_solve_fused_perm(const int32_t* swaps, int64_t B, int64_t N, int32_t rank, int64_t* perm, void* stream);

/* Development / test switch: 0 forces the streaming (multi-kernel) engine, 1 (default) allows the operator-resident
 * fast path (csrc/lo_cg_onchip.hip) for low-rank + Woodbury-preconditioned single-column solves.  Same results
 * up to summation order. */
int lo_cg_set_onchip(int enable);

/* ---- the gate of the resident kernels (round 5, ABI 12) ------------------------------------------ */
/* The resident kernels need every workgroup of a group co-resident.  When a group exchange times out (another
 * kernel -- RCCL's, another process' -- holds part of the CUs) the call is redone on the streaming engines and the
 * next `backoff` entry-point calls (lo_cg_solve_f32 / lo_pivoted_cholesky_f32) skip the resident kernels; then they
 * are tried again.  Timeouts right after a re-arm double the cool-down (16 .. 4096 calls), a clean resident solve
 * resets it.  (Rounds 2 - 4 latched the resident kernels off for the life of the process.)
 * lo_resident_inject_timeouts(n): the next n resident CG launches are treated as timed out (tests, bench --inject). */
typedef struct lo_resident_status {
  int32_t user_disabled;  /* lo_cg_set_onchip(0)                                                   */
  int32_t timeouts;       /* hand-off timeouts since the library was loaded                        */
  int32_t cooldown;       /* entry-point calls still to be served by the streaming engines         */
  int32_t backoff;        /* length of the next cool-down                                          */
  int32_t rearms;         /* cool-downs that ended (resident kernels tried again)                  */
  int32_t fused_timeouts; /* of `timeouts`: inside lo_solve_fused_f32                              */
} lo_resident_status;
int lo_resident_status_get(lo_resident_status* out);
int lo_resident_inject_timeouts(int32_t n);

/* ---- PivotedCholesky.forward (linear_operator/functions/_pivoted_cholesky.py:14-105) ---------- */
/* Greedy partial pivoted Cholesky of the NON-diagonal part of `op` (op->d is ignored, as
 * added_diag_linear_operator.py:125 calls self._linear_op.pivoted_cholesky).
 *   L_rows  [B, max_rank, N] out (row m = column m of L; the host returns L_rows[:, :m].mT.contiguous())
 *   perm    [B, N] int64 out (full permutation; first m entries are the pivots)
 *   rank_out host int: number of pivots m taken (shared by the whole batch, :57)
 * SYNCHRONOUS (reads m back).                                                                      */
size_t lo_pivoted_cholesky_workspace_bytes(const lo_op_desc* op, int32_t max_rank);
int lo_pivoted_cholesky_f32(const lo_op_desc* op, int32_t max_rank, float error_tol, float* L_rows, int64_t* perm,
                            int32_t* rank_out, void* ws, size_t ws_bytes, void* stream);
/* The same for an operator that does not lower to a descriptor: the generic row fetch of the reference
 * (`matrix[..., pi_m, :]`, _pivoted_cholesky.py:81 -> LinearOperator.__getitem__ tensor-index branch,
 * operators/_linear_operator.py:2882-2902) stays a callback, everything else runs in the same kernels.
 *   diag    [B, N] device: matrix._diagonal() (:39)
 *   row_cb  fetches, for the pivots piv [B] (int64, DEVICE pointer), the rows K[b, piv[b], :] into rows [B, N];
 *           must enqueue on `stream` and must not synchronise; called once per pivot.  Returns 0 on success. */
typedef int (*lo_rowfetch_cb)(void* user, const int64_t* piv, float* rows, int64_t B, int64_t N, void* stream);
size_t lo_pivoted_cholesky_cb_workspace_bytes(int64_t B, int64_t N, int32_t max_rank);
int lo_pivoted_cholesky_cb_f32(int64_t B, int64_t N, const float* diag, lo_rowfetch_cb row_cb, void* row_user,
                               int32_t max_rank, float error_tol, float* L_rows, int64_t* perm, int32_t* rank_out,
                               void* ws, size_t ws_bytes, void* stream);

/* The same in float64 (round 4): PivotedCholesky.forward is dtype-generic in the reference.  Same streaming engine and
 * operation order; the descriptor's A0 / A1 are read as `const double*` (kinds LOWRANK / DENSE / KRON / SUM of those; d is
 * ignored), L_rows [B, max_rank, N] double.  No operator-resident fast path.                                          */
typedef int (*lo_rowfetch_cb_f64)(void* user, const int64_t* piv, double* rows, int64_t B, int64_t N, void* stream);
size_t lo_pivoted_cholesky_f64_workspace_bytes(const lo_op_desc* op, int32_t max_rank);
int lo_pivoted_cholesky_f64(const lo_op_desc* op, int32_t max_rank, double error_tol, double* L_rows, int64_t* perm,
                            int32_t* rank_out, void* ws, size_t ws_bytes, void* stream);
size_t lo_pivoted_cholesky_cb_f64_workspace_bytes(int64_t B, int64_t N, int32_t max_rank);
int lo_pivoted_cholesky_cb_f64(int64_t B, int64_t N, const double* diag, lo_rowfetch_cb_f64 row_cb, void* row_user,
                               int32_t max_rank, double error_tol, double* L_rows, int64_t* perm, int32_t* rank_out,
                               void* ws, size_t ws_bytes, void* stream);

/* ---- AddedDiagLinearOperator._init_cache* (added_diag_linear_operator.py:144-184) ------------- */
/* From L [B,N,k] and the diagonal builds Q [B,N,k] (the reference's _q_cache, up to the sign/rotation
 * freedom of a thin QR, which Q Q^T is invariant to), dinv and logdet_p [B].
 * Gram matrix + Cholesky + triangular solve in fp64, rounded once to fp32.            */
size_t lo_precond_build_workspace_bytes(int64_t B, int64_t N, int32_t k);
int lo_precond_build_f32(const float* L, const float* d, int32_t diag_mode, int64_t B, int64_t N, int32_t k, float* Q,
                         float* dinv, float* logdet_p, void* ws, size_t ws_bytes, void* stream);
/* Same with an explicit layout of L: element (member b, row i, column a) = L[b*ld_member + i*ld_row + a*ld_col].
 * (N*k, k, 1) is the reference's [B,N,k]; (max_rank*N, 1, N) consumes the L_rows that
 * lo_pivoted_cholesky_f32 writes without the transposed copy of _pivoted_cholesky.py:105.          */
int lo_precond_build_strided_f32(const float* L, int64_t ld_member, int64_t ld_row, int64_t ld_col, const float* d,
                                 int32_t diag_mode, int64_t B, int64_t N, int32_t k, float* Q, float* dinv,
                                 float* logdet_p, void* ws, size_t ws_bytes, void* stream);
/* Root form of the pivoted-Cholesky preconditioner of a low-rank operator (see lo_precond_desc): from the root
 * C [B, N, R] (R <= 32), the diagonal, the factor L (strided like lo_precond_build_strided_f32, k columns = pivots
 * taken) and the permutation of lo_pivoted_cholesky_f32 (first k entries = pivots) computes, in fp64,
 *   E = C^T D^-1 C,  M (L = C M: the recurrence of _pivoted_cholesky.py:77-92 on the pivot rows),
 *   F = M (I + M^T E M)^-1 M^T,  EF,  logdet P = logdet(I + M^T E M) + sum log d  (== the value of the Q form)
 * and writes F, EF, E as fp32 [B, rf_ld, rf_ld] (rf_ld = 8, 16 or 32 >= R), dinv ([B, N] FULL / [B] CONST), logdet_p [B]. */
size_t lo_precond_root_form_workspace_bytes(int64_t B, int64_t N, int32_t R);
int lo_precond_root_form_f32(const float* C, int32_t R, const float* d, int32_t diag_mode, const float* L,
                             int64_t ld_member, int64_t ld_row, int64_t ld_col, const int64_t* perm, int64_t B,
                             int64_t N, int32_t k, int32_t rf_ld, float* F, float* EF, float* E, float* dinv,
                             float* logdet_p, void* ws, size_t ws_bytes, void* stream);
/* The same with the R-SPACE FORM (lo_precond_desc.RS) as a fifth output: RS fp64 [B, 6, rf_ld, rf_ld] = E | F E | E F E | C^T C | F | E F.
 * E and C^T C are accumulated on the fp64 matrix cores from exact fp32 x fp32 products (the fp32 F / EF / E written next
 * to them are the roundings of the same fp64 matrices).  Needs R % 4 == 0 and a 16-byte aligned C: LO_ERR_UNSUPPORTED
 * otherwise (the caller uses lo_precond_root_form_f32).  Workspace: lo_precond_root_form_rs_workspace_bytes.          */
size_t lo_precond_root_form_rs_workspace_bytes(int64_t B, int64_t N, int32_t R);
int lo_precond_root_form_rs_f32(const float* C, int32_t R, const float* d, int32_t diag_mode, const float* L,
                                int64_t ld_member, int64_t ld_row, int64_t ld_col, const int64_t* perm, int64_t B,
                                int64_t N, int32_t k, int32_t rf_ld, float* F, float* EF, float* E, float* dinv,
                                float* logdet_p, double* RS, void* ws, size_t ws_bytes, void* stream);
/* The DIAGONAL FORM (lo_precond_desc.RSD) from the R-space form RS [B, 6, rf_ld, rf_ld] of a root of rank R (even,
 * <= rf_ld <= 32): RSD [B, 6, rf_ld, rf_ld].  One workgroup per member, fp64, two cyclic Jacobi eigendecompositions in LDS.
 * RSD[b][5][1][0] = 1.0 when the form is usable; -1.0 (an eigenvalue of the preconditioned member is not positive) or -2.0
 * ((s_max / s_min) (1 + s_max^2) of E's kept directions exceeds 1e9: the change of basis V S^-1 would cost more than 1e-7
 * of a solution that is a difference of large terms) otherwise -- the caller keeps the RS form, which stays in the
 * coordinates of C.  Replaces nothing in the reference: it is a cached re-expression of
 * added_diag_linear_operator.py:119-137's closure for linear_cg.py:245-332. */
int lo_precond_eigform_f32(const double* RS, int64_t B, int32_t R, int32_t rf_ld, double* RSD, void* stream);
/* Kronecker root form of the pivoted-Cholesky preconditioner (see lo_precond_desc.kron_*): op = LO_OP_KRON_DIAG with
 * LO_DIAG_CONST, L / perm as lo_precond_root_form_f32 (k <= 16 pivots).  Gathers the pivot rows of the two factors
 * (kron_a [B, n1, 16], kron_b [B, n2, 16]), forms E = KP^T KP / sigma as the Hadamard product of the two small Gram
 * matrices and runs the fp64 algebra of the root form: F [B, 16, 16].  kappa [B]: the factor by which the fp32 rounding
 * of KP^T (r / d) is amplified in P^-1 r, sqrt(sum_m (F E F)_mm E_mm) -- a few units for well separated pivots, large
 * when the pivot rows are nearly dependent (the caller then keeps the Q form).                                     */
size_t lo_precond_kron_root_workspace_bytes(int64_t B);
int lo_precond_kron_root_f32(const lo_op_desc* op, const float* L, int64_t ld_member, int64_t ld_row, int64_t ld_col,
                             const int64_t* perm, int32_t k, float* kron_a, float* kron_b, float* kron_F,
                             float* kappa, void* ws, size_t ws_bytes, void* stream);
/* z = P^{-1} r  (precondition_closure, added_diag_linear_operator.py:135-140) */
size_t lo_precond_apply_workspace_bytes(int64_t B, int64_t N, int32_t k, int64_t c);
int lo_precond_apply_f32(const lo_precond_desc* pre, const float* r, float* z, int64_t B, int64_t N, int64_t c, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- lanczos_tridiag (linear_operator/utils/lanczos.py:9-164) --------------------------------- */
/*   init_vecs [B,N,P];  q_mat [max_iter, B, N, P] out;  t_mat [max_iter, max_iter, B, P] out (reference
 *   storage order, :69-77; the host permutes/crops as :151-161);  iters_out = k+1 (:151). SYNCHRONOUS. */
size_t lo_lanczos_workspace_bytes(const lo_op_desc* op, int64_t P, int32_t max_iter);
int lo_lanczos_tridiag_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const float* init_vecs,
                           int64_t P, int32_t max_iter, float tol, float* q_mat, float* t_mat, int32_t* iters_out,
                           void* ws, size_t ws_bytes, void* stream);
/* q_mat [k, B, N, P] (first k of the stored vectors, working order) -> [P, B, N, k], the layout lanczos.py:154
 * returns (permute(-1, batch, -2, 0).contiguous()).  Since round 4 the host mirror returns the permuted VIEW and the
 * consumers on the path read the native layout (lo_root_from_lanczos_native_f32): this copy is made on request only.  LO_ERR_UNSUPPORTED when k * 32 * (P + 1) floats exceed 64 KiB
 * of LDS (the host then permutes with torch).                                                       */
int lo_lanczos_permute_f32(const float* q_in, int32_t k, int64_t B, int64_t N, int64_t P, float* q_out, void* stream);
/* Epilogue of RootDecomposition.forward (functions/_root_decomposition.py:73-85) for PB = probes x batch members:
 * qv = q evecs, root = qv o sqrt(evals), inverse = qv / sqrt(evals); q, qv, root, inverse [PB, N, k], evecs
 * [PB, k, k], evals [PB, k]; any of the three outputs may be NULL.  k <= 32, else LO_ERR_UNSUPPORTED.      */
int lo_root_from_lanczos_f32(const float* q, const float* evecs, const float* evals, int64_t PB, int64_t N, int32_t k,
                             float* qv, float* root, float* inverse, void* stream);

/* The same epilogue on the basis in the layout lo_lanczos_tridiag_f32 WRITES it: q_native [k, B, N, P] (the first k stored
 * vectors) with evecs [P, B, k, k], evals [P, B, k]; outputs [P, B, N, k].  The [P, B, N, k] copy of lanczos.py:154 is
 * then never made.  LO_ERR_UNSUPPORTED: k > 32 or P (k k + 1 + k + (256 / P) k) floats beyond 64 KiB of LDS.           */
int lo_root_from_lanczos_native_f32(const float* q_native, const float* evecs, const float* evals, int64_t B, int64_t N,
                                    int64_t P, int32_t k, float* qv, float* root, float* inverse, void* stream);

/* ---- host glue of InvQuadLogdet as kernels (csrc/lo_probes.hip; round 4, ABI 10) ----------------------------- */
/* Probe vectors of InvQuadLogdet.forward (functions/_inv_quad_logdet.py:91-110) for the preconditioner P = L L^T + D of an
 * AddedDiagLinearOperator: z = L e1 + sqrt(d) o e2 (a draw from N(0, P): zero_mean_mvn_samples of the PsdSum,
 * sum_linear_operator.py:88-91), the column norms (:108), z / norm (:109) written into the FIRST P columns of the
 * right-hand-side block rhs_out [B, N, P + q], the q inv_quad columns copied behind them (the torch.cat of :131).
 *   L          element (b, n, j) at L[b * l_sb + n * l_sn + j * l_sk]  (any layout / broadcast batch), k <= 32 columns
 *   d          [B, N] (LO_DIAG_FULL) or [B] (LO_DIAG_CONST);  e1 [B, k, P], e2 [B, N, P] standard normal draws
 *   norms      [B, P] out;  ws: lo_probe_vectors_workspace_bytes.  P, q <= 64.  Asynchronous on `stream`.          */
size_t lo_probe_vectors_workspace_bytes(int64_t B, int64_t N, int64_t P);
int lo_probe_vectors_f32(const float* L, int64_t l_sb, int64_t l_sn, int64_t l_sk, int32_t k, const float* d,
                         int32_t diag_mode, const float* e1, const float* e2, const float* inv_quad_rhs, int64_t q,
                         int64_t B, int64_t N, int64_t P, float* rhs_out, float* norms, void* ws, size_t ws_bytes,
                         void* stream);
/* Element-wise part of InvQuadLogdet.backward (:183-213) in one pass.  solves [B, N, P + q]; pp [B, N, ldp] = the
 * preconditioner applied to the NORMALISED probes (first P columns read); norms [B, P]; g_ld [B] logdet grad;
 * g_iq [B, q] inv_quad grad (or NULL when q = 0); coef = 1 / P.  Outputs: left, right [B, N, P + q] (the factors of
 * linear_op._bilinear_derivative: [probe part | inv_quad part]) and pre_left, pre_right [B, N, P] (the factors of the
 * preconditioner's bilinear derivative, :211-213).                                                                  */
int lo_iql_backward_factors_f32(const float* solves, const float* pp, int64_t ldp, const float* norms, const float* g_ld,
                                const float* g_iq, float coef, int64_t B, int64_t N, int64_t P, int64_t q, float* left,
                                float* right, float* pre_left, float* pre_right, void* stream);

/* ---- lanczos_tridiag_to_diag + StochasticLQ.to_dense (lanczos.py:167-189, stochastic_lq.py:45-82) */
/* t_mat [M, T, T] (M = P*B tridiagonals, only the three diagonals are read) ->
 *   evals [M, T], evecs [M, T, T] (column j = eigenvector j; negative eigenvalues -> 1 and their
 *   eigenvector columns zeroed, lanczos.py:185-187).  evecs may be NULL.
 * If logdet != NULL (size B): logdet[b] = (n / P) * sum_p sum_i evecs[p,b,0,i]^2 log(evals[p,b,i]).
 * One thread per tridiagonal, implicit-shift QL in fp64; T <= 32 (the reference itself switches algorithm
 * at T >= 32, lanczos.py:179).                                                                      */
size_t lo_tridiag_eigh_slq_workspace_bytes(int64_t P, int64_t B);
int lo_tridiag_eigh_slq_f32(const float* t_mat, int64_t P, int64_t B, int32_t T, int64_t n, float* evals, float* evecs,
                            float* logdet, void* ws, size_t ws_bytes, void* stream);

/* ---- backward passes: `_bilinear_derivative` contractions -------------------------------- */
/* U = left_vecs [B,N,D], V = right_vecs [B,N,D] (fp32, contiguous); derivative of sum_d u_d^T K v_d w.r.t. the
 * tensor representing K.
 *   dense: out [B,N,N] = U V^T                               (dense_linear_operator.py:69-71)
 *   diag : out [B,N] = sum_d U o V; constant != 0: out [B] = sum_{n,d} U o V (ws: B*N floats)
 *                                                            (diag_linear_operator.py:37-45, :337-344)
 *   root : out [B,N,R] = U (V^T C) + V (U^T C) for K = C C^T, C [B,N,R]
 *          (autograd of root._matmul(root._t_matmul(v)), _linear_operator.py:336-393, root_linear_operator.py:68-72) */
int lo_bilinear_dense_f32(const float* U, const float* V, int64_t B, int64_t N, int64_t D, float* out, void* stream);
int lo_bilinear_diag_f32(const float* U, const float* V, int64_t B, int64_t N, int64_t D, int32_t constant, float* out,
                         void* ws, size_t ws_bytes, void* stream);
size_t lo_bilinear_root_workspace_bytes(int64_t B, int64_t N, int64_t R, int64_t D);
/* rowdot (optional, [B,N]): sum_d U o V of the same rows, i.e. the Diag derivative of an AddedDiag(Root, Diag)
 * operator, produced in the same pass (NULL: not wanted).                                                          */
int lo_bilinear_root_f32(const float* C, const float* U, const float* V, int64_t B, int64_t N, int64_t R, int64_t D,
                         float* out, float* rowdot, void* ws, size_t ws_bytes, void* stream);

/*   kron : dK1 [B,n1,n1] = sum_d U_d K2 V_d^T, dK2 [B,n2,n2] = sum_d U_d^T K1 V_d with U_d, V_d the [n1,n2] views of
 *          the columns (autograd of the Kronecker matvec, kronecker_product_linear_operator.py:34-45)            */
size_t lo_bilinear_kron_workspace_bytes(int64_t B, int64_t n1, int64_t n2, int64_t D);
int lo_bilinear_kron_f32(const float* K1, const float* K2, const float* U, const float* V, int64_t B, int64_t n1,
                         int64_t n2, int64_t D, float* dK1, float* dK2, void* ws, size_t ws_bytes, void* stream);
/* out [B,N,R] += U [B,N,D] T [B,D,R] in one pass over `out` (ABI 15): the N-sized product of the pull-back through the
 * pivoted Cholesky of a root, bar R = G2 (L11^-1 Rm) -- what PivotedCholesky.backward obtains by autograd
 * (functions/_pivoted_cholesky.py:107-147) -- accumulated onto the gradient the operator's bilinear derivative left in
 * `out`.  R in {8, 16, 32}, D <= 32; LO_ERR_UNSUPPORTED otherwise (the caller takes the library product).            */
int lo_root_apply_add_f32(const float* U, const float* T, int64_t B, int64_t N, int64_t D, int64_t R, float* out,
                          void* stream);

/* ---- shifted MINRES: the reference's second iterative solver -------------------------- */
/* Replaces linear_operator.utils.minres.minres (utils/minres.py:10-207; update block :210-282): solutions of
 * (value * K + shift_q I) x_q = rhs for all shifts at once, optional preconditioner (Woodbury descriptor or closure).
 * It is what contour_integral_quad (utils/contour_integral_quad.py:137-144) and sqrt_inv_matmul call.
 *   rhs [B,N,c]; shifts [n_shifts] or [n_shifts, B] (shifts_per_member); x [n_shifts, B, N, c].
 * Stop rule (:178-183): every 10th iteration, mean over shifts / members / columns of ||update|| / ||solution|| <
 * tolerance; the loop body runs at most max_iter + 2 times (:134; max_iter already min'ed with N + 1, :60).        */
typedef struct lo_minres_params {
  int64_t c;
  int32_t n_shifts;
  int32_t max_iter;
  int32_t has_value;          /* 0: value = None (:66-68)                                                          */
  int32_t shifts_per_member;  /* 0: shifts [n_shifts]; 1: shifts [n_shifts, B]                                     */
  float value;
  float tolerance;            /* settings.minres_tolerance                                                          */
  float eps;                  /* 1e-25 (:13)                                                                        */
  float pad;
} lo_minres_params;

typedef struct lo_minres_info {
  int32_t iterations;
  int32_t matvecs;
  int32_t converged;
  float conv;                 /* last evaluated mean relative update norm                                           */
} lo_minres_info;

size_t lo_minres_workspace_bytes(const lo_op_desc* op, const lo_precond_desc* pre, const lo_minres_params* prm);
int lo_minres_f32(const lo_op_desc* op, lo_matvec_cb matvec, void* matvec_user, const lo_precond_desc* pre,
                  lo_matvec_cb precond_cb, void* precond_user, const lo_minres_params* prm, const float* rhs,
                  const float* shifts, float* x, void* ws, size_t ws_bytes, lo_minres_info* info, void* stream);

/* ---- linear_cg in fp64 (linear_operator/utils/linear_cg.py:98-359 with float64 operands) ---------------------- */
/* north_star's path is fp32; this entry exists so that the reference's own fp64 recipes run unmodified (every case of
 * its test/utils/test_linear_cg.py:27-160 builds float64 operands).  Plain streaming formulation, same masking,
 * stopping rule and CG-coefficient tridiagonals as lo_cg_solve_f32.
 *   A         [B, N, N] dense fp64 operator (+ diag [B, N] or NULL), or NULL with (matvec, matvec_user): the closure
 *             is called once per product on [B, N, c] device buffers, must enqueue on `stream`
 *   precond_cb  opaque preconditioner closure or NULL (identity)
 *   rhs, x0 (or NULL), x  [B, N, c];  t_mat [n_tridiag, B, T, T] out, T = max_tridiag_iter, zero-filled by the call
 * SYNCHRONOUS (one control read-back per iteration).  No batch-global stop hook over ranks.                        */
typedef int (*lo_matvec_cb_f64)(void* user, const double* v, double* y, int64_t B, int64_t N, int64_t c, void* stream);
typedef struct lo_cg_params_f64 {
  int64_t c;
  int32_t n_tridiag, max_iter, max_tridiag_iter, floor_max_iter; /* as lo_cg_params */
  double tolerance, eps, stop_updating_after;
} lo_cg_params_f64;
typedef struct lo_cg_info_f64 {
  int32_t iterations, matvecs, tolerance_reached, nan_detected, skipped, last_tridiag_iter; /* as lo_cg_info */
  double mean_residual;
} lo_cg_info_f64;
size_t lo_cg_f64_workspace_bytes(int64_t B, int64_t N, const lo_cg_params_f64* prm);
int lo_cg_solve_f64(const double* A, const double* diag, lo_matvec_cb_f64 matvec, void* matvec_user,
                    lo_matvec_cb_f64 precond_cb, void* precond_user, const lo_cg_params_f64* prm, int64_t B, int64_t N,
                    const double* rhs, const double* x0, double* x, double* t_mat, void* ws, size_t ws_bytes,
                    lo_cg_info_f64* info, void* stream);

/* ---- shifted MINRES in fp64 (linear_operator/utils/minres.py:10-282 with float64 operands) ---------------------- */
/* As lo_cg_solve_f64: the reference's own test/utils/test_minres.py:17-80 builds float64 operands.  Same recurrences
 * and stop test as lo_minres_f32 (every 10th iteration: mean ||update|| / ||solution|| < tolerance).
 *   A / diag / matvec / precond_cb  as lo_cg_solve_f64
 *   rhs [B, N, c];  shifts [Q] (or [Q, B] with shifts_per_member);  x [Q, B, N, c] out (scaled back, zero columns 0) */
typedef struct lo_minres_params_f64 {
  int64_t c;
  int32_t n_shifts, max_iter, has_value, shifts_per_member; /* as lo_minres_params */
  double value, tolerance, eps;
} lo_minres_params_f64;
typedef struct lo_minres_info_f64 {
  int32_t iterations, matvecs, converged, pad;
  double conv;
} lo_minres_info_f64;
size_t lo_minres_f64_workspace_bytes(int64_t B, int64_t N, const lo_minres_params_f64* prm);
int lo_minres_f64(const double* A, const double* diag, lo_matvec_cb_f64 matvec, void* matvec_user,
                  lo_matvec_cb_f64 precond_cb, void* precond_user, const lo_minres_params_f64* prm, int64_t B, int64_t N,
                  const double* rhs, const double* shifts, double* x, void* ws, size_t ws_bytes,
                  lo_minres_info_f64* info, void* stream);

/* lanczos_tridiag (utils/lanczos.py:9-164) with float64 operands: the reference is dtype-generic and its
 * root_decomposition / Lanczos callers may hand it doubles.  Streaming formulation (csrc/lo_lanczos_f64.hip).
 *   A [B,N,N] (+ diag [B,N]) or matvec   the operator, as lo_cg_solve_f64
 *   init_vecs [B,N,P]                     start block (normalised per column inside, :81)
 *   q_mat [max_iter,B,N,P], t_mat [max_iter,max_iter,B,P]   working order, as lo_lanczos_tridiag_f32; the first
 *   *iters_out vectors / rows+columns are the result (:151)
 * Reads two integers back per step (the reference's re-orthogonalisation and stopping tests), so it synchronises.  */
size_t lo_lanczos_f64_workspace_bytes(int64_t B, int64_t N, int64_t P, int32_t max_iter);
int lo_lanczos_tridiag_f64(const double* A, const double* diag, lo_matvec_cb_f64 matvec, void* matvec_user,
                           const double* init_vecs, int64_t B, int64_t N, int64_t P, int32_t max_iter, double tol,
                           double* q_mat, double* t_mat, int32_t* iters_out, void* ws, size_t ws_bytes, void* stream);

/* ---- measurement aid (no reference counterpart) ------------------------------------------------ */
/* Opt-in HIP-event timing of every kernel launch of the library, recorded on the launch stream.
 * lo_prof_report writes "name count total_ms" lines into buf (returns the byte count) and resets.
 * bench.py uses it for the live per-kernel duration behind its roofline line.                     */
int lo_prof_enable(int on);
int lo_prof_report(char* buf, size_t buflen);
/* a = b + s c over n floats (n % 4 == 0): the achievable HBM rate of the box (3 x 4 n bytes per launch), reported by
 * bench.py beside the 8 TB/s spec peak; lo_hbm_copy_f32: a = b (2 x 4 n bytes), the figure the MI355X guide quotes as
 * "achievable" (6.29 TB/s).  Grid sized to the arrays, four 16-byte requests in flight per thread, non-temporal.     */
int lo_hbm_triad_f32(float* a, const float* b, const float* c, float s, size_t n, void* stream);
int lo_hbm_copy_f32(float* a, const float* b, size_t n, void* stream);
/* sweep aid for tools/mb_stream.py: mode 0 triad / 1 copy / 2 read-only; unroll 1, 2, 4, 8; nt 0 / 1               */
int lo_hbm_stream_dev(int mode, int unroll, int nt, float* a, const float* b, const float* c, float s, size_t n,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LO_AMD_H */
