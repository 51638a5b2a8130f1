// lo_matvec.hip -- operator descriptor -> kernel sequence ("op-tree lowering" target), and the public
// lo_matvec_f32 entry point (LinearOperator._matmul of the hot-path classes, lo_amd.h).
#include <algorithm>

#include "lo_internal.h"

namespace lo {

static int padded_rank(int64_t R) {
  int64_t rq = (R + 3) / 4;
  int64_t p = 1;
  while (p < rq) p <<= 1;
  return (int)(4 * p);
}

static bool sum_terms_ok(const lo_op_desc* op) {
  if (op->nterms < 2 || op->nterms > LO_MAX_TERMS || !op->terms) return false;
  for (int i = 0; i < op->nterms; ++i) {
    const lo_op_desc& t = op->terms[i];
    const bool kind_ok = t.kind == LO_OP_LOWRANK_DIAG || t.kind == LO_OP_DENSE_DIAG || t.kind == LO_OP_KRON_DIAG;
    if (!kind_ok || t.diag_mode != LO_DIAG_NONE || t.B != op->B || t.N != op->N) return false;
  }
  return true;
}

size_t matvec_plan_bytes(const lo_op_desc* op, int64_t c, Split sp) {
  Arena ar(nullptr, 0);
  if (op->kind == LO_OP_SUM) {
    if (!sum_terms_ok(op)) return 256;
    size_t total = align_up((size_t)op->B * op->N * c * sizeof(float), 256) + 256;
    for (int i = 0; i < op->nterms; ++i) total += matvec_plan_bytes(&op->terms[i], c, sp);
    return total;
  }
  if (op->kind == LO_OP_LOWRANK_DIAG) {
    const int R4 = padded_rank(op->R);
    if (R4 != op->R) ar.take<float>((size_t)op->B * op->N * R4);
    ar.take<float>((size_t)op->B * sp.S * R4 * c);
  } else if (op->kind == LO_OP_KRON_DIAG) {
    ar.take<float>((size_t)op->B * op->N * c * (kron_mfma_cols_ok((int)op->R, (int)op->n2, c) ? 2 : 1));
  } else if (op->kind == LO_OP_DENSE_DIAG) {
    const int ks = dense_mfma_slices(op->B, op->N, c);
    if (ks > 1) ar.take<float>((size_t)ks * op->B * op->N * c);
  }
  return ar.off + 256;
}

int matvec_plan_init(MatvecPlan* pl, const lo_op_desc* op, lo_matvec_cb cb, void* cb_user, int64_t c, Split sp,
                     Arena* ar, hipStream_t st) {
  pl->op = *op;
  pl->c = c;
  pl->sp = sp;
  pl->cb = cb;
  pl->cb_user = cb_user;
  pl->Apad = nullptr;
  pl->tpart = nullptr;
  pl->mv_resident = false;
  pl->kron_tmp = nullptr;
  pl->dense_part = nullptr;
  pl->lda = pl->R4 = 0;
  pl->S_dot = sp.S;
  pl->nterms = 0;
  pl->sub = nullptr;
  pl->ytmp = nullptr;
  if (op->B < 1 || op->N < 1 || c < 1) return LO_ERR_BADARG;
  if (op->diag_mode != LO_DIAG_NONE && !op->d) return LO_ERR_BADARG;
  switch (op->kind) {
    case LO_OP_LOWRANK_DIAG: {
      if (!op->A0 || op->R < 1) return LO_ERR_BADARG;
      const int R4 = padded_rank(op->R);
      if (R4 > kMaxRank) return LO_ERR_UNSUPPORTED;
      pl->R4 = R4;
      pl->lda = R4;
      if (R4 != op->R) {
        float* p = ar->take<float>((size_t)op->B * op->N * R4);
        if (!ar->ok) return LO_ERR_WORKSPACE;
        int rc = pad_rows(op->A0, (int)op->R, p, R4, op->B * op->N, st);
        if (rc) return rc;
        pl->Apad = p;
      } else {
        pl->Apad = op->A0;
      }
      pl->tpart = ar->take<float>((size_t)op->B * sp.S * R4 * c);
      pl->mv_resident = lowrank_mv_eligible(R4, op->N, c);
      break;
    }
    case LO_OP_DENSE_DIAG: {
      if (!op->A0) return LO_ERR_BADARG;
      pl->S_dot = dense_S_dot(op->B, op->N, c);
      {
        const int ks = dense_mfma_slices(op->B, op->N, c);
        if (ks > 1) pl->dense_part = ar->take<float>((size_t)ks * op->B * op->N * c);
      }
      break;
    }
    case LO_OP_KRON_DIAG: {
      if (!op->A0 || !op->A1 || op->R * op->n2 != op->N) return LO_ERR_BADARG;
      pl->kron_tmp = ar->take<float>((size_t)op->B * op->N * c * (kron_mfma_cols_ok((int)op->R, (int)op->n2, c) ? 2 : 1));
      pl->S_dot = kron_S_dot((int)op->R, (int)op->n2, c, sp.S);
      break;
    }
    case LO_OP_CALLBACK:
      if (!cb) return LO_ERR_BADARG;
      break;
    case LO_OP_SUM: {
      if (!sum_terms_ok(op)) return LO_ERR_BADARG;
      pl->ytmp = ar->take<float>((size_t)op->B * op->N * c);
      pl->nterms = op->nterms;
      pl->sub = new MatvecPlan[op->nterms];
      for (int i = 0; i < op->nterms; ++i) {
        lo_op_desc t = op->terms[i];
        if (i == 0) {  // the tree's one diagonal rides on the first term's epilogue
          t.diag_mode = op->diag_mode;
          t.d = op->d;
        }
        const int rc = matvec_plan_init(&pl->sub[i], &t, nullptr, nullptr, c, sp, ar, st);
        if (rc) {
          matvec_plan_free(pl);
          return rc;
        }
      }
      break;
    }
    default:
      return LO_ERR_BADARG;
  }
  return ar->ok ? LO_OK : LO_ERR_WORKSPACE;
}

void matvec_plan_free(MatvecPlan* pl) {
  if (pl->sub) {
    for (int i = 0; i < pl->nterms; ++i) matvec_plan_free(&pl->sub[i]);
    delete[] pl->sub;
    pl->sub = nullptr;
  }
}

int matvec_run(const MatvecPlan* pl, const float* v, float* y, float* dot_part, const int* stop, hipStream_t st) {
  const lo_op_desc& op = pl->op;
  int rc = LO_OK;
  switch (op.kind) {
    case LO_OP_LOWRANK_DIAG:
      // one pass over C with the rows resident between t = C^T v and y = C t + d o v (lo_lowrank_mv.hip); the fused dot
      // partials of the CG iteration and shapes it does not take run the two streaming passes
      if (pl->mv_resident && !dot_part) {
        rc = lowrank_mv_run(pl->Apad, pl->R4, op.d, op.diag_mode, v, y, op.B, op.N, pl->c, stop, st);
        if (rc != LO_ERR_UNSUPPORTED) return rc;
      }
      rc = skinny_tn(pl->Apad, pl->lda, pl->R4, v, pl->c, pl->tpart, op.B, op.N, pl->sp, stop, st);
      if (rc) return rc;
      return skinny_nn(pl->Apad, pl->lda, pl->R4, pl->tpart, op.d, op.diag_mode, 1.0f, v, pl->c, y, dot_part, op.B,
                       op.N, pl->sp, stop, st);
    case LO_OP_DENSE_DIAG:
      return dense_matvec(op.A0, op.d, op.diag_mode, v, y, dot_part, op.B, op.N, pl->c,
                          dense_rows_per_wg(op.B, op.N), pl->dense_part, stop, st);
    case LO_OP_KRON_DIAG:
      if (kron_mfma_ok((int)op.R, (int)op.n2, pl->c))
        return kron_matvec_mfma(op.A0, op.A1, op.d, op.diag_mode, v, pl->kron_tmp, y, dot_part, op.B, (int)op.R,
                                (int)op.n2, stop, st);
      if (kron_mfma_cols_ok((int)op.R, (int)op.n2, pl->c)) {  // (the diagonal rides on the way back of the columns)
        rc = kron_matvec_mfma_cols(op.A0, op.A1, op.d, op.diag_mode, v, pl->kron_tmp,
                                   pl->kron_tmp + (size_t)op.B * op.N * pl->c, y, op.B, (int)op.R, (int)op.n2, pl->c,
                                   stop, st);
        if (rc) return rc;
      } else {
        rc = kron_matvec(op.A0, op.A1, v, pl->kron_tmp, y, op.B, (int)op.R, (int)op.n2, pl->c, stop, st);
        if (rc) return rc;
        rc = vec_add_diag(op.d, op.diag_mode, v, y, pl->c, op.B, op.N, pl->sp, stop, st);
        if (rc) return rc;
      }
      if (dot_part) rc = vec_dot_part(v, y, pl->c, dot_part, op.B, op.N, pl->sp, stop, st);
      return rc;
    case LO_OP_CALLBACK:
      rc = pl->cb(pl->cb_user, v, y, op.B, op.N, pl->c, (void*)st);
      if (rc) return LO_ERR_LAUNCH;
      if (dot_part) rc = vec_dot_part(v, y, pl->c, dot_part, op.B, op.N, pl->sp, stop, st);
      return rc;
    case LO_OP_SUM:  // sum(op._matmul(rhs) for op in linear_ops), left to right (sum_linear_operator.py:47-51)
      rc = matvec_run(&pl->sub[0], v, y, nullptr, stop, st);
      for (int i = 1; i < pl->nterms && !rc; ++i) {
        rc = matvec_run(&pl->sub[i], v, pl->ytmp, nullptr, stop, st);
        if (!rc) rc = vec_axpy1(y, pl->ytmp, (size_t)op.B * op.N * pl->c, stop, st);
      }
      if (!rc && dot_part) rc = vec_dot_part(v, y, pl->c, dot_part, op.B, op.N, pl->sp, stop, st);
      return rc;
  }
  return LO_ERR_BADARG;
}

bool matvec_can_fuse_pupdate(const MatvecPlan* pl) { return pl->op.kind == LO_OP_LOWRANK_DIAG; }

int matvec_run_pupdate(const MatvecPlan* pl, float* p, const float* z, const float* beta, int first, float* y,
                       float* dot_part, const int* stop, hipStream_t st) {
  const lo_op_desc& op = pl->op;
  if (op.kind != LO_OP_LOWRANK_DIAG) return LO_ERR_BADARG;
  int rc = skinny_tn_pupdate(pl->Apad, pl->lda, pl->R4, p, z, beta, first, pl->c, pl->tpart, op.B, op.N, pl->sp, stop,
                             st);
  if (rc) return rc;
  return skinny_nn(pl->Apad, pl->lda, pl->R4, pl->tpart, op.d, op.diag_mode, 1.0f, p, pl->c, y, dot_part, op.B, op.N,
                   pl->sp, stop, st);
}

}  // namespace lo

using namespace lo;

extern "C" {

int lo_abi_version(void) { return 15; }
const char* lo_target_arch(void) { return "gfx950"; }

size_t lo_matvec_workspace_bytes(const lo_op_desc* op, int64_t c) {
  if (!op) return 0;
  Split sp = choose_split(op->B, op->N, 256);
  return matvec_plan_bytes(op, c, sp);
}

int lo_matvec_f32(const lo_op_desc* op, const float* v, float* y, int64_t c, void* ws, size_t ws_bytes, void* stream) {
  if (!op || !v || !y || op->kind == LO_OP_CALLBACK) return LO_ERR_BADARG;
  hipStream_t st = (hipStream_t)stream;
  Split sp = choose_split(op->B, op->N, 256);
  Arena ar(ws, ws_bytes);
  if (op->kind == LO_OP_LOWRANK_DIAG) resident_tick();  // (an entry point that may run a resident kernel: serves the cool-down)
  if (matvec_plan_bytes(op, c, sp) > 256 && !ws) return LO_ERR_WORKSPACE;
  MatvecPlan pl;
  int rc = matvec_plan_init(&pl, op, nullptr, nullptr, c, sp, &ar, st);
  if (rc) return rc;
  rc = matvec_run(&pl, v, y, nullptr, nullptr, st);
  matvec_plan_free(&pl);
  return rc;
}

}  // extern "C"
