"""ctypes binding of the C oracle (oracle/lo_oracle_c.c) -- TEST INFRASTRUCTURE ONLY, never a product path.

The C file restates the reference's hot path with the argument structures of include/lo_amd.h and HOST pointers; this
module declares those structures independently of the package's own binding (a second reading of the header), wraps the
entry points for numpy arrays and builds the library with gcc on first use (`make -C oracle`).  Only tests/,
__graft_entry__ (build / smoke) and bench.py's cpu_baseline leg import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liblo_oracle.so")

LO_OP_LOWRANK_DIAG, LO_OP_DENSE_DIAG, LO_OP_KRON_DIAG, LO_OP_CALLBACK, LO_OP_SUM = 0, 1, 2, 3, 4
LO_DIAG_NONE, LO_DIAG_FULL, LO_DIAG_CONST = 0, 1, 2


class OpDesc(C.Structure):
    pass


OpDesc._fields_ = [("kind", C.c_int32), ("diag_mode", C.c_int32), ("B", C.c_int64), ("N", C.c_int64), ("R", C.c_int64),
                   ("n2", C.c_int64), ("A0", C.c_void_p), ("A1", C.c_void_p), ("d", C.c_void_p), ("nterms", C.c_int32),
                   ("reserved", C.c_int32), ("terms", C.POINTER(OpDesc))]


class PrecondDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("ldq", C.c_int32), ("constant_diag", C.c_int32), ("reserved", C.c_int32),
                ("Q", C.c_void_p), ("dinv", C.c_void_p), ("F", C.c_void_p), ("EF", C.c_void_p), ("E", C.c_void_p),
                ("rf_ld", C.c_int32), ("reserved2", C.c_int32), ("kron_a", C.c_void_p), ("kron_b", C.c_void_p),
                ("kron_F", C.c_void_p)]


class CgParams(C.Structure):
    _fields_ = [("c", C.c_int64), ("n_tridiag", C.c_int32), ("max_iter", C.c_int32), ("max_tridiag_iter", C.c_int32),
                ("floor_max_iter", C.c_int32), ("tolerance", C.c_float), ("eps", C.c_float),
                ("stop_updating_after", C.c_float), ("pad", C.c_float), ("stop_reduce", C.c_void_p),
                ("stop_reduce_user", C.c_void_p)]


class CgInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("matvecs", C.c_int32), ("tolerance_reached", C.c_int32),
                ("nan_detected", C.c_int32), ("skipped", C.c_int32), ("last_tridiag_iter", C.c_int32),
                ("mean_residual", C.c_float), ("reserved", C.c_float)]


MATVEC_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/_build/liblo_oracle.so with gcc (seconds); returns its path."""
    src = os.path.join(HERE, "lo_oracle_c.c")
    stale = (not os.path.exists(LIB)) or os.path.getmtime(LIB) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return LIB


def usable_cpus() -> int:
    """CPUs this process may actually run on: the affinity mask, cut by the cgroup quota if there is one (the container
    of a GPU box can see every core of the host and own a handful)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def load():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # (idle OpenMP threads must not spin on shared cores)
        lib = C.CDLL(build())
        lib.lo_cpu_set_num_threads.restype = C.c_int
        lib.lo_cpu_set_num_threads.argtypes = [C.c_int]
        lib.lo_cpu_set_num_threads(int(os.environ.get("LO_ORACLE_THREADS", usable_cpus())))
        P = C.POINTER
        lib.lo_cpu_num_threads.restype = C.c_int
        lib.lo_cpu_matvec_f32.restype = C.c_int
        lib.lo_cpu_matvec_f32.argtypes = [P(OpDesc), C.c_void_p, C.c_void_p, C.c_int64]
        lib.lo_cpu_pivoted_cholesky_f32.restype = C.c_int
        lib.lo_cpu_pivoted_cholesky_f32.argtypes = [P(OpDesc), C.c_int32, C.c_float, C.c_void_p, C.c_void_p, P(C.c_int32)]
        lib.lo_cpu_precond_build_f32.restype = C.c_int
        lib.lo_cpu_precond_build_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        lib.lo_cpu_precond_apply_f32.restype = C.c_int
        lib.lo_cpu_precond_apply_f32.argtypes = [P(PrecondDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
        lib.lo_cpu_cg_solve_f32.restype = C.c_int
        lib.lo_cpu_cg_solve_f32.argtypes = [P(OpDesc), MATVEC_CB, C.c_void_p, P(PrecondDesc), MATVEC_CB, C.c_void_p,
                                            P(CgParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(CgInfo)]
        lib.lo_cpu_lanczos_tridiag_f32.restype = C.c_int
        lib.lo_cpu_lanczos_tridiag_f32.argtypes = [P(OpDesc), MATVEC_CB, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                                   C.c_void_p, C.c_void_p, P(C.c_int32)]
        lib.lo_cpu_tridiag_eigh_slq_f32.restype = C.c_int
        lib.lo_cpu_tridiag_eigh_slq_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Operator:
    """Host-side lo_op_desc over numpy arrays (kept alive by this object)."""

    def __init__(self, kind, B, N, A0=None, A1=None, d=None, const_diag=False, R=0, n2=0, terms=()):
        self.keep = [_f32(A0) if A0 is not None else None, _f32(A1) if A1 is not None else None,
                     _f32(d) if d is not None else None]
        self.terms = tuple(terms)
        s = OpDesc()
        s.kind, s.B, s.N, s.R, s.n2 = kind, B, N, R, n2
        s.diag_mode = LO_DIAG_NONE if d is None else (LO_DIAG_CONST if const_diag else LO_DIAG_FULL)
        s.A0, s.A1, s.d = (None if a is None else a.ctypes.data for a in self.keep)
        s.nterms = len(self.terms)
        if self.terms:
            self._arr = (OpDesc * len(self.terms))(*[t.s for t in self.terms])
            s.terms = C.cast(self._arr, C.POINTER(OpDesc))
        self.s, self.B, self.N = s, B, N


def lowrank_diag(Cm, d=None, const_diag=False):
    Cm = _f32(Cm)
    B = int(np.prod(Cm.shape[:-2], dtype=np.int64))
    return Operator(LO_OP_LOWRANK_DIAG, B, Cm.shape[-2], A0=Cm, d=d, const_diag=const_diag, R=Cm.shape[-1])


def dense_diag(K, d=None, const_diag=False):
    K = _f32(K)
    B = int(np.prod(K.shape[:-2], dtype=np.int64))
    return Operator(LO_OP_DENSE_DIAG, B, K.shape[-1], A0=K, d=d, const_diag=const_diag)


def kron_diag(K1, K2, d=None, const_diag=False):
    K1, K2 = _f32(K1), _f32(K2)
    B = int(np.prod(K1.shape[:-2], dtype=np.int64))
    n1, n2 = K1.shape[-1], K2.shape[-1]
    return Operator(LO_OP_KRON_DIAG, B, n1 * n2, A0=K1, A1=K2, d=d, const_diag=const_diag, R=n1, n2=n2)


def sum_op(terms, d=None, const_diag=False):
    t0 = terms[0]
    return Operator(LO_OP_SUM, t0.B, t0.N, d=d, const_diag=const_diag, terms=terms)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"liblo_oracle {what} failed: {rc}")


def matvec(op: Operator, v):
    v = _f32(v)
    y = np.empty_like(v)
    _check(load().lo_cpu_matvec_f32(C.byref(op.s), _ptr(v), _ptr(y), v.shape[-1]), "lo_cpu_matvec_f32")
    return y


def pivoted_cholesky(op: Operator, rank, error_tol=1e-3):
    """(L [B, N, m], permutation [B, N] int64): the reference's return values (_pivoted_cholesky.py:105)."""
    rank = int(rank)
    L_rows = np.empty((op.B, rank, op.N), dtype=np.float32)
    perm = np.empty((op.B, op.N), dtype=np.int64)
    m = C.c_int32(0)
    _check(load().lo_cpu_pivoted_cholesky_f32(C.byref(op.s), rank, float(error_tol), _ptr(L_rows), _ptr(perm), C.byref(m)),
           "lo_cpu_pivoted_cholesky_f32")
    return np.ascontiguousarray(np.swapaxes(L_rows[:, : m.value], -1, -2)), perm


class Preconditioner:
    def __init__(self, L, d, const_diag=False):
        L = _f32(L)
        B, N, k = L.shape
        self.Q = np.empty((B, N, k), dtype=np.float32)
        d = _f32(d)
        self.dinv = np.empty_like(d)
        self.logdet = np.empty(B, dtype=np.float32)
        _check(load().lo_cpu_precond_build_f32(_ptr(L), _ptr(d), LO_DIAG_CONST if const_diag else LO_DIAG_FULL, B, N, k,
                                               _ptr(self.Q), _ptr(self.dinv), _ptr(self.logdet)), "lo_cpu_precond_build_f32")
        s = PrecondDesc()
        s.k, s.ldq, s.constant_diag = k, k, int(const_diag)
        s.Q, s.dinv = self.Q.ctypes.data, self.dinv.ctypes.data
        self.s, self.B, self.N = s, B, N

    def apply(self, r):
        r = _f32(r)
        z = np.empty_like(r)
        _check(load().lo_cpu_precond_apply_f32(C.byref(self.s), _ptr(r), _ptr(z), self.B, self.N, r.shape[-1]),
               "lo_cpu_precond_apply_f32")
        return z


def linear_cg(op: Operator, rhs, pre: Preconditioner = None, x0=None, n_tridiag=0, tolerance=1.0, eps=1e-10,
              stop_updating_after=1e-10, max_iter=1000, max_tridiag_iter=20, floor_max_iter=0):
    """(x, t_mat or None, CgInfo): t_mat [n_tridiag, B, T', T'] cropped as linear_cg.py:353."""
    rhs = _f32(rhs)
    x = np.empty_like(rhs)
    prm = CgParams()
    prm.c, prm.n_tridiag, prm.max_iter, prm.max_tridiag_iter = rhs.shape[-1], n_tridiag, max_iter, max_tridiag_iter
    prm.floor_max_iter, prm.tolerance, prm.eps, prm.stop_updating_after = floor_max_iter, tolerance, eps, stop_updating_after
    t_mat = np.empty((n_tridiag, op.B, max_tridiag_iter, max_tridiag_iter), dtype=np.float32) if n_tridiag else None
    info = CgInfo()
    x0 = None if x0 is None else _f32(x0)
    _check(load().lo_cpu_cg_solve_f32(C.byref(op.s), MATVEC_CB(), None, C.byref(pre.s) if pre is not None else None,
                                      MATVEC_CB(), None, C.byref(prm), _ptr(rhs), _ptr(x0), _ptr(x), _ptr(t_mat),
                                      C.byref(info)), "lo_cpu_cg_solve_f32")
    if t_mat is not None:
        m = info.last_tridiag_iter + 1
        t_mat = np.ascontiguousarray(t_mat[:, :, :m, :m])
    return x, t_mat, info


def num_threads() -> int:
    return int(load().lo_cpu_num_threads())


def lanczos_tridiag(op: Operator, init_vecs, max_iter, tol=1e-5):
    """(q_mat [P, B, N, k'], t_mat [P, B, k', k']) as the reference returns them (utils/lanczos.py:151-161; the leading
    dimension is squeezed iff P == 1)."""
    V = _f32(init_vecs)
    B, N, Pn = V.shape
    max_iter = int(max_iter)
    num = min(max_iter, N)
    q = np.empty((num, B, N, Pn), dtype=np.float32)
    t = np.empty((num, num, B, Pn), dtype=np.float32)
    m = C.c_int32(0)
    _check(load().lo_cpu_lanczos_tridiag_f32(C.byref(op.s), MATVEC_CB(), None, _ptr(V), Pn, max_iter, float(tol), _ptr(q),
                                             _ptr(t), C.byref(m)), "lo_cpu_lanczos_tridiag_f32")
    k = m.value
    q_out = np.ascontiguousarray(np.transpose(q[:k], (3, 1, 2, 0)))
    t_out = np.ascontiguousarray(np.transpose(t[:k, :k], (3, 2, 0, 1)))
    if Pn == 1:
        q_out, t_out = q_out[0], t_out[0]
    return q_out, t_out


def tridiag_eigh_slq(t_mat, n, want_spectrum=False):
    """logdet [B] = (n / P) sum_p e1^T log(T_p) e1 over the tridiagonals t_mat [P, B, k, k] (lanczos_tridiag_to_diag +
    StochasticLQ.to_dense, utils/lanczos.py:167-189, stochastic_lq.py:45-82); eigenvalues by implicit QL in double.
    want_spectrum: also (clamped eigenvalues [P, B, k], first eigenvector components [P, B, k])."""
    t = _f32(t_mat)
    Pn, B, k, ld = t.shape
    out = np.empty(B, dtype=np.float32)
    ev = np.empty((Pn, B, k), dtype=np.float64) if want_spectrum else None
    v0 = np.empty((Pn, B, k), dtype=np.float64) if want_spectrum else None
    _check(load().lo_cpu_tridiag_eigh_slq_f32(_ptr(t), Pn, B, k, ld, int(n), _ptr(out), _ptr(ev), _ptr(v0)),
           "lo_cpu_tridiag_eigh_slq_f32")
    return (out, ev, v0) if want_spectrum else out


def inv_quad_logdet(op: Operator, row_op: Operator, d, inv_quad_rhs, probes, tolerance=1.0, max_iter=1000,
                    max_tridiag_iter=20, rank=15, precond_tol=1e-3, const_diag=False):
    """InvQuadLogdet.forward with injected (normalised) probes (functions/_inv_quad_logdet.py:112-153), all in C:
    pivoted Cholesky of `row_op` (the operator without its diagonal) -> preconditioner -> linear_cg with the probe
    tridiagonals -> eigh + SLQ.  Returns (inv_quad [B, c_rhs], logdet [B], solves, t_mat, CgInfo, pivots)."""
    L, piv = pivoted_cholesky(row_op, rank, precond_tol)
    pre = Preconditioner(L, d, const_diag)
    Pn = probes.shape[-1]
    rhs = np.concatenate([_f32(probes), _f32(inv_quad_rhs)], axis=-1)
    x, t_mat, info = linear_cg(op, rhs, pre=pre, n_tridiag=Pn, tolerance=tolerance, max_iter=max_iter,
                               max_tridiag_iter=max_tridiag_iter)
    logdet = tridiag_eigh_slq(t_mat, probes.shape[-2]) + pre.logdet
    inv_quad = np.sum(x[..., Pn:] * _f32(inv_quad_rhs), axis=-2)
    return inv_quad, logdet, x, t_mat, info, piv
