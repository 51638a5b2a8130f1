"""Tensor-level wrappers over the C ABI (include/lo_amd.h): shape checks, batch flattening, workspaces.

Everything here takes contiguous fp32 HIP tensors and launches hand-written gfx950 kernels on the
current stream; there is no other implementation behind these functions.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch

from . import _hip


@dataclass
class OperatorDescriptor:
    """Lowered form of an operator tree (see operators/_lowering.py): what the kernels need."""

    kind: int
    B: int
    N: int
    A0: Optional[torch.Tensor] = None  # C [B,N,R] | K [B,N,N] | K1 [B,n1,n1]
    A1: Optional[torch.Tensor] = None  # K2 [B,n2,n2]
    d: Optional[torch.Tensor] = None  # [B,N] (FULL) or [B] (CONST)
    diag_mode: int = _hip.LO_DIAG_NONE
    R: int = 0
    n2: int = 0
    batch_shape: torch.Size = torch.Size()
    terms: tuple = ()  # LO_OP_SUM: the summed structured terms (descriptors without a diagonal), left to right

    @property
    def device(self):
        for t in (self.A0, self.A1, self.d):
            if t is not None:
                return t.device
        for term in self.terms:
            return term.device
        raise ValueError("empty descriptor")

    def c_struct(self) -> _hip.OpDesc:
        s = _hip.OpDesc()
        s.kind, s.diag_mode, s.B, s.N, s.R, s.n2 = self.kind, self.diag_mode, self.B, self.N, self.R, self.n2
        s.A0 = None if self.A0 is None else self.A0.data_ptr()
        s.A1 = None if self.A1 is None else self.A1.data_ptr()
        s.d = None if self.d is None else self.d.data_ptr()
        s.nterms = len(self.terms)
        if self.terms:  # host array of term descriptors; kept alive by the returned struct
            arr = (_hip.OpDesc * len(self.terms))(*[t.c_struct() for t in self.terms])
            s.terms = C.cast(arr, C.POINTER(_hip.OpDesc))
            s._terms_keepalive = arr
        return s

    def without_diag(self) -> "OperatorDescriptor":
        return OperatorDescriptor(self.kind, self.B, self.N, self.A0, self.A1, None, _hip.LO_DIAG_NONE, self.R,
                                  self.n2, self.batch_shape, self.terms)


def sum_descriptor(terms, d: Optional[torch.Tensor] = None, const_diag: bool = False):
    """SumLinearOperator(A_1, ..., A_n) (+ one diagonal): y = sum_i A_i v + d o v (sum_linear_operator.py:47-51).
    `terms`: 2 .. LO_MAX_TERMS descriptors of kind low-rank / dense / Kronecker WITHOUT a diagonal, same batch and N."""
    terms = tuple(terms)
    if not 2 <= len(terms) <= _hip.LO_MAX_TERMS:
        raise ValueError(f"a lowered sum holds 2 .. {_hip.LO_MAX_TERMS} terms, got {len(terms)}")
    t0 = terms[0]
    for t in terms:
        if t.kind == _hip.LO_OP_SUM or t.diag_mode != _hip.LO_DIAG_NONE or (t.B, t.N) != (t0.B, t0.N):
            raise ValueError("sum terms must be plain structured operators of one batch and size")
    return _with_diag(OperatorDescriptor(_hip.LO_OP_SUM, t0.B, t0.N, batch_shape=t0.batch_shape, terms=terms), d,
                      const_diag)


def _flat(t: torch.Tensor, keep: int) -> torch.Tensor:
    """[*batch, <keep trailing dims>] -> contiguous [B, ...]."""
    t = t.contiguous()
    return t.reshape(-1, *t.shape[t.dim() - keep:])


def lowrank_diag_descriptor(Croot: torch.Tensor, d: Optional[torch.Tensor], const_diag: bool = False):
    _hip.require_hip(Croot, d)
    batch = Croot.shape[:-2]
    N, R = Croot.shape[-2:]
    C3 = _flat(Croot, 2)
    return _with_diag(OperatorDescriptor(_hip.LO_OP_LOWRANK_DIAG, C3.shape[0], N, A0=C3, R=R, batch_shape=batch), d,
                      const_diag)


def dense_diag_descriptor(K: torch.Tensor, d: Optional[torch.Tensor], const_diag: bool = False):
    _hip.require_hip(K, d)
    batch = K.shape[:-2]
    N = K.shape[-1]
    K3 = _flat(K, 2)
    return _with_diag(OperatorDescriptor(_hip.LO_OP_DENSE_DIAG, K3.shape[0], N, A0=K3, batch_shape=batch), d,
                      const_diag)


def kron_diag_descriptor(K1: torch.Tensor, K2: torch.Tensor, d: Optional[torch.Tensor], const_diag: bool = False):
    _hip.require_hip(K1, K2, d)
    batch = K1.shape[:-2]
    n1, n2 = K1.shape[-1], K2.shape[-1]
    A = _flat(K1, 2)
    Bm = _flat(K2, 2)
    return _with_diag(OperatorDescriptor(_hip.LO_OP_KRON_DIAG, A.shape[0], n1 * n2, A0=A, A1=Bm, R=n1, n2=n2,
                                         batch_shape=batch), d, const_diag)


def _with_diag(desc: OperatorDescriptor, d, const_diag):
    if d is None:
        return desc
    if const_diag:
        desc.d = d.contiguous().reshape(-1)
        assert desc.d.numel() == desc.B, "constant diagonal must hold one value per batch member"
        desc.diag_mode = _hip.LO_DIAG_CONST
    else:
        desc.d = _flat(d, 1)
        assert desc.d.shape == (desc.B, desc.N)
        desc.diag_mode = _hip.LO_DIAG_FULL
    return desc


# ------------------------------------------------------------------------------------------------
def matvec(desc: OperatorDescriptor, v: torch.Tensor) -> torch.Tensor:
    """y = A v for v [*batch, N, c] (LinearOperator._matmul of the lowered operator)."""
    lib = _hip.load()
    _hip.require_hip(v)
    c = v.shape[-1]
    v3 = _flat(v, 2)
    if v3.shape[0] != desc.B or v3.shape[1] != desc.N:
        raise RuntimeError(f"matvec: rhs of shape {tuple(v.shape)} does not match operator batch {desc.B} x N {desc.N}")
    y = torch.empty_like(v3)
    s = desc.c_struct()
    ws_bytes = lib.lo_matvec_workspace_bytes(C.byref(s), c)
    ws = _hip.workspace(ws_bytes, v.device)
    _hip.check(lib.lo_matvec_f32(C.byref(s), _hip.ptr(v3), _hip.ptr(y), c, _hip.ptr(ws), ws.numel(),
                                 _hip.stream_ptr(v.device)), "lo_matvec_f32")
    return y.reshape(v.shape)


_CACHE_GENERATIONS = itertools.count(1)


@dataclass
class WoodburyPreconditioner:
    """Device form of the reference's cached (Q, noise) pair, added_diag_linear_operator.py:63-70."""

    Q: Optional[torch.Tensor]  # [B, N, ldq]; None for a root-form-only preconditioner (built on demand: ensure_q)
    dinv: torch.Tensor  # [B, N] or [B]
    k: int
    constant_diag: bool
    logdet: Optional[torch.Tensor] = None  # [B]
    # root form (lo_precond_desc.F / EF / E): P^-1 r = (r - C F C^T (r o dinv)) o dinv for a low-rank operator whose
    # pivoted-Cholesky factor this preconditioner was built from; [B, rf_ld, rf_ld] each
    F: Optional[torch.Tensor] = None
    EF: Optional[torch.Tensor] = None
    E: Optional[torch.Tensor] = None
    # R-space form (lo_precond_desc.RS): fp64 [B, 6, rf_ld, rf_ld] = E | F E | E F E | C^T C | F | E F -- single-column result-only solves
    # run their iterations on R + 1 coordinates (csrc/lo_rspace.hip)
    RS: Optional[torch.Tensor] = None
    # diagonal form of the R-space iteration (lo_precond_desc.RSD, csrc/lo_eigform.hip): fp64 [B, 6, rf_ld, rf_ld], built by
    # ensure_eigform() once the cache has served enough single-column solves to pay for it (_eigform_due)
    RSD: Optional[torch.Tensor] = None
    rsd_refused: Optional[str] = None  # why lo_precond_eigform_f32's form is not used for this cache (None: not built yet / in use)
    rs_rank: int = 0      # true rank of the root the RS form was built from
    rs_uses: int = 0      # single-column result-only solves this cache has served
    source: Optional[tuple] = None  # (L, d) the preconditioner was built from (to build Q later)
    # Kronecker root form (lo_precond_desc.kron_*): (kron_a [B, n1, 16], kron_b [B, n2, 16], kron_F [B, 16, 16]) of a
    # Kronecker operator with a constant diagonal -- the single-column CG of large N forms the rows of the tall matrix
    # on the fly instead of streaming Q
    kron: Optional[tuple] = None
    kron_kappa: Optional[float] = None  # worst rounding amplification of the Kronecker root form over the batch
    rebuild: Optional[Callable] = None  # root form from the fused solve (no L exists): () -> the full preconditioner
    # lo_precond_desc.generation: one number per cache OBJECT (never 0), part of the key of the library's "this solve missed
    # the stop rule in its result-only pass" memo -- a tensor address the allocator hands out again does not inherit it
    generation: int = field(default_factory=lambda: next(_CACHE_GENERATIONS))

    @property
    def rf_ld(self) -> int:
        return 0 if self.F is None else int(self.F.shape[-1])

    def ensure_q(self) -> "WoodburyPreconditioner":
        """Build the generic (Q, dinv) form if this preconditioner only carries the root form."""
        if self.Q is None:
            if self.source is not None:
                L, d = self.source
                full = precond_build(L, d, self.constant_diag)
            elif self.rebuild is not None:
                full = self.rebuild().ensure_q()
            else:
                raise _hip.HipExtensionError("root-form-only preconditioner without its factor: Q cannot be built")
            self.Q, self.dinv, self.k = full.Q, full.dinv, full.k
        return self

    def ensure_eigform(self) -> "WoodburyPreconditioner":
        """Build the diagonal form (lo_precond_eigform_f32) next to the R-space form.  The cache keeps the dense form when a
        member refuses it (`rsd_refused`: an eigenvalue of the preconditioned member not positive, or a change of basis that
        would cost accuracy -- DESIGN 4.14); one attempt per cache."""
        if self.RSD is None and self.rsd_refused is None and self.RS is not None:
            if self.rs_rank < 2 or self.rs_rank % 2:
                self.rsd_refused = "root rank"
                return self
            lib = _hip.load()
            B, ld = self.RS.shape[0], self.RS.shape[-1]
            RSD = torch.empty_like(self.RS)
            _hip.check(lib.lo_precond_eigform_f32(_hip.ptr(self.RS), B, self.rs_rank, ld, _hip.ptr(RSD),
                                                  _hip.stream_ptr(self.RS.device)), "lo_precond_eigform_f32")
            worst = float(RSD[:, 5, 1, 0].amin().item())
            if worst > 0:
                self.RSD = RSD
            else:
                self.rsd_refused = "ill-conditioned basis" if worst == -2.0 else "eigenvalue not positive"
        return self

    def c_struct(self) -> _hip.PrecondDesc:
        s = _hip.PrecondDesc()
        s.k, s.constant_diag, s.reserved = self.k, int(self.constant_diag), 0
        s.generation = (self.generation & 0x7FFFFFFF) or 1
        s.ldq = padded_rank(self.k) if self.Q is None else self.Q.shape[-1]
        s.Q = None if self.Q is None else self.Q.data_ptr()
        s.dinv = self.dinv.data_ptr()
        if self.F is not None:
            s.F, s.EF, s.rf_ld = self.F.data_ptr(), self.EF.data_ptr(), self.rf_ld
            s.E = None if self.E is None else self.E.data_ptr()
            s.RS = None if self.RS is None else self.RS.data_ptr()
            s.RSD = self.RSD.data_ptr() if (self.RS is not None and self.RSD is not None) else None
        if self.kron is not None:
            s.kron_a, s.kron_b, s.kron_F = (t.data_ptr() for t in self.kron)
        return s


@dataclass
class CGResult:
    x: torch.Tensor
    t_mat: Optional[torch.Tensor]
    iterations: int
    matvecs: int
    tolerance_reached: bool
    nan_detected: bool
    skipped: bool
    mean_residual: float


def _wrap_closure(fn: Callable, device):
    """Python closure (tensor -> tensor) as an lo_matvec_cb."""
    err = []

    def cb(user, v_ptr, y_ptr, B, N, c, stream):
        try:
            v = _hip.as_tensor(v_ptr, (B, N, c), device)
            y = _hip.as_tensor(y_ptr, (B, N, c), device)
            out = fn(v)
            y.copy_(out.reshape(B, N, c))
            return 0
        except BaseException as e:  # noqa: BLE001 -- must not unwind through C
            err.append(e)
            return 1

    return _hip.MATVEC_CB(cb), err


@dataclass
class MinresResult:
    x: torch.Tensor  # [n_shifts, *batch, N, c]
    iterations: int
    matvecs: int
    converged: bool
    conv: float


def minres_solve(desc: Optional[OperatorDescriptor], rhs: torch.Tensor, shifts: torch.Tensor, *,
                 value: Optional[float] = None, precond: Optional[WoodburyPreconditioner] = None,
                 matvec_closure: Optional[Callable] = None, precond_closure: Optional[Callable] = None,
                 closure_batch_shape=None, max_iter: int = 1000, tolerance: float = 1e-4,
                 eps: float = 1e-25) -> MinresResult:
    """lo_minres_f32: the reference's shifted MINRES (utils/minres.py:10-207) on the device.
    rhs [*batch, N, c]; shifts [Q] or [Q, *batch]; returns x [Q, *batch, N, c] (scaled back by the rhs norms)."""
    lib = _hip.load()
    _hip.require_hip(rhs, shifts)
    N, c = rhs.shape[-2:]
    rhs3 = _flat(rhs, 2)
    B = rhs3.shape[0]
    dev = rhs.device
    Q = shifts.shape[0]
    per_member = shifts.dim() > 1
    sh = shifts.to(torch.float32)
    sh = (sh.expand(Q, *rhs.shape[:-2]).reshape(Q, B) if per_member else sh.reshape(Q)).contiguous()
    if desc is None:
        if matvec_closure is None:
            raise ValueError("need an operator descriptor or a matvec closure")
        s = _hip.OpDesc()
        s.kind, s.diag_mode, s.B, s.N, s.R, s.n2 = _hip.LO_OP_CALLBACK, _hip.LO_DIAG_NONE, B, N, 0, 0
        bshape = tuple(closure_batch_shape) if closure_batch_shape is not None else tuple(rhs.shape[:-2])
        mv_cb, mv_err = _wrap_closure(lambda v: matvec_closure(v.reshape(*bshape, N, c)), dev)
    else:
        if desc.B != B or desc.N != N:
            raise RuntimeError(f"minres: rhs {tuple(rhs.shape)} does not match operator batch {desc.B}, N {desc.N}")
        s = desc.c_struct()
        mv_cb, mv_err = _hip.MATVEC_CB(), []
    if precond_closure is not None:
        bshape_p = tuple(closure_batch_shape) if closure_batch_shape is not None else tuple(rhs.shape[:-2])
        pc_cb, pc_err = _wrap_closure(lambda v: precond_closure(v.reshape(*bshape_p, N, c)), dev)
    else:
        pc_cb, pc_err = _hip.MATVEC_CB(), []
    pre_s = precond.ensure_q().c_struct() if precond is not None else None  # (MINRES applies the Q form)
    pre_p = C.byref(pre_s) if pre_s is not None else None
    prm = _hip.MinresParams()
    prm.c, prm.n_shifts, prm.max_iter = c, Q, max_iter
    prm.has_value, prm.value = (0, 1.0) if value is None else (1, float(value))
    prm.shifts_per_member, prm.tolerance, prm.eps = int(per_member), tolerance, eps
    ws = _hip.workspace(lib.lo_minres_workspace_bytes(C.byref(s), pre_p, C.byref(prm)), dev)
    x = torch.empty(Q, *rhs3.shape, dtype=torch.float32, device=dev)
    info = _hip.MinresInfo()
    rc = lib.lo_minres_f32(C.byref(s), mv_cb, None, pre_p, pc_cb, None, C.byref(prm), _hip.ptr(rhs3), _hip.ptr(sh),
                           _hip.ptr(x), _hip.ptr(ws), ws.numel(), C.byref(info), _hip.stream_ptr(dev))
    for e in (mv_err + pc_err):
        raise e
    _hip.check(rc, "lo_minres_f32")
    return MinresResult(x.reshape(Q, *rhs.shape), info.iterations, info.matvecs, bool(info.converged),
                        float(info.conv))


def minres_solve_f64(A: Optional[torch.Tensor], rhs: torch.Tensor, shifts: torch.Tensor, *,
                     value: Optional[float] = None, matvec_closure: Optional[Callable] = None,
                     precond_closure: Optional[Callable] = None, max_iter: int = 1000, tolerance: float = 1e-4,
                     eps: float = 1e-25) -> MinresResult:
    """lo_minres_f64: the reference's shifted MINRES (utils/minres.py:10-207) with float64 operands -- what the
    reference's test/utils/test_minres.py runs.  `A` [*batch, N, N] is multiplied by the library's fp64 kernel, any
    other operator is called back once per product; `precond_closure`: any callable or None."""
    lib = _hip.load()
    _hip.require_hip(rhs, A, dtype=torch.float64)
    N, c = rhs.shape[-2:]
    rhs3 = _flat(rhs, 2)
    B = rhs3.shape[0]
    dev = rhs.device
    bshape = tuple(rhs.shape[:-2])
    Q = shifts.shape[0]
    per_member = shifts.dim() > 1
    sh = shifts.to(device=dev, dtype=torch.float64)
    sh = (sh.expand(Q, *bshape).reshape(Q, B) if per_member else sh.reshape(Q)).contiguous()
    A3 = None
    if A is not None:
        A3 = A.expand(*bshape, N, N).reshape(B, N, N).contiguous()
    elif matvec_closure is None:
        raise ValueError("need a dense fp64 operator or a matvec closure")

    def wrap(fn):
        err = []

        def cb(user, v_ptr, y_ptr, B_, N_, c_, stream):
            try:
                v = _hip.as_tensor(v_ptr, (B_, N_, c_), dev, "<f8")
                y = _hip.as_tensor(y_ptr, (B_, N_, c_), dev, "<f8")
                y.copy_(fn(v.reshape(*bshape, N_, c_)).reshape(B_, N_, c_))
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not unwind through C
                err.append(e)
                return 1

        return _hip.MATVEC_CB(cb), err

    mv_cb, mv_err = wrap(matvec_closure) if A3 is None else (_hip.MATVEC_CB(), [])
    pc_cb, pc_err = wrap(precond_closure) if precond_closure is not None else (_hip.MATVEC_CB(), [])
    prm = _hip.MinresParamsF64()
    prm.c, prm.n_shifts, prm.max_iter = c, Q, max_iter
    prm.has_value, prm.value = (0, 1.0) if value is None else (1, float(value))
    prm.shifts_per_member, prm.tolerance, prm.eps = int(per_member), tolerance, eps
    ws = _hip.workspace(lib.lo_minres_f64_workspace_bytes(B, N, C.byref(prm)), dev)
    x = torch.empty(Q, *rhs3.shape, dtype=torch.float64, device=dev)
    info = _hip.MinresInfoF64()
    rc = lib.lo_minres_f64(_hip.ptr(A3), None, mv_cb, None, pc_cb, None, C.byref(prm), B, N, _hip.ptr(rhs3),
                           _hip.ptr(sh), _hip.ptr(x), _hip.ptr(ws), ws.numel(), C.byref(info), _hip.stream_ptr(dev))
    for e in (mv_err + pc_err):
        raise e
    _hip.check(rc, "lo_minres_f64")
    return MinresResult(x.reshape(Q, *rhs.shape), info.iterations, info.matvecs, bool(info.converged),
                        float(info.conv))


# The diagonal form of the R-space iteration (csrc/lo_eigform.hip) is a cache-lifetime investment: two R x R Jacobi
# eigendecompositions per member (measured 0.45 ms + 0.45 us per member on the MI355X) against ~3.5 us per solve and
# round of 64 resident members (k_cg_rspace 200 -> 171 us at 512 members).  A cache gets the form when the solves it
# has served would have paid for it (the ski-rental rule: never more than twice the optimal cost):
#   512 members: on the 25th single-column solve, 4096 members: on the 11th, 40 members: on the 134th.
# EIGFORM_AFTER_USES >= 0 (LO_EIGFORM_AFTER_USES) fixes the count instead (0: with the first solve), -1: never; None: the model.
_e = os.environ.get("LO_EIGFORM_AFTER_USES")
EIGFORM_AFTER_USES = None if _e is None else int(_e)
del _e


def _eigform_due(uses: int, B: int) -> bool:
    if EIGFORM_AFTER_USES is not None:
        return EIGFORM_AFTER_USES >= 0 and uses > EIGFORM_AFTER_USES
    build_us = 450.0 + 0.45 * B
    gain_us = 3.5 * ((B + 63) // 64)
    return uses * gain_us >= build_us


def cg_solve(desc: Optional[OperatorDescriptor], rhs: torch.Tensor, *, x0: Optional[torch.Tensor] = None,
             precond: Optional[WoodburyPreconditioner] = None, matvec_closure: Optional[Callable] = None,
             precond_closure: Optional[Callable] = None, closure_batch_shape=None, n_tridiag: int = 0,
             max_iter: int = 1000, max_tridiag_iter: int = 20, tolerance: float = 1.0, eps: float = 1e-10,
             stop_updating_after: float = 1e-10, floor_max_iter: int = 0,
             stop_reduce: Optional[Callable] = None, _retry: bool = False) -> CGResult:
    """lo_cg_solve_f32: the reference's linear_cg (utils/linear_cg.py:98-359) on the device.

    rhs [*batch, N, c].  Either `desc` (structured operator, fully native loop) or `matvec_closure`
    (opaque callable on [B,N,c] tensors).  `closure_batch_shape` reshapes [B,...] back for closures.
    """
    lib = _hip.load()
    _hip.require_hip(rhs, x0)
    N, c = rhs.shape[-2:]
    rhs3 = _flat(rhs, 2)
    B = rhs3.shape[0]
    x03 = None if x0 is None else _flat(x0.expand_as(rhs), 2)
    dev = rhs.device
    keep = []
    if desc is None:
        if matvec_closure is None:
            raise ValueError("need an operator descriptor or a matvec closure")
        s = _hip.OpDesc()
        s.kind, s.diag_mode, s.B, s.N, s.R, s.n2 = _hip.LO_OP_CALLBACK, _hip.LO_DIAG_NONE, B, N, 0, 0
        bshape = tuple(closure_batch_shape) if closure_batch_shape is not None else tuple(rhs.shape[:-2])

        def mv(v):
            return matvec_closure(v.reshape(*bshape, N, c))

        mv_cb, mv_err = _wrap_closure(mv, dev)
    else:
        if desc.B != B or desc.N != N:
            raise RuntimeError(f"cg_solve: rhs {tuple(rhs.shape)} does not match operator batch {desc.B}, N {desc.N}")
        s = desc.c_struct()
        mv_cb, mv_err = _hip.MATVEC_CB(), []
    if precond_closure is not None:
        bshape_p = tuple(closure_batch_shape) if closure_batch_shape is not None else tuple(rhs.shape[:-2])

        def pc(v):
            return precond_closure(v.reshape(*bshape_p, N, c))

        pc_cb, pc_err = _wrap_closure(pc, dev)
    else:
        pc_cb, pc_err = _hip.MATVEC_CB(), []
    keep += [mv_cb, pc_cb]
    # The diagonal form (0.5 - 0.7 ms of Jacobi work, 25 MB at 512 x 32, one blocking read-back) is built once the cache has
    # served enough solves ON THE RESIDENT R-SPACE KERNEL to pay for it: solves that kernel does not take (members beyond its
    # range, the batch-global stop rule, a cool-down of the resident kernels, LO_OC_NO_RSPACE) are not counted -- they are
    # counted AFTER the solve, from what lo_cg_last_executed says ran.
    rs_candidate = (precond is not None and precond.RS is not None and c == 1 and not n_tridiag and x0 is None
                    and stop_reduce is None and not _retry)
    if rs_candidate and precond.RSD is None and precond.rsd_refused is None and _eigform_due(precond.rs_uses + 1, B) \
            and (precond.rs_uses > 0 or EIGFORM_AFTER_USES is not None):
        precond.ensure_eigform()
    if stop_reduce is not None and precond is not None and precond.Q is None and precond_closure is None:
        # batch-global stop rule over ranks: a root-form-only cache could send THIS rank through the "unsupported -> build Q
        # -> start over" retry below while another rank continues -- the collective counts would diverge.  Q up front.
        precond.ensure_q()
    pre_s = precond.c_struct() if precond is not None else None
    prm = _hip.CgParams()
    prm.c, prm.n_tridiag, prm.max_iter, prm.max_tridiag_iter = c, n_tridiag, max_iter, max_tridiag_iter
    prm.floor_max_iter = floor_max_iter  # the reference's stop-rule floors use the unclipped max_iter (:303-305)
    prm.tolerance, prm.eps, prm.stop_updating_after = tolerance, eps, stop_updating_after
    sr_err = []
    if stop_reduce is not None:  # batch-global stopping rule over the ranks of a process group (distributed.py)
        def _sr(user, vals):
            try:
                out = stop_reduce([vals[0], vals[1], vals[2]])
                vals[0], vals[1], vals[2] = float(out[0]), float(out[1]), float(out[2])
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not unwind through C
                sr_err.append(e)
                return 1

        prm.stop_reduce = _hip.STOP_REDUCE_CB(_sr)
        keep.append(prm.stop_reduce)
    ws_bytes = lib.lo_cg_workspace_bytes(C.byref(s), C.byref(pre_s) if pre_s is not None else None, C.byref(prm))
    ws = _hip.workspace(ws_bytes, dev)
    x = torch.empty_like(rhs3)
    t_mat = None
    if n_tridiag:
        t_mat = torch.empty(n_tridiag, B, max_tridiag_iter, max_tridiag_iter, dtype=torch.float32, device=dev)
    info = _hip.CgInfo()
    rc = lib.lo_cg_solve_f32(C.byref(s), mv_cb, None, C.byref(pre_s) if pre_s is not None else None, pc_cb, None,
                             C.byref(prm), _hip.ptr(rhs3), _hip.ptr(x03), _hip.ptr(x), _hip.ptr(t_mat), _hip.ptr(ws),
                             ws.numel(), C.byref(info), _hip.stream_ptr(dev))
    for e in (mv_err + pc_err + sr_err):
        raise e
    if rc == _hip.LO_ERR_UNSUPPORTED and precond is not None and precond.Q is None:
        # root-form-only preconditioner and the operator-resident kernels could not take the solve (shape outside
        # their range, or a group hand-off timed out): build the generic Q form and run again
        precond.ensure_q()
        return cg_solve(desc, rhs, x0=x0, precond=precond, matvec_closure=matvec_closure,
                        precond_closure=precond_closure, closure_batch_shape=closure_batch_shape, n_tridiag=n_tridiag,
                        max_iter=max_iter, max_tridiag_iter=max_tridiag_iter, tolerance=tolerance, eps=eps,
                        stop_updating_after=stop_updating_after, floor_max_iter=floor_max_iter, stop_reduce=stop_reduce,
                        _retry=True)
    _hip.check(rc, "lo_cg_solve_f32")
    if rs_candidate:
        ex = _hip.CgPlan()
        if lib.lo_cg_last_executed(C.byref(ex)) == 0 and ex.rspace == 2:
            precond.rs_uses += 1
    if t_mat is not None:
        m = info.last_tridiag_iter + 1
        t_mat = t_mat[:, :, :m, :m].contiguous()
    return CGResult(x.reshape(rhs.shape), t_mat, info.iterations, info.matvecs, bool(info.tolerance_reached),
                    bool(info.nan_detected), bool(info.skipped), float(info.mean_residual))


def _plan_dict(pl) -> dict:
    return dict(resident=bool(pl.resident), resident_iterations=pl.resident_iterations, lockstep_cols=pl.lockstep_cols,
                lockstep_group=pl.lockstep_group, serial_engine=_hip.ENGINE_NAMES[pl.serial_engine],
                serial_group=pl.serial_group, lean=bool(pl.lean), needs_q=bool(pl.needs_q),
                streaming_precond=_hip.STREAM_PRE_NAMES[pl.streaming_precond], poll_chunk=pl.poll_chunk,
                first_stop_iteration=pl.first_stop_iteration, streaming_iterations=pl.reserved,
                rspace={0: "none", 1: "cols", 2: "resident"}[pl.rspace], rspace_diag=bool(pl.reserved2))


def cg_plan(desc: OperatorDescriptor, c: int, *, precond: Optional[WoodburyPreconditioner] = None, has_x0: bool = False,
            has_precond_closure: bool = False, n_tridiag: int = 0, max_iter: int = 1000, max_tridiag_iter: int = 20,
            floor_max_iter: int = 0, global_rule: bool = False, cus: int = 0) -> dict:
    """lo_cg_plan_f32: which engines `cg_solve` will run for these arguments (pure; nothing is launched).  cus = 0:
    this device."""
    lib = _hip.load()
    s = desc.c_struct()
    pre_s = precond.c_struct() if precond is not None else None
    prm = _hip.CgParams()
    prm.c, prm.n_tridiag, prm.max_iter, prm.max_tridiag_iter, prm.floor_max_iter = c, n_tridiag, max_iter, max_tridiag_iter, floor_max_iter
    prm.tolerance, prm.eps, prm.stop_updating_after = 1.0, 1e-10, 1e-10
    if global_rule:
        prm.stop_reduce = _hip.STOP_REDUCE_CB(lambda user, vals: 0)
    out = _hip.CgPlan()
    _hip.check(lib.lo_cg_plan_f32(C.byref(s), C.byref(pre_s) if pre_s is not None else None, int(has_precond_closure),
                                  int(has_x0), C.byref(prm), int(cus), C.byref(out)), "lo_cg_plan_f32")
    return _plan_dict(out)


def cg_last_executed() -> dict:
    """lo_cg_last_executed: the plan the calling thread's last `cg_solve` actually ran (after run-time fall-backs)."""
    out = _hip.CgPlan()
    _hip.check(_hip.load().lo_cg_last_executed(C.byref(out)), "lo_cg_last_executed")
    return _plan_dict(out)


@dataclass
class FusedSolveResult:
    """Outcome of the one-launch end-to-end solve: the linear_cg result, the root-form preconditioner it built on the
    way (for later solves with the same operator) and the recorded pivots."""

    cg: CGResult
    precond: WoodburyPreconditioner  # root form only (Q is None); `source` is None: L was never materialised
    swaps: torch.Tensor  # [B, rank] int32: position exchanged with position m at pivot m
    rank: int

    def permutation(self, N: int) -> torch.Tensor:
        """The reference's permutation [B, N] int64 (_pivoted_cholesky.py:47-48, :67-70) from the recorded exchanges."""
        lib = _hip.load()
        B = self.swaps.shape[0]
        perm = torch.empty(B, N, dtype=torch.int64, device=self.swaps.device)
        _hip.check(lib.lo_solve_fused_perm(_hip.ptr(self.swaps), B, N, self.rank, _hip.ptr(perm),
                                           _hip.stream_ptr(perm.device)), "lo_solve_fused_perm")
        return perm


def _cg_params(c, n_tridiag, max_iter, max_tridiag_iter, tolerance, eps, stop_updating_after, floor_max_iter):
    prm = _hip.CgParams()
    prm.c, prm.n_tridiag, prm.max_iter, prm.max_tridiag_iter = c, n_tridiag, max_iter, max_tridiag_iter
    prm.floor_max_iter = floor_max_iter
    prm.tolerance, prm.eps, prm.stop_updating_after = tolerance, eps, stop_updating_after
    return prm


_fused_ws_bytes: dict = {}  # workspace size of lo_solve_fused_f32 per (B, rank, c, max_iter, floor): a pure function


def solve_fused_supported(desc: Optional[OperatorDescriptor], c: int, rank: int, max_iter: int = 1000,
                          floor_max_iter: int = 0) -> bool:
    """True if lo_solve_fused_f32 takes this operator / right-hand side shape (see include/lo_amd.h)."""
    if desc is None or desc.kind != _hip.LO_OP_LOWRANK_DIAG or desc.A0 is None or not desc.A0.is_cuda:
        return False
    prm = _cg_params(c, 0, max_iter, min(20, max_iter), 1.0, 1e-10, 1e-10, floor_max_iter)
    s = desc.c_struct()
    return bool(_hip.load().lo_solve_fused_supported(C.byref(s), int(rank), C.byref(prm)))


def solve_fused(desc: OperatorDescriptor, rhs: torch.Tensor, rank: int, error_tol: float = 1e-3, *,
                max_iter: int = 1000, tolerance: float = 1.0, eps: float = 1e-10, stop_updating_after: float = 1e-10,
                floor_max_iter: int = 0) -> Optional[FusedSolveResult]:
    """lo_solve_fused_f32: pivoted Cholesky (rank pivots) -> root-form preconditioner -> preconditioned CG of
    AddedDiag(LowRankRoot, Diag) in ONE resident launch, the operator read once.  Returns None when the result would
    not be the reference's (a member's own pivot error reached the tolerance before `rank` pivots, the CG tolerance was
    not met at the iteration floor, or a group exchange timed out): the caller then takes the three-launch path."""
    lib = _hip.load()
    _hip.require_hip(rhs)
    N, c = rhs.shape[-2:]
    rhs3 = _flat(rhs, 2)
    B = rhs3.shape[0]
    if desc.B != B or desc.N != N:
        raise RuntimeError(f"solve_fused: rhs {tuple(rhs.shape)} does not match operator batch {desc.B}, N {desc.N}")
    dev = rhs.device
    rank = min(int(rank), N)
    prm = _cg_params(c, 0, max_iter, min(20, max_iter), tolerance, eps, stop_updating_after, floor_max_iter)
    s = desc.c_struct()  # (shapes the kernel does not take come back as LO_ERR_UNSUPPORTED from the call itself)
    R = desc.R
    const = desc.diag_mode == _hip.LO_DIAG_CONST
    x = torch.empty_like(rhs3)
    small = torch.empty(B * (3 * R * R + 1 + rank), dtype=torch.float32, device=dev)  # one allocation: F | EF | E | logdet | swaps
    F, EF, E = (small[i * B * R * R:(i + 1) * B * R * R].view(B, R, R) for i in range(3))
    logdet = small[3 * B * R * R:3 * B * R * R + B]
    swaps = small[3 * B * R * R + B:].view(torch.int32).view(B, rank)
    dinv = torch.empty(B if const else (B, N), dtype=torch.float32, device=dev)
    wkey = (B, rank, c, max_iter, floor_max_iter)
    nbytes = _fused_ws_bytes.get(wkey)
    if nbytes is None:
        nbytes = _fused_ws_bytes[wkey] = lib.lo_solve_fused_workspace_bytes(C.byref(s), rank, C.byref(prm))
    ws = _hip.workspace(nbytes, dev)
    info = _hip.FusedInfo()
    rc = lib.lo_solve_fused_f32(C.byref(s), rank, float(error_tol), C.byref(prm), _hip.ptr(rhs3), _hip.ptr(x),
                                _hip.ptr(F), _hip.ptr(EF), _hip.ptr(E), _hip.ptr(dinv), _hip.ptr(logdet),
                                _hip.ptr(swaps), _hip.ptr(ws), ws.numel(), C.byref(info), _hip.stream_ptr(dev))
    if rc == _hip.LO_ERR_UNSUPPORTED:
        return None
    _hip.check(rc, "lo_solve_fused_f32")
    if info.status != _hip.LO_FUSED_OK:
        return None
    pre = WoodburyPreconditioner(None, dinv, rank, const, logdet.reshape(desc.batch_shape) if desc.batch_shape else logdet,
                                 F, EF, E)
    cg = CGResult(x.reshape(rhs.shape), None, info.iterations, info.matvecs, bool(info.tolerance_reached),
                  bool(info.nan_detected), bool(info.skipped), float(info.mean_residual))
    return FusedSolveResult(cg, pre, swaps, rank)


def cg_solve_f64(A: Optional[torch.Tensor], diag: Optional[torch.Tensor], rhs: torch.Tensor, *,
                 x0: Optional[torch.Tensor] = None, matvec_closure: Optional[Callable] = None,
                 precond_closure: Optional[Callable] = None, n_tridiag: int = 0, max_iter: int = 1000,
                 max_tridiag_iter: int = 20, tolerance: float = 1.0, eps: float = 1e-10,
                 stop_updating_after: float = 1e-10, floor_max_iter: int = 0) -> CGResult:
    """lo_cg_solve_f64: the reference's linear_cg (utils/linear_cg.py:98-359) with float64 operands -- what every case
    of the reference's test/utils/test_linear_cg.py runs.  `A` [*batch, N, N] (+ `diag` [*batch, N]) is multiplied by
    the library's fp64 kernel; otherwise `matvec_closure` is called back once per product.  `precond_closure`: any
    callable or None."""
    lib = _hip.load()
    _hip.require_hip(rhs, x0, A, diag, dtype=torch.float64)
    N, c = rhs.shape[-2:]
    rhs3 = _flat(rhs, 2)
    B = rhs3.shape[0]
    dev = rhs.device
    bshape = tuple(rhs.shape[:-2])
    x03 = None if x0 is None else _flat(x0.expand_as(rhs), 2)
    A3 = d2 = None
    if A is not None:
        A3 = A.expand(*bshape, N, N).reshape(B, N, N).contiguous()
        d2 = None if diag is None else diag.expand(*bshape, N).reshape(B, N).contiguous()
    elif matvec_closure is None:
        raise ValueError("need a dense fp64 operator or a matvec closure")

    def wrap(fn):
        err = []

        def cb(user, v_ptr, y_ptr, B_, N_, c_, stream):
            try:
                v = _hip.as_tensor(v_ptr, (B_, N_, c_), dev, "<f8")
                y = _hip.as_tensor(y_ptr, (B_, N_, c_), dev, "<f8")
                y.copy_(fn(v.reshape(*bshape, N_, c_)).reshape(B_, N_, c_))
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not unwind through C
                err.append(e)
                return 1

        return _hip.MATVEC_CB(cb), err

    mv_cb, mv_err = wrap(matvec_closure) if A3 is None else (_hip.MATVEC_CB(), [])
    pc_cb, pc_err = wrap(precond_closure) if precond_closure is not None else (_hip.MATVEC_CB(), [])
    prm = _hip.CgParamsF64()
    prm.c, prm.n_tridiag, prm.max_iter, prm.max_tridiag_iter = c, n_tridiag, max_iter, max_tridiag_iter
    prm.floor_max_iter = floor_max_iter
    prm.tolerance, prm.eps, prm.stop_updating_after = tolerance, eps, stop_updating_after
    ws = _hip.workspace(lib.lo_cg_f64_workspace_bytes(B, N, C.byref(prm)), dev)
    x = torch.empty_like(rhs3)
    t_mat = None
    if n_tridiag:
        t_mat = torch.empty(n_tridiag, B, max_tridiag_iter, max_tridiag_iter, dtype=torch.float64, device=dev)
    info = _hip.CgInfoF64()
    rc = lib.lo_cg_solve_f64(_hip.ptr(A3), _hip.ptr(d2), mv_cb, None, pc_cb, None, C.byref(prm), B, N, _hip.ptr(rhs3),
                             _hip.ptr(x03), _hip.ptr(x), _hip.ptr(t_mat), _hip.ptr(ws), ws.numel(), C.byref(info),
                             _hip.stream_ptr(dev))
    for e in (mv_err + pc_err):
        raise e
    _hip.check(rc, "lo_cg_solve_f64")
    if t_mat is not None:
        m = info.last_tridiag_iter + 1
        t_mat = t_mat[:, :, :m, :m].contiguous()
    return CGResult(x.reshape(rhs.shape), t_mat, info.iterations, info.matvecs, bool(info.tolerance_reached),
                    bool(info.nan_detected), bool(info.skipped), float(info.mean_residual))


def precond_apply(pre: WoodburyPreconditioner, r: torch.Tensor) -> torch.Tensor:
    lib = _hip.load()
    _hip.require_hip(r)
    N, c = r.shape[-2:]
    r3 = _flat(r, 2)
    B = r3.shape[0]
    z = torch.empty_like(r3)
    s = pre.ensure_q().c_struct()
    ws = _hip.workspace(lib.lo_precond_apply_workspace_bytes(B, N, pre.k, c), r.device)
    _hip.check(lib.lo_precond_apply_f32(C.byref(s), _hip.ptr(r3), _hip.ptr(z), B, N, c, _hip.ptr(ws), ws.numel(),
                                        _hip.stream_ptr(r.device)), "lo_precond_apply_f32")
    return z.reshape(r.shape)


def probe_vectors(L: torch.Tensor, d: torch.Tensor, e1: torch.Tensor, e2: torch.Tensor,
                  inv_quad_rhs: Optional[torch.Tensor], batch_shape) -> tuple:
    """lo_probe_vectors_f32: the probe vectors of InvQuadLogdet.forward (functions/_inv_quad_logdet.py:91-110, :131) for
    the preconditioner L L^T + Diag(d): z = L e1 + sqrt(d) o e2, normalised, with the inv_quad columns behind them.
    L [*b, N, k] (any strides, broadcast batch), d [*b, N] or [*b, 1] / [*b] (constant), e1 [*batch, k, P], e2 [*batch, N, P].
    Returns (rhs [*batch, N, P + q], norms [*batch, 1, P]); the probes are rhs[..., :P]."""
    lib = _hip.load()
    bs = tuple(batch_shape)
    N, P = e2.shape[-2:]
    k = L.shape[-1]
    e13, e23 = _flat(e1.expand(*bs, k, P), 2), _flat(e2.expand(*bs, N, P), 2)
    B = e23.shape[0]
    _hip.require_hip(L, d, e13, e23, inv_quad_rhs)
    Lx = L.expand(*bs, N, k)
    if len(bs) > 1:  # (one batch stride is all the kernel takes: several batch dimensions are flattened by a copy)
        Lx = Lx.reshape(B, N, k)
    elif len(bs) == 0:
        Lx = Lx.unsqueeze(0)
    const = d.shape[-1] == 1 and N != 1
    d2 = (d.expand(*bs, 1) if const else d.expand(*bs, N)).contiguous().reshape(B, -1)
    q = 0 if inv_quad_rhs is None else inv_quad_rhs.shape[-1]
    iq3 = None if inv_quad_rhs is None else _flat(inv_quad_rhs.expand(*bs, N, q), 2)
    dev = e2.device
    out = torch.empty(B, N, P + q, dtype=torch.float32, device=dev)
    norms = torch.empty(B, P, dtype=torch.float32, device=dev)
    ws = _hip.workspace(lib.lo_probe_vectors_workspace_bytes(B, N, P), dev)
    rc = lib.lo_probe_vectors_f32(_hip.ptr(Lx), Lx.stride(0), Lx.stride(1), Lx.stride(2), k, _hip.ptr(d2),
                                  _hip.LO_DIAG_CONST if const else _hip.LO_DIAG_FULL, _hip.ptr(e13), _hip.ptr(e23),
                                  _hip.ptr(iq3), q, B, N, P, _hip.ptr(out), _hip.ptr(norms), _hip.ptr(ws), ws.numel(),
                                  _hip.stream_ptr(dev))
    _hip.check(rc, "lo_probe_vectors_f32")
    return out.reshape(*bs, N, P + q), norms.reshape(*bs, 1, P)


def iql_backward_factors(solves: torch.Tensor, pp: torch.Tensor, norms: torch.Tensor, g_ld: torch.Tensor,
                         g_iq: Optional[torch.Tensor], P: int):
    """lo_iql_backward_factors_f32: the element-wise part of InvQuadLogdet.backward (functions/_inv_quad_logdet.py:183-213)
    in one pass.  solves [*batch, N, P + q]; pp [*batch, N, >= P] the preconditioner applied to the NORMALISED probes;
    norms [*batch, 1, P]; g_ld [*batch]; g_iq [*batch, q] or None.  Returns (left, right [*batch, N, P + q], pre_left,
    pre_right [*batch, N, P])."""
    lib = _hip.load()
    bs = solves.shape[:-2]
    N, c = solves.shape[-2:]
    q = c - P
    s3, p3 = _flat(solves, 2), _flat(pp, 2)
    B = s3.shape[0]
    n2 = norms.expand(*bs, 1, P).contiguous().reshape(B, P)
    gl = g_ld.expand(bs).contiguous().reshape(B).to(torch.float32)
    gq = None if (g_iq is None or q == 0) else g_iq.expand(*bs, q).contiguous().reshape(B, q).to(torch.float32)
    _hip.require_hip(s3, p3, n2, gl, gq)
    left, right = torch.empty_like(s3), torch.empty_like(s3)
    pre_left = torch.empty(B, N, P, dtype=torch.float32, device=s3.device)
    pre_right = torch.empty_like(pre_left)
    rc = lib.lo_iql_backward_factors_f32(_hip.ptr(s3), _hip.ptr(p3), p3.shape[-1], _hip.ptr(n2), _hip.ptr(gl), _hip.ptr(gq),
                                         1.0 / P, B, N, P, q, _hip.ptr(left), _hip.ptr(right), _hip.ptr(pre_left),
                                         _hip.ptr(pre_right), _hip.stream_ptr(s3.device))
    _hip.check(rc, "lo_iql_backward_factors_f32")
    return (left.reshape(*bs, N, c), right.reshape(*bs, N, c), pre_left.reshape(*bs, N, P), pre_right.reshape(*bs, N, P))


# ------------------------------------------------------------------------------------------------
def pivoted_cholesky(desc: OperatorDescriptor, rank: int, error_tol: float = 1e-3, contiguous: bool = True):
    """lo_pivoted_cholesky_f32: PivotedCholesky.forward (functions/_pivoted_cholesky.py:14-105) of the
    NON-diagonal part of `desc`.  Returns (L [*batch, N, m], permutation [*batch, N] int64).
    contiguous=False skips the transposed copy of :105 and returns L as a strided view of the [B, m, N] rows the
    kernels write (what precond_build consumes directly)."""
    lib = _hip.load()
    dev = desc.device
    B, N = desc.B, desc.N
    max_rank = min(int(rank), N)
    L_rows = torch.empty(B, max_rank, N, dtype=torch.float32, device=dev)
    perm = torch.empty(B, N, dtype=torch.int64, device=dev)
    s = desc.without_diag().c_struct()
    ws = _hip.workspace(lib.lo_pivoted_cholesky_workspace_bytes(C.byref(s), max_rank), dev)
    m = C.c_int32(0)
    _hip.check(lib.lo_pivoted_cholesky_f32(C.byref(s), max_rank, float(error_tol), _hip.ptr(L_rows), _hip.ptr(perm),
                                           C.byref(m), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev)),
               "lo_pivoted_cholesky_f32")
    L = L_rows[:, : m.value, :].mT  # _pivoted_cholesky.py:105
    if contiguous:
        L = L.contiguous()
    bs = tuple(desc.batch_shape)
    return L.reshape(*bs, N, m.value), perm.reshape(*bs, N)


def pivoted_cholesky_generic(diag: torch.Tensor, row_fetch: Callable, rank: int, error_tol: float = 1e-3,
                             contiguous: bool = True):
    """lo_pivoted_cholesky_cb_f32: PivotedCholesky.forward for an operator that does not lower to a descriptor.
    diag [*batch, N] = matrix._diagonal(); row_fetch(piv [*batch] int64 device tensor) -> rows [*batch, N] is the
    reference's generic row access `matrix[..., pi_m, :]` (_pivoted_cholesky.py:81), called once per pivot without a
    host synchronisation; pivot search, Schur update and the stopping rule run in the same kernels as for lowered
    operators.  Returns (L [*batch, N, m], permutation [*batch, N])."""
    lib = _hip.load()
    f64 = diag.dtype == torch.float64  # (round 4: the reference is dtype-generic; same kernels instantiated for double)
    _hip.require_hip(diag, dtype=diag.dtype if f64 else torch.float32)
    dev = diag.device
    bs = tuple(diag.shape[:-1])
    N = diag.shape[-1]
    d2 = _flat(diag, 1)
    B = d2.shape[0]
    max_rank = min(int(rank), N)
    L_rows = torch.empty(B, max_rank, N, dtype=diag.dtype, device=dev)
    perm = torch.empty(B, N, dtype=torch.int64, device=dev)
    err = []

    def cb(user, piv_ptr, rows_ptr, B_, N_, stream):
        try:
            piv = _hip.as_tensor(piv_ptr, (B_,), dev, "<i8")
            rows = _hip.as_tensor(rows_ptr, (B_, N_), dev, "<f8" if f64 else "<f4")
            rows.copy_(row_fetch(piv.reshape(bs)).reshape(B_, N_))
            return 0
        except BaseException as e:  # noqa: BLE001 -- must not unwind through C
            err.append(e)
            return 1

    c_cb = _hip.ROWFETCH_CB(cb)
    m = C.c_int32(0)
    if f64:
        ws = _hip.workspace(lib.lo_pivoted_cholesky_cb_f64_workspace_bytes(B, N, max_rank), dev)
        rc = lib.lo_pivoted_cholesky_cb_f64(B, N, _hip.ptr(d2), c_cb, None, max_rank, float(error_tol), _hip.ptr(L_rows),
                                            _hip.ptr(perm), C.byref(m), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev))
    else:
        ws = _hip.workspace(lib.lo_pivoted_cholesky_cb_workspace_bytes(B, N, max_rank), dev)
        rc = lib.lo_pivoted_cholesky_cb_f32(B, N, _hip.ptr(d2), c_cb, None, max_rank, float(error_tol), _hip.ptr(L_rows),
                                            _hip.ptr(perm), C.byref(m), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev))
    if err:
        raise err[0]
    _hip.check(rc, "lo_pivoted_cholesky_cb_f64" if f64 else "lo_pivoted_cholesky_cb_f32")
    L = L_rows[:, : m.value, :].mT
    if contiguous:
        L = L.contiguous()
    return L.reshape(*bs, N, m.value), perm.reshape(*bs, N)


def _root_form(lib, root, perm, L3, d, constant_diag, B, N, k, dev) -> WoodburyPreconditioner:
    R = root.shape[-1]
    C3 = _flat(root, 2)
    p2 = _flat(perm, 1)
    ld = padded_rank(R)
    F = torch.empty(B, ld, ld, dtype=torch.float32, device=dev)
    EF = torch.empty_like(F)
    E = torch.empty_like(F)
    if constant_diag:
        d2 = d.contiguous().reshape(-1)
        dinv = torch.empty(B, dtype=torch.float32, device=dev)
        mode = _hip.LO_DIAG_CONST
    else:
        d2 = _flat(d, 1)
        dinv = torch.empty(B, N, dtype=torch.float32, device=dev)
        mode = _hip.LO_DIAG_FULL
    logdet = torch.empty(B, dtype=torch.float32, device=dev)
    sm, sr, sc = L3.stride()
    if B == 1:
        sm = 0
    if R % 4 == 0 and C3.data_ptr() % 16 == 0 and not os.environ.get("LO_NO_RSPACE_FORM"):
        # with the R-space form: E and C^T C on the fp64 matrix cores (the fp32 F / EF / E are roundings of the same)
        RS = torch.empty(B, 6, ld, ld, dtype=torch.float64, device=dev)
        ws = _hip.workspace(lib.lo_precond_root_form_rs_workspace_bytes(B, N, R), dev)
        _hip.check(lib.lo_precond_root_form_rs_f32(_hip.ptr(C3), R, _hip.ptr(d2), mode, _hip.ptr(L3), sm, sr, sc,
                                                   _hip.ptr(p2), B, N, k, ld, _hip.ptr(F), _hip.ptr(EF), _hip.ptr(E),
                                                   _hip.ptr(dinv), _hip.ptr(logdet), _hip.ptr(RS), _hip.ptr(ws),
                                                   ws.numel(), _hip.stream_ptr(dev)), "lo_precond_root_form_rs_f32")
        return WoodburyPreconditioner(None, dinv, k, constant_diag, logdet, F, EF, E, RS, rs_rank=R)
    ws = _hip.workspace(lib.lo_precond_root_form_workspace_bytes(B, N, R), dev)
    _hip.check(lib.lo_precond_root_form_f32(_hip.ptr(C3), R, _hip.ptr(d2), mode, _hip.ptr(L3), sm, sr, sc, _hip.ptr(p2),
                                            B, N, k, ld, _hip.ptr(F), _hip.ptr(EF), _hip.ptr(E), _hip.ptr(dinv),
                                            _hip.ptr(logdet), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev)),
               "lo_precond_root_form_f32")
    return WoodburyPreconditioner(None, dinv, k, constant_diag, logdet, F, EF, E)


# the Kronecker root form multiplies the fp32 rounding of KP^T (r/d) by `kappa` (lo_amd.h); beyond this bound the
# pivot rows are too close to dependent for 1e-4 solves and the orthonormal Q form is kept
KRON_ROOT_MAX_KAPPA = 200.0


def _kron_root(lib, desc: "OperatorDescriptor", perm, L3, k, dev):
    """lo_precond_kron_root_f32 for a Kronecker operator with a constant diagonal (descriptor with the diagonal
    attached) -> ((kron_a, kron_b, kron_F) or None when some member's pivot rows are too ill conditioned, worst kappa);
    one read-back of B floats."""
    B, n1, n2 = desc.B, desc.R, desc.n2
    ka = torch.empty(B, n1, 16, dtype=torch.float32, device=dev)
    kb = torch.empty(B, n2, 16, dtype=torch.float32, device=dev)
    kF = torch.empty(B, 16, 16, dtype=torch.float32, device=dev)
    kappa = torch.empty(B, dtype=torch.float32, device=dev)
    p2 = _flat(perm, 1)
    ws = _hip.workspace(lib.lo_precond_kron_root_workspace_bytes(B), dev)
    sm, sr, sc = L3.stride()
    if B == 1:
        sm = 0
    s = desc.c_struct()
    _hip.check(lib.lo_precond_kron_root_f32(C.byref(s), _hip.ptr(L3), sm, sr, sc, _hip.ptr(p2), k, _hip.ptr(ka),
                                            _hip.ptr(kb), _hip.ptr(kF), _hip.ptr(kappa), _hip.ptr(ws), ws.numel(),
                                            _hip.stream_ptr(dev)), "lo_precond_kron_root_f32")
    worst = float(kappa.max())
    if not (worst < KRON_ROOT_MAX_KAPPA):  # (also NaN)
        return None, worst
    return (ka, kb, kF), worst


def padded_rank(k: int) -> int:
    rq, p = (k + 3) // 4, 1
    while p < rq:
        p <<= 1
    return 4 * p


def precond_build(L: torch.Tensor, d: torch.Tensor, constant_diag: bool, root: Optional[torch.Tensor] = None,
                  perm: Optional[torch.Tensor] = None, need_q: bool = True,
                  kron: Optional["OperatorDescriptor"] = None) -> WoodburyPreconditioner:
    """lo_precond_build_f32: AddedDiagLinearOperator._init_cache* (added_diag_linear_operator.py:144-184).
    L [*batch, N, k]; d [*batch, N] (or [*batch] when constant_diag).
    root, perm: the root C [*batch, N, R <= 32] of the low-rank operator L was factored from and the permutation
    pivoted_cholesky returned -- adds the ROOT FORM (lo_precond_root_form_f32) the operator-resident CG kernels
    prefer (one all-reduce per iteration, no second tall matrix); need_q=False then skips the generic Q.
    kron, perm: the descriptor (LO_OP_KRON_DIAG, constant diagonal) of the Kronecker operator L was factored from --
    adds the Kronecker root form (lo_precond_kron_root_f32) for the single-column CG of 8192 <= N <= 65536."""
    lib = _hip.load()
    _hip.require_hip(L, d)
    N, k = L.shape[-2:]
    L3 = L.reshape(-1, N, k)  # stays a view for the strided rows layout pivoted_cholesky(contiguous=False) returns
    if min(L3.stride()) < 0 or L3.dtype != torch.float32:
        L3 = L3.contiguous()
    B = L3.shape[0]
    dev = L.device
    if root is not None and perm is not None and root.shape[-1] <= 32 and 1 <= k <= 32:
        rf = _root_form(lib, root, perm, L3, d, constant_diag, B, N, k, dev)
        rf.source = (L, d)
        if not need_q:
            return rf
        full = precond_build(L, d, constant_diag)
        full.F, full.EF, full.E, full.RS, full.source = rf.F, rf.EF, rf.E, rf.RS, rf.source
        full.rs_rank = rf.rs_rank
        return full
    ldq = padded_rank(k)
    Q = torch.empty(B, N, ldq, dtype=torch.float32, device=dev)
    if constant_diag:
        d2 = d.contiguous().reshape(-1)
        assert d2.numel() == B
        dinv = torch.empty(B, dtype=torch.float32, device=dev)
        mode = _hip.LO_DIAG_CONST
    else:
        d2 = _flat(d, 1)
        assert d2.shape == (B, N)
        dinv = torch.empty(B, N, dtype=torch.float32, device=dev)
        mode = _hip.LO_DIAG_FULL
    logdet = torch.empty(B, dtype=torch.float32, device=dev)
    ws = _hip.workspace(lib.lo_precond_build_workspace_bytes(B, N, k), dev)
    sm, sr, sc = L3.stride()
    if B == 1:
        sm = 0
    _hip.check(lib.lo_precond_build_strided_f32(_hip.ptr(L3), sm, sr, sc, _hip.ptr(d2), mode, B, N, k, _hip.ptr(Q),
                                                _hip.ptr(dinv), _hip.ptr(logdet), _hip.ptr(ws), ws.numel(),
                                                _hip.stream_ptr(dev)),
               "lo_precond_build_strided_f32")
    out = WoodburyPreconditioner(Q, dinv, k, constant_diag, logdet.reshape(L.shape[:-2]))
    if (kron is not None and perm is not None and constant_diag and kron.kind == _hip.LO_OP_KRON_DIAG
            and kron.diag_mode == _hip.LO_DIAG_CONST and kron.B == B and kron.N == N and 1 <= k <= 16
            and 8192 <= N <= 65536):
        out.kron, out.kron_kappa = _kron_root(lib, kron, perm, L3, k, dev)
    return out


def tridiag_eigh_slq(t_mat: torch.Tensor, n: int, want_evecs: bool = False, want_logdet: bool = True):
    """lo_tridiag_eigh_slq_f32 on t_mat [P, *batch, T, T] -> (evals [P,*batch,T], evecs or None, logdet [*batch])."""
    lib = _hip.load()
    _hip.require_hip(t_mat)
    P = t_mat.shape[0]
    T = t_mat.shape[-1]
    batch = t_mat.shape[1:-2]
    t3 = t_mat.contiguous().reshape(-1, T, T)
    B = t3.shape[0] // P
    dev = t_mat.device
    evals = torch.empty(P * B, T, dtype=torch.float32, device=dev)
    evecs = torch.empty(P * B, T, T, dtype=torch.float32, device=dev) if want_evecs else None
    logdet = torch.empty(B, dtype=torch.float32, device=dev) if want_logdet else None
    ws = _hip.workspace(lib.lo_tridiag_eigh_slq_workspace_bytes(P, B), dev)
    _hip.check(lib.lo_tridiag_eigh_slq_f32(_hip.ptr(t3), P, B, T, int(n), _hip.ptr(evals), _hip.ptr(evecs),
                                           _hip.ptr(logdet), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev)),
               "lo_tridiag_eigh_slq_f32")
    return (evals.reshape(P, *batch, T), None if evecs is None else evecs.reshape(P, *batch, T, T),
            None if logdet is None else logdet.reshape(batch))


def lanczos_tridiag(desc: Optional[OperatorDescriptor], init_vecs: torch.Tensor, max_iter: int, tol: float = 1e-5,
                    matvec_closure: Optional[Callable] = None, contiguous: bool = False):
    """lo_lanczos_tridiag_f32: utils/lanczos.py:9-164.  init_vecs [*batch, N, P].
    Returns (q_mat [P,*batch,N,k'], t_mat [P,*batch,k',k']) with the reference's output shapes (:151-161); q_mat is a
    strided view of the basis as the kernels wrote it unless `contiguous=True` asks for the reference's memory layout
    (one more pass over the basis: lo_lanczos_permute_f32)."""
    lib = _hip.load()
    _hip.require_hip(init_vecs)
    N, P = init_vecs.shape[-2:]
    batch = tuple(init_vecs.shape[:-2])
    v3 = _flat(init_vecs, 2)
    B = v3.shape[0]
    dev = init_vecs.device
    if desc is None:
        s = _hip.OpDesc()
        s.kind, s.diag_mode, s.B, s.N, s.R, s.n2 = _hip.LO_OP_CALLBACK, _hip.LO_DIAG_NONE, B, N, 0, 0

        def mv(v):
            return matvec_closure(v.reshape(*batch, N, P))

        cb, err = _wrap_closure(mv, dev)
    else:
        s = desc.c_struct()
        cb, err = _hip.MATVEC_CB(), []
    max_iter = int(max_iter)
    q = torch.empty(max_iter, B, N, P, dtype=torch.float32, device=dev)
    t = torch.empty(max_iter, max_iter, B, P, dtype=torch.float32, device=dev)
    ws = _hip.workspace(lib.lo_lanczos_workspace_bytes(C.byref(s), P, max_iter), dev)
    iters = C.c_int32(0)
    rc = lib.lo_lanczos_tridiag_f32(C.byref(s), cb, None, _hip.ptr(v3), P, max_iter, float(tol), _hip.ptr(q),
                                    _hip.ptr(t), C.byref(iters), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev))
    for e in err:
        raise e
    _hip.check(rc, "lo_lanczos_tridiag_f32")
    k = iters.value
    nb = len(batch)
    t = t[:k, :k].reshape(k, k, *batch, P)
    if contiguous:
        q_out = torch.empty(P, *batch, N, k, dtype=torch.float32, device=dev)
        rc = lib.lo_lanczos_permute_f32(_hip.ptr(q), k, B, N, P, _hip.ptr(q_out), _hip.stream_ptr(dev))  # lanczos.py:154
        if rc == _hip.LO_ERR_UNSUPPORTED:
            q_out = q[:k].reshape(k, *batch, N, P).permute(-1, *range(1, 1 + nb), -2, 0).contiguous()
        else:
            _hip.check(rc, "lo_lanczos_permute_f32")
    else:
        # the reference's q_mat [P, *batch, N, k] (lanczos.py:154) as a VIEW of the basis in the layout the step kernels
        # write it ([k, B, N, P]): same shape, same values, no 5 GB copy at the cfg3 shape.  The consumers on the path
        # (root_from_lanczos below) read this layout directly; anything else sees an ordinary strided tensor.
        # (an early stop -- k well below the allocated steps -- must not keep the whole basis buffer alive behind the view:
        #  ADVICE r4)
        qk = q[:k] if k == q.shape[0] else q[:k].clone()
        q_out = qk.reshape(k, *batch, N, P).permute(-1, *range(1, 1 + nb), -2, 0)
    t_out = t.permute(-1, *range(2, 2 + nb), 0, 1).contiguous()  # :156
    if P == 1:  # squeeze_(0) (:159-161) only acts on a size-1 leading dim
        q_out, t_out = q_out[0], t_out[0]
    return q_out, t_out


def lanczos_tridiag_f64(A: Optional[torch.Tensor], diag: Optional[torch.Tensor], init_vecs: torch.Tensor,
                        max_iter: int, tol: float = 1e-5, matvec_closure: Optional[Callable] = None):
    """lo_lanczos_tridiag_f64: utils/lanczos.py:9-164 with float64 operands (the reference is dtype-generic).  `A`
    [*batch, N, N] (+ `diag` [*batch, N]) is multiplied by the library's fp64 kernel, otherwise `matvec_closure` is
    called back once per step.  Same return shapes as `lanczos_tridiag`; q_mat is the permuted view of the basis."""
    lib = _hip.load()
    _hip.require_hip(init_vecs, A, diag, dtype=torch.float64)
    N, P = init_vecs.shape[-2:]
    batch = tuple(init_vecs.shape[:-2])
    v3 = _flat(init_vecs, 2)
    B = v3.shape[0]
    dev = init_vecs.device
    A3 = d2 = None
    err = []
    cb = _hip.MATVEC_CB()
    if A is not None:
        A3 = A.expand(*batch, N, N).reshape(B, N, N).contiguous()
        d2 = None if diag is None else diag.expand(*batch, N).reshape(B, N).contiguous()
    elif matvec_closure is None:
        raise ValueError("need a dense fp64 operator or a matvec closure")
    else:
        def _cb(user, v_ptr, y_ptr, B_, N_, c_, stream):
            try:
                v = _hip.as_tensor(v_ptr, (B_, N_, c_), dev, "<f8")
                y = _hip.as_tensor(y_ptr, (B_, N_, c_), dev, "<f8")
                y.copy_(matvec_closure(v.reshape(*batch, N_, c_)).reshape(B_, N_, c_))
                return 0
            except BaseException as e:  # noqa: BLE001 -- must not unwind through C
                err.append(e)
                return 1

        cb = _hip.MATVEC_CB(_cb)
    max_iter = int(max_iter)
    q = torch.empty(max_iter, B, N, P, dtype=torch.float64, device=dev)
    t = torch.empty(max_iter, max_iter, B, P, dtype=torch.float64, device=dev)
    ws = _hip.workspace(lib.lo_lanczos_f64_workspace_bytes(B, N, P, max_iter), dev)
    iters = C.c_int32(0)
    rc = lib.lo_lanczos_tridiag_f64(_hip.ptr(A3), _hip.ptr(d2), cb, None, _hip.ptr(v3), B, N, P, max_iter, float(tol),
                                    _hip.ptr(q), _hip.ptr(t), C.byref(iters), _hip.ptr(ws), ws.numel(),
                                    _hip.stream_ptr(dev))
    for e in err:
        raise e
    _hip.check(rc, "lo_lanczos_tridiag_f64")
    k = iters.value
    nb = len(batch)
    q_out = (q[:k] if k == q.shape[0] else q[:k].clone()).reshape(k, *batch, N, P).permute(-1, *range(1, 1 + nb), -2, 0)
    t_out = t[:k, :k].reshape(k, k, *batch, P).permute(-1, *range(2, 2 + nb), 0, 1).contiguous()
    if P == 1:
        q_out, t_out = q_out[0], t_out[0]
    return q_out, t_out


def _native_lanczos_layout(q_mat: torch.Tensor):
    """(P, B) if q_mat [P, *batch, N, k] is the permuted view `lanczos_tridiag` returns of a basis stored [k, B, N, P]
    (element (p, b, n, a) at ((a B + b) N + n) P + p from its first element), else None.  Size-1 dimensions carry no
    information (the callers unsqueeze / squeeze them)."""
    if q_mat.dim() < 3:
        return None
    N, k = q_mat.shape[-2:]
    lead = list(q_mat.shape[:-2])
    P = lead[0]
    B = 1
    for s in lead[1:]:
        B *= s
    want = [1]  # stride of the probe dimension
    acc = N * P
    batch_strides = []
    for s in reversed(lead[1:]):
        batch_strides.append(acc)
        acc *= s
    want += list(reversed(batch_strides)) + [P, B * N * P]
    for size, have, exp in zip(q_mat.shape, q_mat.stride(), want):
        if size > 1 and have != exp:
            return None
    return P, B


def root_from_lanczos(q_mat: torch.Tensor, evecs: torch.Tensor, evals: torch.Tensor, want_root: bool = True,
                      want_inverse: bool = False):
    """lo_root_from_lanczos_f32: the epilogue of RootDecomposition.forward (functions/_root_decomposition.py:73-85).
    q_mat [*lead, N, k], evecs [*lead, k, k], evals [*lead, k] -> (q V, q V o sqrt(evals) or None, q V / sqrt(evals)
    or None).  k > 32: the same three expressions with torch (as the reference)."""
    lib = _hip.load()
    lead = q_mat.shape[:-2]
    N, k = q_mat.shape[-2:]
    if k > 32 or not (q_mat.is_cuda and q_mat.dtype == torch.float32):
        qv = q_mat.matmul(evecs)
        s = evals.sqrt().unsqueeze(-2)
        return qv, (qv * s if want_root else None), (qv / s if want_inverse else None)
    _hip.require_hip(q_mat, evecs, evals)
    native = _native_lanczos_layout(q_mat)
    if native is not None:
        P, B = native
        dev = q_mat.device
        v4 = evecs.contiguous().reshape(P * B, k, k)
        e4 = evals.contiguous().reshape(P * B, k)
        qv = torch.empty(P * B, N, k, dtype=torch.float32, device=dev)
        root = torch.empty_like(qv) if want_root else None
        inv = torch.empty_like(qv) if want_inverse else None
        rc = lib.lo_root_from_lanczos_native_f32(_hip.ptr(q_mat), _hip.ptr(v4), _hip.ptr(e4), B, N, P, k, _hip.ptr(qv),
                                                 _hip.ptr(root), _hip.ptr(inv), _hip.stream_ptr(dev))
        if rc == 0:
            shp = (*lead, N, k)
            return qv.reshape(shp), (None if root is None else root.reshape(shp)), (None if inv is None else inv.reshape(shp))
        if rc != _hip.LO_ERR_UNSUPPORTED:
            _hip.check(rc, "lo_root_from_lanczos_native_f32")
    q3 = q_mat.contiguous().reshape(-1, N, k)
    PB = q3.shape[0]
    v3 = evecs.contiguous().reshape(PB, k, k)
    e3 = evals.contiguous().reshape(PB, k)
    dev = q_mat.device
    qv = torch.empty_like(q3)
    root = torch.empty_like(q3) if want_root else None
    inv = torch.empty_like(q3) if want_inverse else None
    _hip.check(lib.lo_root_from_lanczos_f32(_hip.ptr(q3), _hip.ptr(v3), _hip.ptr(e3), PB, N, k, _hip.ptr(qv),
                                            _hip.ptr(root), _hip.ptr(inv), _hip.stream_ptr(dev)),
               "lo_root_from_lanczos_f32")
    shp = (*lead, N, k)
    return qv.reshape(shp), (None if root is None else root.reshape(shp)), (None if inv is None else inv.reshape(shp))


def _uv(left_vecs: torch.Tensor, right_vecs: torch.Tensor, batch_shape):
    """Broadcast left / right vectors [*b, N, D] to the operator's batch and flatten it."""
    if left_vecs.dim() == 1:
        left_vecs, right_vecs = left_vecs.unsqueeze(-1), right_vecs.unsqueeze(-1)
    bs = torch.broadcast_shapes(tuple(batch_shape), left_vecs.shape[:-2], right_vecs.shape[:-2])
    N, D = left_vecs.shape[-2:]
    U = left_vecs.expand(*bs, N, D).contiguous().reshape(-1, N, D)
    V = right_vecs.expand(*bs, N, D).contiguous().reshape(-1, N, D)
    _hip.require_hip(U, V, dtype=U.dtype if U.dtype == torch.float64 else torch.float32)
    return U, V, bs


def _is_f64(*ts):
    """fp64 operands of the backward contractions (the pull-backs of an fp64 linear_cg solve): plain library GEMMs /
    reductions on the device -- the hand-written contraction kernels are fp32."""
    return all(t.is_cuda and t.dtype == torch.float64 for t in ts)


def bilinear_dense(left_vecs: torch.Tensor, right_vecs: torch.Tensor, batch_shape=()):
    """lo_bilinear_dense_f32: U V^T [*batch, N, N] (dense_linear_operator.py:69-71)."""
    lib = _hip.load()
    U, V, bs = _uv(left_vecs, right_vecs, batch_shape)
    B, N, D = U.shape
    if _is_f64(U, V):
        return torch.matmul(U, V.mT).reshape(*bs, N, N)
    out = torch.empty(B, N, N, dtype=torch.float32, device=U.device)
    _hip.check(lib.lo_bilinear_dense_f32(_hip.ptr(U), _hip.ptr(V), B, N, D, _hip.ptr(out), _hip.stream_ptr(U.device)),
               "lo_bilinear_dense_f32")
    return out.reshape(*bs, N, N)


def bilinear_diag(left_vecs: torch.Tensor, right_vecs: torch.Tensor, batch_shape=(), constant: bool = False):
    """lo_bilinear_diag_f32: sum_d U o V [*batch, N], or its sum over N as [*batch, 1] for a constant diagonal
    (diag_linear_operator.py:37-45, :337-344)."""
    lib = _hip.load()
    U, V, bs = _uv(left_vecs, right_vecs, batch_shape)
    B, N, D = U.shape
    dev = U.device
    if _is_f64(U, V):
        rd = (U * V).sum(-1)
        return rd.sum(-1).reshape(*bs, 1) if constant else rd.reshape(*bs, N)
    out = torch.empty(B if constant else B * N, dtype=torch.float32, device=dev)
    ws = _hip.workspace(4 * B * N + 256, dev) if constant else None
    _hip.check(lib.lo_bilinear_diag_f32(_hip.ptr(U), _hip.ptr(V), B, N, D, 1 if constant else 0, _hip.ptr(out),
                                        _hip.ptr(ws), 0 if ws is None else ws.numel(), _hip.stream_ptr(dev)),
               "lo_bilinear_diag_f32")
    return out.reshape(*bs, 1) if constant else out.reshape(*bs, N)


def bilinear_root(root: torch.Tensor, left_vecs: torch.Tensor, right_vecs: torch.Tensor, with_rowdot: bool = False):
    """lo_bilinear_root_f32: U (V^T C) + V (U^T C) [*batch, N, R] for K = C C^T (autograd of the two-GEMM matvec).
    `with_rowdot`: also return sum_d U o V [*batch, N] (the Diag derivative for the same factors) from the same pass."""
    lib = _hip.load()
    U, V, bs = _uv(left_vecs, right_vecs, root.shape[:-2])
    B, N, D = U.shape
    R = root.shape[-1]
    Cm = root.expand(*bs, N, R).contiguous().reshape(B, N, R)
    if _is_f64(U, V, Cm):
        out = U @ (V.mT @ Cm) + V @ (U.mT @ Cm)
        if with_rowdot:
            return out.reshape(*bs, N, R), (U * V).sum(-1).reshape(*bs, N)
        return out.reshape(*bs, N, R)
    _hip.require_hip(Cm)
    dev = U.device
    # the kernel keeps a D x R tile on chip (D R <= 2048): wider problems go in column chunks, the derivative is a sum
    # over the columns
    # (phase A: D R <= 2048; phase B: T1 | T2 padded to 32-column blocks + 32 staged rows of U, V in 64 KB of LDS)
    rp = (R + 31) // 32 * 32
    dmax = max(1, min(2048 // R, (16000 // (2 * rp + 64)) & ~1))
    out, rowdot = None, None
    for d0 in range(0, D, dmax):
        Uc, Vc = (U, V) if D <= dmax else (U[..., d0:d0 + dmax].contiguous(), V[..., d0:d0 + dmax].contiguous())
        Dc = Uc.shape[-1]
        o = torch.empty(B, N, R, dtype=torch.float32, device=dev)
        rd = torch.empty(B, N, dtype=torch.float32, device=dev) if with_rowdot else None
        ws = _hip.workspace(lib.lo_bilinear_root_workspace_bytes(B, N, R, Dc), dev)
        _hip.check(lib.lo_bilinear_root_f32(_hip.ptr(Cm), _hip.ptr(Uc), _hip.ptr(Vc), B, N, R, Dc, _hip.ptr(o),
                                            _hip.ptr(rd), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev)),
                   "lo_bilinear_root_f32")
        out = o if out is None else out.add_(o)
        if with_rowdot:
            rowdot = rd if rowdot is None else rowdot.add_(rd)
    if with_rowdot:
        return out.reshape(*bs, N, R), rowdot.reshape(*bs, N)
    return out.reshape(*bs, N, R)


def root_apply_add(U: torch.Tensor, T: torch.Tensor, out: torch.Tensor) -> bool:
    """lo_root_apply_add_f32: out [*batch, N, R] += U [*batch, N, D] @ T [*batch, D, R] in place, one pass over `out`
    (the N-sized product of the pull-back through the pivoted Cholesky of a root, functions/_pivoted_cholesky.py).
    Returns False -- nothing touched -- when the kernel does not take the operands (the caller runs the library product):
    not fp32 CUDA, `out` not contiguous or not the whole of its storage (a slice of somebody else's tensor), R not 8 / 16 /
    32, D > 32, more than 65535 members."""
    if not (out.is_cuda and out.dtype == torch.float32 and U.dtype == torch.float32 and T.dtype == torch.float32
            and out.is_contiguous() and out.storage_offset() == 0
            and out.untyped_storage().nbytes() == out.numel() * 4 and not out.requires_grad) \
            or os.environ.get("LO_NO_ROOT_APPLY_ADD"):
        return False
    *bs, N, R = out.shape
    D = U.shape[-1]
    B = 1
    for x in bs:
        B *= int(x)
    if R not in (8, 16, 32) or not 1 <= D <= 32 or not 1 <= B <= 65535 or tuple(U.shape) != (*bs, N, D) \
            or tuple(T.shape[-2:]) != (D, R):
        return False
    lib = _hip.load()
    _hip.require_hip(out)
    Uc = U.contiguous()
    Tc = T.expand(*bs, D, R).contiguous()
    _hip.check(lib.lo_root_apply_add_f32(_hip.ptr(Uc), _hip.ptr(Tc), B, N, D, R, _hip.ptr(out),
                                         _hip.stream_ptr(out.device)), "lo_root_apply_add_f32")
    return True


def bilinear_kron(K1: torch.Tensor, K2: torch.Tensor, left_vecs: torch.Tensor, right_vecs: torch.Tensor):
    """lo_bilinear_kron_f32: (dK1, dK2) of sum_d u_d^T (K1 (x) K2) v_d (autograd of the Kronecker matvec)."""
    lib = _hip.load()
    bs0 = torch.broadcast_shapes(K1.shape[:-2], K2.shape[:-2])
    U, V, bs = _uv(left_vecs, right_vecs, bs0)
    B, N, D = U.shape
    n1, n2 = K1.shape[-1], K2.shape[-1]
    assert n1 * n2 == N
    A1 = K1.expand(*bs, n1, n1).contiguous().reshape(B, n1, n1)
    A2 = K2.expand(*bs, n2, n2).contiguous().reshape(B, n2, n2)
    _hip.require_hip(A1, A2)
    dev = U.device
    d1 = torch.empty(B, n1, n1, dtype=torch.float32, device=dev)
    d2 = torch.empty(B, n2, n2, dtype=torch.float32, device=dev)
    ws = _hip.workspace(lib.lo_bilinear_kron_workspace_bytes(B, n1, n2, D), dev)
    _hip.check(lib.lo_bilinear_kron_f32(_hip.ptr(A1), _hip.ptr(A2), _hip.ptr(U), _hip.ptr(V), B, n1, n2, D, _hip.ptr(d1),
                                        _hip.ptr(d2), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev)),
               "lo_bilinear_kron_f32")
    return d1.reshape(*bs, n1, n1), d2.reshape(*bs, n2, n2)


def set_onchip_cg(enable: bool):
    """Allow (default) or forbid the operator-resident CG fast path (csrc/lo_cg_onchip.hip); test / A-B switch.
    Enabling also ends a cool-down of the resident kernels (`resident_status`)."""
    _hip.load().lo_cg_set_onchip(1 if enable else 0)


def resident_status() -> dict:
    """lo_resident_status_get: the gate of the resident kernels -- hand-off timeouts seen by this process, calls still
    to be served by the streaming engines (cool-down), length of the next cool-down, re-arms so far."""
    out = _hip.ResidentStatus()
    _hip.check(_hip.load().lo_resident_status_get(C.byref(out)), "lo_resident_status_get")
    return {n: int(getattr(out, n)) for n, _ in _hip.ResidentStatus._fields_}


def inject_resident_timeouts(n: int):
    """lo_resident_inject_timeouts: the next n resident CG launches are treated as timed out (the solve is redone on
    the streaming engine and a cool-down starts) -- tests and `bench.py --inject-timeouts`."""
    _hip.check(_hip.load().lo_resident_inject_timeouts(int(n)), "lo_resident_inject_timeouts")


def peer_gather_set(bufs=(), member_offset: int = 0) -> None:
    """lo_peer_gather_set (prototype): up to seven [B_total, N] fp32 buffers the resident single-column solve writes its
    solutions into next to its own output -- the gather of SURVEY 8(e) as peer writes.  () removes them."""
    lib = _hip.load()
    n = len(bufs)
    arr = (C.c_void_p * max(n, 1))(*[b.data_ptr() for b in bufs])
    _hip.check(lib.lo_peer_gather_set(arr, n, int(member_offset)), "lo_peer_gather_set")
