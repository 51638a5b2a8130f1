"""KroneckerProductLinearOperator K1 (x) ... (x) KP -- `_matmul`, `_diagonal`, `_get_indices` only
(reference: operators/kronecker_product_linear_operator.py:20-45, 62-96, 188-216, 272-284).  Two dense factors
lower to the batched-GEMM kernel pair in csrc/lo_kron.hip; products of more dense factors are regrouped into two
dense groups (`_two_groups`) and lower the same way.  `+ Diag` / `add_diagonal` build the
KroneckerProductAddedDiagLinearOperator like the reference (:98-145): eigendecomposition closed forms for a
constant diagonal, the CG path otherwise (an explicit AddedDiagLinearOperator(kron, diag) is always the CG path)."""
from __future__ import annotations

import math

import torch
from torch import Tensor

from .. import kernels as K
from .. import settings
from ..utils.broadcasting import _matmul_broadcast_shape
from ._linear_operator import LinearOperator
from .dense_linear_operator import DenseLinearOperator, to_linear_operator
from .diag_linear_operator import DiagLinearOperator


def _kron_diag(*ops) -> Tensor:
    lead = ops[0]._diagonal()
    if len(ops) == 1:
        return lead
    trail = _kron_diag(*ops[1:])
    d = lead.unsqueeze(-2) * trail.unsqueeze(-1)
    return d.mT.reshape(*d.shape[:-2], -1)


def _kron_matmul(ops, kp_shape, rhs):
    """Per factor: view [n_i, -1], multiply, transposing reshape (reference :34-45) -- ATen fallback path."""
    out_shape = _matmul_broadcast_shape(kp_shape, rhs.shape)
    batch = out_shape[:-2]
    res = rhs.expand(*batch, *rhs.shape[-2:])
    ncols = rhs.size(-1)
    for op in ops:
        res = res.reshape(*batch, op.size(-1), -1)
        f = op._matmul(res)
        res = f.view(*batch, op.size(-2), -1, ncols).transpose(-3, -2).reshape(*batch, -1, ncols)
    return res


def _dense_kron(ts):
    res = ts[0]
    for nxt in ts[1:]:
        res = (res.unsqueeze(-1).unsqueeze(-3) * nxt.unsqueeze(-2).unsqueeze(-4)).reshape(
            *torch.broadcast_shapes(res.shape[:-2], nxt.shape[:-2]), res.shape[-2] * nxt.shape[-2],
            res.shape[-1] * nxt.shape[-1])
    return res


def _group_pullback(dG: Tensor, ts):
    """Gradients of the factors of G = T_1 (x) .. (x) T_p from dG: dT_i[a, b] = sum over the other factors' indices of
    dG[(.., a, ..), (.., b, ..)] prod_{j != i} T_j[r_j, c_j]."""
    if len(ts) == 1:
        return [dG]
    p = len(ts)
    sizes = [t.shape[-1] for t in ts]
    dG = dG.reshape(*dG.shape[:-2], *sizes, *sizes)
    rows, cols = "abcdefgh"[:p], "ijklmnop"[:p]
    out = []
    for i in range(p):
        others = ",".join(f"...{rows[j]}{cols[j]}" for j in range(p) if j != i)
        expr = f"...{rows}{cols},{others}->...{rows[i]}{cols[i]}"
        out.append(torch.einsum(expr, dG, *[ts[j] for j in range(p) if j != i]))
    return out


class KroneckerProductLinearOperator(LinearOperator):
    def __init__(self, *linear_ops):
        try:
            linear_ops = tuple(to_linear_operator(op) for op in linear_ops)
        except TypeError:
            raise RuntimeError("KroneckerProductLinearOperator is intended to wrap lazy tensors.")
        try:
            batch = torch.broadcast_shapes(*(op.batch_shape for op in linear_ops))
        except RuntimeError:
            raise RuntimeError(
                "Batch shapes of LinearOperators "
                f"({', '.join([str(tuple(op.shape)) for op in linear_ops])}) "
                "are incompatible for a Kronecker product."
            )
        if len(batch):
            linear_ops = tuple(op._expand_batch(batch) if op.batch_shape != batch else op for op in linear_ops)
        super().__init__(*linear_ops)
        self.linear_ops = linear_ops

    # ---- lowering: the kernels take TWO dense factors.  A product of more factors is regrouped as
    #      (K_1 (x) .. (x) K_j) (x) (K_j+1 (x) .. (x) K_m) with the split that balances the two sides, the groups formed
    #      densely (they are small: the whole point of the structure is n_i << N); gradients are pulled back to the
    #      individual factors by contracting the group's gradient with the other factors of the group.
    _kMaxGroup = 2048  # largest side of a regrouped factor (B x n^2 floats are materialised)

    def _two_groups(self):
        """(A, B, j): dense group tensors and the split index, or None when the product does not lower."""
        ops = self.linear_ops
        if len(ops) < 2 or not all(isinstance(op, DenseLinearOperator) for op in ops):
            return None
        ts = [op.tensor for op in ops]
        if not all(t.is_cuda and t.dtype == torch.float32 and t.shape[-1] == t.shape[-2] for t in ts):
            return None
        if len(ts) == 2:
            return ts[0], ts[1], 1
        sizes = [t.shape[-1] for t in ts]
        best = None
        for j in range(1, len(ts)):
            na, nb = math.prod(sizes[:j]), math.prod(sizes[j:])
            if best is None or max(na, nb) < best[0]:
                best = (max(na, nb), j)
        if best[0] > self._kMaxGroup:
            return None
        j = best[1]
        # the regrouped dense factors are memoised per operator object, keyed on the factor tensors' storage AND version
        # counters: an in-place update of a factor (optimizer step on a reused operator) rebuilds the groups instead of
        # silently multiplying with stale ones
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) if not t.is_inference() else None for t in ts)
        cache = getattr(self, "_groups_cache", None)
        if cache is None or None in key or cache[0] != key:
            with torch.no_grad():
                cache = (key, (_dense_kron(ts[:j]), _dense_kron(ts[j:]), j))
            self._groups_cache = cache
        return cache[1]

    def _kernel_descriptor(self, batch_shape=None):
        groups = self._two_groups()
        if groups is None:
            return None
        k1, k2, _ = groups
        bs = torch.Size(batch_shape) if batch_shape is not None else self.batch_shape
        k1 = k1.expand(*bs, *k1.shape[-2:])
        k2 = k2.expand(*bs, *k2.shape[-2:])
        return K.kron_diag_descriptor(k1, k2, None)

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):
        """(dK1, dK2) = (sum_d U_d K2 V_d^T, sum_d U_d^T K1 V_d): the reference's generic autograd version
        (_linear_operator.py:336-393) applied to the Kronecker matvec (:34-45); two dense factors on this path."""
        groups = self._two_groups()
        if groups is None:
            return super()._bilinear_derivative(left_vecs, right_vecs)
        k1, k2, j = groups
        d1, d2 = K.bilinear_kron(k1, k2, left_vecs, right_vecs)
        ts = [op.tensor for op in self.linear_ops]
        grads = _group_pullback(d1, ts[:j]) + _group_pullback(d2, ts[j:])
        return tuple(g if tuple(g.shape) == tuple(t.shape) else g.sum_to_size(*t.shape) for g, t in zip(grads, ts))

    def __add__(self, other):  # reference :98-114
        from .diag_linear_operator import ConstantDiagLinearOperator, DiagLinearOperator
        from .kronecker_product_added_diag_linear_operator import KroneckerProductAddedDiagLinearOperator

        if isinstance(other, (KroneckerProductDiagLinearOperator, ConstantDiagLinearOperator)):
            return KroneckerProductAddedDiagLinearOperator(self, other)
        if isinstance(other, DiagLinearOperator):
            return self.add_diagonal(other._diagonal())
        return super().__add__(other)

    def add_diagonal(self, diag: Tensor):  # reference :116-145
        from .diag_linear_operator import ConstantDiagLinearOperator, DiagLinearOperator
        from .kronecker_product_added_diag_linear_operator import KroneckerProductAddedDiagLinearOperator

        if not self.is_square:
            raise RuntimeError("add_diag only defined for square matrices")
        diag_shape = diag.shape
        if len(diag_shape) == 0:  # scalar tensor = constant diagonal
            diag_tensor = ConstantDiagLinearOperator(diag.unsqueeze(-1), diag_shape=self.shape[-1])
        elif diag_shape[-1] == 1:
            diag_tensor = ConstantDiagLinearOperator(diag, diag_shape=self.shape[-1])
        else:
            try:
                expanded_diag = diag.expand(self.shape[:-1])
            except RuntimeError:
                raise RuntimeError(
                    "add_diag for LinearOperator of size {} received invalid diagonal of size {}.".format(
                        self.shape, diag_shape
                    )
                )
            diag_tensor = DiagLinearOperator(expanded_diag)
        return KroneckerProductAddedDiagLinearOperator(self, diag_tensor)

    def diagonalization(self, method=None):  # reference :147-152
        return super().diagonalization(method="symeig" if method is None else method)

    def _symeig(self, eigenvectors: bool = False, return_evals_as_lazy: bool = False, symeig_dtype_evals: bool = False):
        """Per-factor eigendecompositions (reference :338-360): evals = Kronecker product of the factors' eigenvalues
        [*batch, N], evecs = Kronecker product of the factors' eigenvector matrices.  `symeig_dtype_evals` keeps the
        eigenvalues in `settings._linalg_dtype_symeig` (the closed-form solve shifts and inverts them there, like
        the reference's fp64 solve, kronecker_product_added_diag_linear_operator.py:147-161)."""
        evals, evecs = None, []
        for op in self.linear_ops:
            dense = op.to_dense()
            ev, q = torch.linalg.eigh(dense.to(dtype=settings._linalg_dtype_symeig.value()))
            ev = ev.clamp_min(0.0)
            if not symeig_dtype_evals:
                ev = ev.to(dtype=dense.dtype)
            evals = ev if evals is None else (evals.unsqueeze(-1) * ev.unsqueeze(-2)).reshape(*ev.shape[:-1], -1)
            evecs.append(DenseLinearOperator(q.to(dtype=dense.dtype)))
        return evals, (KroneckerProductLinearOperator(*evecs) if eigenvectors else None)

    def _diagonal(self) -> Tensor:
        return _kron_diag(*self.linear_ops)

    def _expand_batch(self, batch_shape):
        return self.__class__(*[op._expand_batch(batch_shape) for op in self.linear_ops])

    def _get_indices(self, row_index, col_index, *batch_indices) -> Tensor:  # reference :198-216
        row_factor, col_factor = self.size(-2), self.size(-1)
        res = None
        for op in self.linear_ops:
            nr, nc = op.size(-2), op.size(-1)
            row_factor //= nr
            col_factor //= nc
            sub = op._get_indices(
                torch.div(row_index, row_factor, rounding_mode="floor").fmod(nr),
                torch.div(col_index, col_factor, rounding_mode="floor").fmod(nc),
                *batch_indices,
            )
            res = sub if res is None else (sub * res)
        return res

    def _matmul(self, rhs: Tensor) -> Tensor:  # reference :272-284
        is_vec = rhs.ndimension() == 1
        if is_vec:
            rhs = rhs.unsqueeze(-1)
        desc = None
        if rhs.is_cuda and rhs.dtype == torch.float32:
            desc = self._kernel_descriptor(torch.broadcast_shapes(self.batch_shape, rhs.shape[:-2]))
        if desc is not None:
            res = K.matvec(desc, rhs.expand(*desc.batch_shape, *rhs.shape[-2:]))
        else:
            res = _kron_matmul(self.linear_ops, self.shape, rhs.contiguous())
        return res.squeeze(-1) if is_vec else res

    def _t_matmul(self, rhs):
        return self.mT._matmul(rhs)

    def _size(self) -> torch.Size:
        rows = 1
        cols = 1
        for op in self.linear_ops:
            rows *= op.size(-2)
            cols *= op.size(-1)
        return torch.Size((*self.linear_ops[0].batch_shape, rows, cols))

    def _transpose_nonbatch(self):
        return self.__class__(*(op._transpose_nonbatch() for op in self.linear_ops))

    def to_dense(self) -> Tensor:
        res = self.linear_ops[0].to_dense()
        for op in self.linear_ops[1:]:
            nxt = op.to_dense()
            res = (res.unsqueeze(-1).unsqueeze(-3) * nxt.unsqueeze(-2).unsqueeze(-4)).reshape(
                *res.shape[:-2], res.shape[-2] * nxt.shape[-2], res.shape[-1] * nxt.shape[-1])
        return res


class KroneckerProductDiagLinearOperator(DiagLinearOperator):
    """D_1 (x) .. (x) D_P of diagonal operators (reference :436-541): a diagonal whose N = prod n_i entries are never
    stored as leaves -- the representation is the factors' own tensors, so gradients reach the factors."""

    def __init__(self, *linear_ops):
        if not all(isinstance(op, DiagLinearOperator) for op in linear_ops):
            raise RuntimeError("Components of KroneckerProductDiagLinearOperator must be DiagLinearOperator.")
        LinearOperator.__init__(self, *linear_ops)
        self.linear_ops = linear_ops

    @property
    def _diag(self) -> Tensor:
        return _kron_diag(*self.linear_ops)

    def _size(self) -> torch.Size:
        shapes = [op._diag.shape for op in self.linear_ops]
        n = math.prod(sh[-1] for sh in shapes)
        return torch.Size((*torch.broadcast_shapes(*(sh[:-1] for sh in shapes)), n, n))

    def _expand_batch(self, batch_shape):
        return self.__class__(*[op._expand_batch(batch_shape) for op in self.linear_ops])

    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):
        """sum_cols u o v is the gradient of the full diagonal [*batch, N]; factor i receives its contraction with the
        other factors' diagonals, handed to the factor's own rule (full / constant diagonal)."""
        g = K.bilinear_diag(left_vecs, right_vecs, self.batch_shape)
        diags = [op._diag for op in self.linear_ops]
        p = len(diags)
        g = g.reshape(*g.shape[:-1], *(dg.shape[-1] for dg in diags))
        idx = "abcdefgh"[:p]
        out = []
        for i, op in enumerate(self.linear_ops):
            if p == 1:
                gi = g
            else:
                others = ",".join(f"...{idx[j]}" for j in range(p) if j != i)
                gi = torch.einsum(f"...{idx},{others}->...{idx[i]}", g, *[diags[j] for j in range(p) if j != i])
            leaf = op.representation()[0]
            if leaf.shape[-1] == 1 and op._diag.shape[-1] != 1:  # constant factor: d(sigma) = sum of its diagonal's
                gi = gi.sum(-1, keepdim=True)
            out.append(gi if tuple(gi.shape) == tuple(leaf.shape) else gi.sum_to_size(*leaf.shape))
        return tuple(out)

    def abs(self):
        return self.__class__(*[op.abs() for op in self.linear_ops])

    def sqrt(self):
        return self.__class__(*[op.sqrt() for op in self.linear_ops])

    def inverse(self):
        return self.__class__(*[op.inverse() for op in self.linear_ops])

    def exp(self):
        raise NotImplementedError(f"torch.exp({self.__class__.__name__}) is not implemented.")

    def log(self):
        raise NotImplementedError(f"torch.log({self.__class__.__name__}) is not implemented.")


__all__ = ["KroneckerProductLinearOperator", "KroneckerProductDiagLinearOperator"]
