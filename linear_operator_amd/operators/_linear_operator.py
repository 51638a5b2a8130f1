"""LinearOperator base class -- the part of the reference's API surface the solve / logdet path needs
(reference: linear_operator/operators/_linear_operator.py, 3039 lines; SURVEY.md section 8(b) lists the kept
signatures).  Operators stay thin Python objects holding tensors.  What is new: `_kernel_descriptor()` -- the
lowering of an operator tree to the descriptor liblo_amd's kernels consume -- which lets `_solve`,
`pivoted_cholesky`, `_matmul` ... hand whole loops to the device instead of calling closures per iteration.

Kept seams: operator protocol (_matmul/_size/_transpose_nonbatch), `_solve` / `_preconditioner` /
`_solve_preconditioner` / `_probe_vectors_and_norms` overrides, `__torch_function__` dispatch by method NAME
(so subclass overrides win, :3006-3009), `utils.linear_cg` looked up at call time (:796).
"""
from __future__ import annotations

import itertools
import numbers
from collections import OrderedDict
from typing import Callable, Optional

import torch
from torch import Tensor

from .. import settings, utils
from ..utils.broadcasting import _matmul_broadcast_shape
from ..utils.cholesky import cholesky_solve
from .linear_operator_representation_tree import LinearOperatorRepresentationTree

_HANDLED_FUNCTIONS = {}
_HANDLED_SECOND_ARG_FUNCTIONS = {}


def _implements(torch_function: Callable) -> Callable:
    """Register `method` as the override of `torch_function` (stored by name, reference :61-74)."""

    def decorator(func):
        _HANDLED_FUNCTIONS[torch_function] = func.__name__
        return func

    return decorator


def _implements_second_arg(torch_function: Callable) -> Callable:
    """Override for torch functions whose SECOND argument is the operator (reference :77-95)."""

    def decorator(func):
        _HANDLED_SECOND_ARG_FUNCTIONS[torch_function] = func.__name__
        return func

    return decorator


def _implements_symmetric(torch_function: Callable) -> Callable:
    def decorator(func):
        _HANDLED_FUNCTIONS[torch_function] = func.__name__
        _HANDLED_SECOND_ARG_FUNCTIONS[torch_function] = func.__name__
        return func

    return decorator


class LinearOperator(object):
    """A (batch of) matrices of size (... x M x N) represented by its action `_matmul`."""

    def _check_args(self, *args, **kwargs) -> Optional[str]:
        return None

    def __init__(self, *args, **kwargs):
        if settings.debug.on():
            err = self._check_args(*args, **kwargs)
            if err is not None:
                raise ValueError(err)
        self._args = args
        self._differentiable_kwargs = OrderedDict()
        self._nondifferentiable_kwargs = dict()
        for name, val in sorted(kwargs.items()):  # sorted: deterministic flattening (reference :160-166)
            if torch.is_tensor(val) or isinstance(val, LinearOperator):
                self._differentiable_kwargs[name] = val
            else:
                self._nondifferentiable_kwargs[name] = val

    # ------------------------------------------------------------------ operator protocol (reference :169-221)
    def _matmul(self, rhs: Tensor) -> Tensor:
        raise NotImplementedError("The class {} requires a _matmul function!".format(self.__class__.__name__))

    def _size(self) -> torch.Size:
        raise NotImplementedError("The class {} requires a _size function!".format(self.__class__.__name__))

    def _transpose_nonbatch(self) -> "LinearOperator":
        raise NotImplementedError(
            "The class {} requires a _transpose_nonbatch function!".format(self.__class__.__name__)
        )

    # ------------------------------------------------------------------ lowering hook (new)
    def _kernel_descriptor(self, batch_shape=None):
        """OperatorDescriptor for liblo_amd, or None if this tree has no native kernel (-> closure path).
        `batch_shape`: batch the descriptor must be expanded to (defaults to the operator's own)."""
        return None

    # ------------------------------------------------------------------ optional overrides
    def _diagonal(self) -> Tensor:
        raise NotImplementedError(f"{self.__class__.__name__} does not define _diagonal")

    def _approx_diagonal(self) -> Tensor:  # reference :483-497
        return self._diagonal()

    def _expand_batch(self, batch_shape) -> "LinearOperator":
        raise NotImplementedError(f"{self.__class__.__name__} does not define _expand_batch")

    def _t_matmul(self, rhs: Tensor) -> Tensor:
        return self.mT._matmul(rhs)

    def _preconditioner(self):  # reference :618-627
        """(closure P^-1(.), LinearOperator P, log|P|) or (None, None, None)."""
        return None, None, None

    def _probe_vectors_and_norms(self):  # reference :629-633 -- hook for fixed probes
        return None, None

    def _solve_preconditioner(self):  # reference :805-846 (default_preconditioner beta feature: out of scope)
        base_precond, _, _ = self._preconditioner()
        return base_precond

    def _solve(self, rhs: Tensor, preconditioner: Optional[Callable] = None, num_tridiag: Optional[int] = 0):
        """reference :781-803.  `utils.linear_cg` is resolved through the module at call time (seam)."""
        return utils.linear_cg(
            self._matmul,
            rhs,
            n_tridiag=num_tridiag,
            max_iter=settings.max_cg_iterations.value(),
            max_tridiag_iter=settings.max_lanczos_quadrature_iterations.value(),
            preconditioner=preconditioner,
        )

    def _cholesky_solve(self, rhs, upper: bool = False):
        raise NotImplementedError(f"_cholesky_solve not implemented for {self.__class__.__name__}")

    # ------------------------------------------------------------------ shape / dtype
    @property
    def shape(self) -> torch.Size:
        # the size of an operator never changes after construction (its tensors may be updated in place, never
        # reshaped): computed once per object -- `_size` of a sum is a torch.broadcast_shapes call (~15 us) and the solve
        # path asks for the shape a dozen times
        s = self.__dict__.get("_shape_memo")
        if s is None:
            s = self.__dict__["_shape_memo"] = self._size()
        return s

    def size(self, dim: Optional[int] = None):
        s = self.shape
        return s if dim is None else s[dim]

    def dim(self) -> int:
        return len(self.shape)

    ndimension = dim

    @property
    def batch_shape(self) -> torch.Size:
        return self.shape[:-2]

    @property
    def batch_dim(self) -> int:
        return len(self.batch_shape)

    @property
    def matrix_shape(self) -> torch.Size:
        return torch.Size(self.shape[-2:])

    @property
    def is_square(self) -> bool:
        return self.matrix_shape[0] == self.matrix_shape[1]

    def numel(self) -> int:
        return self.shape.numel()

    def _leaf_tensors(self):
        for arg in itertools.chain(self._args, self._differentiable_kwargs.values()):
            if torch.is_tensor(arg):
                yield arg
            elif isinstance(arg, LinearOperator):
                yield from arg._leaf_tensors()

    @property
    def dtype(self) -> torch.dtype:
        for t in self._leaf_tensors():
            return t.dtype
        return torch.get_default_dtype()

    @property
    def device(self) -> torch.device:
        for t in self._leaf_tensors():
            return t.device
        return torch.device("cpu")

    @property
    def requires_grad(self) -> bool:
        return any(t.requires_grad for t in self._leaf_tensors())

    def detach(self) -> "LinearOperator":
        def conv(a):
            return a.detach() if hasattr(a, "detach") else a

        return self.__class__(*[conv(a) for a in self._args],
                              **{k: conv(v) for k, v in self._differentiable_kwargs.items()},
                              **self._nondifferentiable_kwargs)

    def to(self, *args, **kwargs) -> "LinearOperator":
        def conv(a):
            return a.to(*args, **kwargs) if hasattr(a, "to") else a

        return self.__class__(*[conv(a) for a in self._args],
                              **{k: conv(v) for k, v in self._differentiable_kwargs.items()},
                              **self._nondifferentiable_kwargs)

    def cuda(self, device_id=None) -> "LinearOperator":
        return self.to(torch.device("cuda", device_id) if device_id is not None else "cuda")

    def cpu(self) -> "LinearOperator":
        return self.to("cpu")

    # ------------------------------------------------------------------ representation (reference :2076-2101)
    def _bilinear_derivative(self, left_vecs: Tensor, right_vecs: Tensor):
        """d/d(theta) sum_d u_d^T K(theta) v_d for the tensors theta that represent this operator (reference
        :336-393).  The reference's generic version differentiates `_matmul` with autograd; the matvecs here are HIP
        kernels, so every operator class on the path supplies the closed form (csrc/lo_bilinear.hip)."""
        raise NotImplementedError(
            f"{self.__class__.__name__}._bilinear_derivative is not implemented in linear_operator_amd: gradients flow "
            "through Dense, Diag, ConstantDiag, Root / LowRankRoot and Sum / AddedDiag operators (SURVEY 8(f) rank 1)"
        )

    def representation(self):
        rep = []
        for arg in itertools.chain(self._args, self._differentiable_kwargs.values()):
            if torch.is_tensor(arg):
                rep.append(arg)
            elif hasattr(arg, "representation") and callable(arg.representation):
                rep += list(arg.representation())
            else:
                raise RuntimeError("Representation of a LinearOperator should consist only of Tensors")
        return tuple(rep)

    def representation_tree(self) -> LinearOperatorRepresentationTree:
        return LinearOperatorRepresentationTree(self)

    def evaluate_kernel(self) -> "LinearOperator":
        return self

    # ------------------------------------------------------------------ dense evaluation helpers
    def to_dense(self) -> Tensor:
        n = self.size(-1)
        eye = torch.eye(n, dtype=self.dtype, device=self.device).expand(*self.batch_shape, n, n).contiguous()
        return self._matmul(eye)

    # ------------------------------------------------------------------ element / row access (reference :395-461, :2829-2925)
    def _get_indices(self, row_index: Tensor, col_index: Tensor, *batch_indices: Tensor) -> Tensor:
        """Elements K[batch_indices..., row_index, col_index] for broadcastable tensor indices, one per dimension.
        Dense / Diag / Root / Kronecker / Sum override this with closed forms (as in the reference); the default for
        an opaque operator fetches the distinct rows with `_t_matmul` on one-hot columns and gathers from them."""
        final_shape = torch.broadcast_shapes(*(i.shape for i in batch_indices), row_index.shape, col_index.shape)
        row_index = row_index.expand(final_shape).reshape(-1)
        col_index = col_index.expand(final_shape).reshape(-1)
        bidx = [i.expand(final_shape).reshape(-1) for i in batch_indices]
        bshape = self.batch_shape
        flat_b = torch.zeros_like(row_index)
        for i, size in zip(bidx, bshape):
            flat_b = flat_b * size + i
        n = self.size(-2)
        key = flat_b * n + row_index  # one entry per requested (member, row)
        uniq, inverse = torch.unique(key, return_inverse=True)
        ub, ur = uniq // n, uniq % n
        nb = max(1, bshape.numel())
        # rows per member, padded to the largest count: column j of member b selects its j-th requested row
        counts = torch.bincount(ub, minlength=nb)
        m = int(counts.max().item())
        start = torch.cumsum(counts, 0) - counts
        slot = torch.arange(uniq.numel(), device=uniq.device) - start[ub]
        onehot = torch.zeros(nb, n, m, dtype=self.dtype, device=self.device)
        onehot[ub, ur, slot] = 1.0
        rows = self._t_matmul(onehot.reshape(*bshape, n, m)).reshape(nb, self.size(-1), m)  # K^T e = rows of K
        res = rows[ub[inverse], col_index, slot[inverse]]
        return res.reshape(final_shape)

    def _get_rows(self, row_index: Tensor) -> Tensor:
        """K[..., row_index, :] for one row per batch member (row_index [*batch] int64) -> [*batch, N]: the access
        PivotedCholesky.forward makes per pivot (`apply_permutation(matrix, pi_m.unsqueeze(-1))`,
        _pivoted_cholesky.py:81 -> utils/permutation.py:64-88 -> __getitem__ with tensor indices on every dimension)."""
        if type(self)._get_indices is LinearOperator._get_indices:
            # opaque operator: row pi of K is (K^T e_pi)^T -- one product per pivot, no host synchronisation
            onehot = torch.zeros(*self.batch_shape, self.size(-2), 1, dtype=self.dtype, device=self.device)
            onehot.scatter_(-2, row_index.reshape(*self.batch_shape, 1, 1), 1.0)
            return self._t_matmul(onehot).squeeze(-1)
        left = row_index.unsqueeze(-1)
        batch_idx = []
        for i, size in enumerate(self.batch_shape):
            shape = [1] * (len(self.batch_shape) + 2)
            shape[i] = size
            batch_idx.append(torch.arange(size, device=self.device).view(*shape))
        right = torch.arange(self.size(-1), device=self.device)
        res = self.__getitem__((*batch_idx, left.unsqueeze(-1), right.unsqueeze(-2)))
        return to_dense(res).squeeze(-2)

    def _getitem(self, row_index, col_index, *batch_indices):
        """Sub-operator for int / slice (and lone tensor) indices.  Generic version: index the dense evaluation (the
        reference builds lazily indexed operators per class, :436-461; this path is not on the iterative hot path)."""
        from .dense_linear_operator import DenseLinearOperator

        return DenseLinearOperator(self.to_dense()[(*batch_indices, row_index, col_index)])

    def __getitem__(self, index):  # reference :2829-2925
        ndimension = self.dim()
        index = index if isinstance(index, tuple) else (index,)
        index = tuple(torch.tensor(idx) if isinstance(idx, list) else idx for idx in index)
        index = tuple(idx.item() if torch.is_tensor(idx) and not len(idx.shape) else idx for idx in index)
        ellipsis_locs = tuple(i for i, item in enumerate(index) if item is Ellipsis)
        if len(ellipsis_locs) > 1:
            raise RuntimeError(f"Cannot have multiple ellipsis in a __getitem__ call. Received index {index}.")
        noop = slice(None, None, None)
        if len(ellipsis_locs) == 1:
            loc = ellipsis_locs[0]
            fill = ndimension - (len(index) - 1)
            index = index[:loc] + tuple(noop for _ in range(fill)) + index[loc + 1:]
        index = index + tuple(noop for _ in range(ndimension - len(index)))
        *batch_indices, row_index, col_index = index
        batch_has_tensor = bool(batch_indices) and any(torch.is_tensor(i) for i in batch_indices)
        row_tensor, col_tensor = torch.is_tensor(row_index), torch.is_tensor(col_index)
        absorbed = (batch_has_tensor and (row_tensor or col_tensor)) or (not batch_has_tensor and row_tensor and col_tensor)
        squeeze_row = squeeze_col = False
        if isinstance(row_index, int):
            row_index, squeeze_row = slice(row_index, row_index + 1, None), True
        if isinstance(col_index, int):
            col_index, squeeze_col = slice(col_index, col_index + 1, None), True
        if absorbed:
            orig = [*batch_indices, row_index, col_index]
            tshape = torch.broadcast_shapes(*[i.shape for i in orig if torch.is_tensor(i)])
            flat = []
            for dim, i in enumerate(orig):
                if torch.is_tensor(i):
                    flat.append(i.expand(tshape).reshape(-1))
                elif isinstance(i, int):
                    flat.append(torch.full((tshape.numel(),), i, dtype=torch.long, device=self.device))
                else:
                    raise NotImplementedError("mixing slices with tensor indices that absorb the matrix dimensions")
            *nb, nr, nc = flat
            res = self._get_indices(nr, nc, *nb)
            res = res.view(*tshape)
        else:
            res = self._getitem(row_index, col_index, *batch_indices)
        if squeeze_row or squeeze_col or absorbed:
            res = to_dense(res)
        if squeeze_row:
            res = res.squeeze(-2)
        if squeeze_col:
            res = res.squeeze(-1)
        return res

    def diagonal(self, offset: int = 0, dim1: int = -2, dim2: int = -1) -> Tensor:
        if not (offset == 0 and dim1 in (-2, self.dim() - 2) and dim2 in (-1, self.dim() - 1)):
            raise NotImplementedError("LinearOperator.diagonal only computes the main diagonal of the last two dims")
        if not self.is_square:
            raise RuntimeError("LinearOperator#diagonal is only defined for square matrices")
        return self._diagonal()

    def cholesky(self, upper: bool = False):
        """Dense Cholesky factor for the N <= max_cholesky_size branch (reference :1211-1225 -> _cholesky)."""
        from .dense_linear_operator import DenseLinearOperator
        from ..utils.cholesky import psd_safe_cholesky

        L = psd_safe_cholesky(self.to_dense(), upper=upper)
        return _TriangularFactor(L, upper=upper)

    # ------------------------------------------------------------------ transpose
    def transpose(self, dim1: int, dim2: int) -> "LinearOperator":
        nd = self.dim()
        dim1, dim2 = dim1 % nd, dim2 % nd
        if {dim1, dim2} == {nd - 2, nd - 1}:
            return self._transpose_nonbatch()
        raise NotImplementedError("only the last two dimensions can be transposed on this path")

    @property
    def mT(self) -> "LinearOperator":
        return self.transpose(-1, -2)

    # ------------------------------------------------------------------ algebra
    @_implements(torch.matmul)
    def matmul(self, other):  # reference :1844-1866
        from ..functions._matmul import Matmul

        if isinstance(other, LinearOperator):
            raise NotImplementedError("operator @ operator (MatmulLinearOperator) is outside the solve/logdet path")
        _matmul_broadcast_shape(self.shape, other.shape)
        return Matmul.apply(self.representation_tree(), other, *self.representation())

    def __matmul__(self, other):
        return self.matmul(other)

    @_implements_second_arg(torch.matmul)
    def rmatmul(self, other):
        if other.ndim == 1:
            return self.mT.matmul(other)
        return self.mT.matmul(other.mT).mT

    def __rmatmul__(self, other):
        return self.rmatmul(other)

    def add_diagonal(self, diag: Tensor) -> "LinearOperator":  # reference :953-1001
        from .added_diag_linear_operator import AddedDiagLinearOperator
        from .diag_linear_operator import ConstantDiagLinearOperator, DiagLinearOperator

        if not self.is_square:
            raise RuntimeError("add_diagonal only defined for square matrices")
        diag_shape = diag.shape
        if len(diag_shape) and diag_shape[-1] != 1:
            try:
                expanded = diag.expand(self.shape[:-1])
            except RuntimeError:
                raise RuntimeError(
                    "add_diagonal for LinearOperator of size {} received invalid diagonal of size {}.".format(
                        self.shape, diag_shape
                    )
                )
            diag_op = DiagLinearOperator(expanded)
        else:
            try:
                expanded = diag.expand(*self.batch_shape, 1)
            except RuntimeError:
                raise RuntimeError(
                    "add_diagonal for LinearOperator of size {} received invalid diagonal of size {}.".format(
                        self.shape, diag_shape
                    )
                )
            diag_op = ConstantDiagLinearOperator(expanded, diag_shape=self.shape[-1])
        return AddedDiagLinearOperator(self, diag_op)

    def add_jitter(self, jitter_val: float = 1e-3) -> "LinearOperator":  # reference :1003-1017
        return self.add_diagonal(torch.tensor(jitter_val, dtype=self.dtype, device=self.device))

    @_implements_symmetric(torch.add)
    def add(self, other, alpha=None):
        return self + other if alpha is None else self + alpha * other

    def __add__(self, other):  # reference :2801-2827
        from .added_diag_linear_operator import AddedDiagLinearOperator
        from .dense_linear_operator import to_linear_operator
        from .diag_linear_operator import DiagLinearOperator
        from .sum_linear_operator import SumLinearOperator

        if isinstance(other, DiagLinearOperator):
            return AddedDiagLinearOperator(self, other)
        if isinstance(other, Tensor):
            other = to_linear_operator(other)
            shape = torch.broadcast_shapes(self.shape, other.shape)
            new_self = self if self.shape[:-2] == shape[:-2] else self._expand_batch(shape[:-2])
            new_other = other if other.shape[:-2] == shape[:-2] else other._expand_batch(shape[:-2])
            return SumLinearOperator(new_self, new_other)
        if isinstance(other, numbers.Number) and other == 0:
            return self
        return SumLinearOperator(self, other)

    def __radd__(self, other):
        return self + other

    # ------------------------------------------------------------------ solves / quadratic forms / logdet
    @_implements(torch.linalg.solve)
    def solve(self, right_tensor: Tensor, left_tensor: Optional[Tensor] = None) -> Tensor:  # reference :2324-2379
        from ..functions._solve import Solve

        if not self.is_square:
            raise RuntimeError(
                "solve only operates on (batches of) square (positive semi-definite) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        if self.dim() == 2 and right_tensor.dim() == 1:
            if self.shape[-1] != right_tensor.numel():
                raise RuntimeError(
                    "LinearOperator (size={}) cannot be multiplied with right-hand-side Tensor (size={}).".format(
                        self.shape, right_tensor.shape
                    )
                )
        if left_tensor is None:
            if not (torch.is_grad_enabled() and (right_tensor.requires_grad or self.requires_grad)):
                # nothing to differentiate: the solve itself, without the Function's representation / rebuild round
                # trip (~40 us of host time, a tenth of a resident solve of 64 members)
                from ..functions._solve import _solve

                if right_tensor.dim() == 1:
                    return _solve(self, right_tensor.unsqueeze(-1)).squeeze(-1)
                return _solve(self, right_tensor)
            return Solve.apply(self.representation_tree(), False, right_tensor, *self.representation())
        return Solve.apply(self.representation_tree(), True, left_tensor, right_tensor, *self.representation())

    def inv_quad(self, inv_quad_rhs: Tensor, reduce_inv_quad: bool = True) -> Tensor:  # reference :1637-1686
        from ..functions._inv_quad import InvQuad

        if not self.is_square:
            raise RuntimeError(
                "inv_quad only operates on (batches of) square (positive semi-definite) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        try:
            result_shape = _matmul_broadcast_shape(self.shape, inv_quad_rhs.shape)
        except RuntimeError:
            raise RuntimeError(
                "LinearOperator (size={}) cannot be multiplied with right-hand-side Tensor (size={}).".format(
                    self.shape, inv_quad_rhs.shape
                )
            )
        args = (inv_quad_rhs.expand(*result_shape[:-2], *inv_quad_rhs.shape[-2:]),) + self.representation()
        term = InvQuad.apply(self.representation_tree(), *args)
        return term.sum(-1) if reduce_inv_quad else term

    def inv_quad_logdet(self, inv_quad_rhs: Optional[Tensor] = None, logdet: bool = False,
                        reduce_inv_quad: bool = True):  # reference :1688-1804
        from ..functions._inv_quad_logdet import InvQuadLogdet
        from .identity_linear_operator import IdentityLinearOperator

        if settings.fast_computations.log_prob.off() or (self.size(-1) <= settings.max_cholesky_size.value()):
            return self.cholesky().inv_quad_logdet(inv_quad_rhs=inv_quad_rhs, logdet=logdet,
                                                   reduce_inv_quad=reduce_inv_quad)
        if not logdet:
            if inv_quad_rhs is None:
                raise RuntimeError("Either `inv_quad_rhs` or `logdet` must be specifed.")
            return self.inv_quad(inv_quad_rhs, reduce_inv_quad=reduce_inv_quad), torch.zeros(
                [], dtype=self.dtype, device=self.device
            )
        if not self.is_square:
            raise RuntimeError(
                "inv_quad_logdet only operates on (batches of) square (positive semi-definite) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        if inv_quad_rhs is not None:
            if self.dim() == 2 and inv_quad_rhs.dim() == 1:
                if self.shape[-1] != inv_quad_rhs.numel():
                    raise RuntimeError(
                        "LinearOperator (size={}) cannot be multiplied with right-hand-side Tensor (size={}).".format(
                            self.shape, inv_quad_rhs.shape
                        )
                    )
            elif self.dim() != inv_quad_rhs.dim():
                raise RuntimeError(
                    "LinearOperator (size={}) and right-hand-side Tensor (size={}) should have the same number "
                    "of dimensions.".format(self.shape, inv_quad_rhs.shape)
                )
            elif self.batch_shape != inv_quad_rhs.shape[:-2] or self.shape[-1] != inv_quad_rhs.shape[-2]:
                raise RuntimeError(
                    "LinearOperator (size={}) cannot be multiplied with right-hand-side Tensor (size={}).".format(
                        self.shape, inv_quad_rhs.shape
                    )
                )
        args = self.representation()
        if inv_quad_rhs is not None:
            args = [inv_quad_rhs] + list(args)
        preconditioner, precond_lt, logdet_p = self._preconditioner()
        if precond_lt is None:
            precond_lt = IdentityLinearOperator(diag_shape=self.size(-1), batch_shape=self.batch_shape,
                                                dtype=self.dtype, device=self.device)
            logdet_p = 0.0
        precond_args = precond_lt.representation()
        probe_vectors, probe_vector_norms = self._probe_vectors_and_norms()
        inv_quad_term, pinvk_logdet = InvQuadLogdet.apply(
            self.representation_tree(), precond_lt.representation_tree(), preconditioner, len(precond_args),
            (inv_quad_rhs is not None), probe_vectors, probe_vector_norms, *(list(args) + list(precond_args)),
        )
        logdet_term = pinvk_logdet + logdet_p
        if inv_quad_term.numel() and reduce_inv_quad:
            inv_quad_term = inv_quad_term.sum(-1)
        return inv_quad_term, logdet_term

    @_implements(torch.logdet)
    def logdet(self) -> Tensor:  # reference :1834-1842
        _, res = self.inv_quad_logdet(inv_quad_rhs=None, logdet=True)
        return res

    def pivoted_cholesky(self, rank: int, error_tol: Optional[float] = None, return_pivots: bool = False):
        """reference :1975-2008 -> functions/_pivoted_cholesky.py"""
        from ..functions._pivoted_cholesky import PivotedCholesky

        res, pivots = PivotedCholesky.apply(self.representation_tree(), rank, error_tol, *self.representation())
        return (res, pivots) if return_pivots else res

    # ------------------------------------------------------------------ root decompositions
    def _root_decomposition_size(self) -> int:  # reference :715-721
        return settings.max_root_decomposition_size.value()

    def _choose_root_method(self) -> str:  # reference :543-561 (no eig caches on this path)
        if self.size(-1) <= settings.max_cholesky_size.value() or settings.fast_computations.covar_root_decomposition.off():
            return "cholesky"
        return "lanczos"

    def _root_decomposition(self):  # reference :689-713
        from ..functions._root_decomposition import RootDecomposition

        res, _ = RootDecomposition.apply(self.representation_tree(), self._root_decomposition_size(), self.dtype,
                                         self.device, self.batch_shape, self.matrix_shape, True, False, None,
                                         *self.representation())
        return res

    def _root_inv_decomposition(self, initial_vectors=None, test_vectors=None):  # reference :723-763
        from ..functions._root_decomposition import RootDecomposition

        roots, inv_roots = RootDecomposition.apply(self.representation_tree(), self._root_decomposition_size(),
                                                   self.dtype, self.device, self.batch_shape, self.matrix_shape, True,
                                                   True, initial_vectors, *self.representation())
        if initial_vectors is not None and initial_vectors.size(-1) > 1:
            self._root_decomposition_cache = roots[0]
        else:
            self._root_decomposition_cache = roots
        return inv_roots

    def sqrt_inv_matmul(self, rhs: Tensor, lhs: Optional[Tensor] = None):
        """A^{-1/2} rhs, or (lhs A^{-1/2} rhs, diag(lhs A^-1 lhs^T)) when `lhs` is given, by contour integral
        quadrature over shifted MINRES solves (reference :2422-2466, functions/_sqrt_inv_matmul.py)."""
        from ..functions._sqrt_inv_matmul import SqrtInvMatmul

        squeeze = rhs.dim() == 1
        if squeeze:
            rhs = rhs.unsqueeze(-1)
        res, inv_quad_res = SqrtInvMatmul.apply(self.representation_tree(), rhs, lhs, *self.representation())
        if squeeze:
            res = res.squeeze(-1)
        return res if lhs is None else (res, inv_quad_res)

    def _symeig(self, eigenvectors: bool = False, return_evals_as_lazy: bool = False):
        """Dense symmetric eigendecomposition in `settings._linalg_dtype_symeig` (ATen plumbing; reference :878-901)."""
        from .dense_linear_operator import DenseLinearOperator

        if settings.verbose_linalg.on():
            settings.verbose_linalg.logger.debug(f"Running symeig on a matrix of size {self.shape}.")
        dtype = self.dtype
        evals, evecs = torch.linalg.eigh(self.to_dense().to(dtype=settings._linalg_dtype_symeig.value()))
        evals = evals.clamp_min(0.0).to(dtype=dtype)
        return evals, (DenseLinearOperator(evecs.to(dtype=dtype)) if eigenvectors else None)

    def diagonalization(self, method: Optional[str] = None):
        """(evals, evecs) of a (usually partial) diagonalization Q diag(lambda) Q^T ~= A (reference :1439-1482):
        "lanczos" (device Lanczos, functions/_diagonalization.py) or "symeig"."""
        from ..functions._diagonalization import Diagonalization
        from . import to_linear_operator

        if not self.is_square:
            raise RuntimeError(
                "diagonalization only operates on (batches of) square (symmetric) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        if method is None:
            method = "symeig" if self.size(-1) <= settings.max_cholesky_size.value() else "lanczos"
        if method == "lanczos":
            evals, evecs = Diagonalization.apply(
                self.representation_tree(), self.device, self.dtype, self.matrix_shape,
                self._root_decomposition_size(), self.batch_shape, *self.representation(),
            )
            return evals, to_linear_operator(evecs)
        if method == "symeig":
            return self._symeig(eigenvectors=True)
        raise RuntimeError(f"Unknown diagonalization method '{method}'")

    def root_decomposition(self, method: Optional[str] = None):
        """R with R R^T ~= A (reference :2158-2218).  Methods on this path: "lanczos" (device Lanczos + tridiagonal
        eigh), "cholesky" (dense factor, N <= max_cholesky_size), "pivoted_cholesky"."""
        from .root_linear_operator import RootLinearOperator

        if not self.is_square:
            raise RuntimeError(
                "root_decomposition only operates on (batches of) square (symmetric) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        if self.shape[-2:].numel() == 1:
            return RootLinearOperator(self.to_dense().sqrt())
        if method is None:
            method = self._choose_root_method()
        if method == "cholesky":
            return RootLinearOperator(self.cholesky().to_dense())
        if method == "pivoted_cholesky":
            return RootLinearOperator(self.pivoted_cholesky(rank=self._root_decomposition_size()))
        if method == "lanczos":
            return RootLinearOperator(self._root_decomposition())
        raise RuntimeError(f"Unknown root decomposition method '{method}'")

    def root_inv_decomposition(self, initial_vectors=None, test_vectors=None, method: Optional[str] = None):
        """R with R R^T ~= A^-1 (reference :2220-2312); "lanczos" and "cholesky" on this path."""
        from .root_linear_operator import RootLinearOperator

        if not self.is_square:
            raise RuntimeError(
                "root_inv_decomposition only operates on (batches of) square (symmetric) LinearOperators. "
                "Got a {} of size {}.".format(self.__class__.__name__, self.size())
            )
        if self.shape[-2:].numel() == 1:
            return RootLinearOperator(1 / self.to_dense().sqrt())
        if method is None:
            method = self._choose_root_method()
        if method == "cholesky":
            L = self.cholesky().to_dense()
            eye = torch.eye(L.shape[-2], device=L.device, dtype=L.dtype)
            return RootLinearOperator(torch.linalg.solve_triangular(L, eye, upper=False).mT)
        if method == "lanczos":
            if initial_vectors is not None:
                if self.dim() == 2 and initial_vectors.dim() == 1:
                    if self.shape[-1] != initial_vectors.numel():
                        raise RuntimeError(
                            "LinearOperator (size={}) cannot be multiplied with initial_vectors (size={}).".format(
                                self.shape, initial_vectors.shape
                            )
                        )
                elif self.dim() != initial_vectors.dim():
                    raise RuntimeError(
                        "LinearOperator (size={}) and initial_vectors (size={}) should have the same number "
                        "of dimensions.".format(self.shape, initial_vectors.shape)
                    )
                elif self.batch_shape != initial_vectors.shape[:-2] or self.shape[-1] != initial_vectors.shape[-2]:
                    raise RuntimeError(
                        "LinearOperator (size={}) cannot be multiplied with initial_vectors (size={}).".format(
                            self.shape, initial_vectors.shape
                        )
                    )
            inv_root = self._root_inv_decomposition(initial_vectors)
            if initial_vectors is not None and initial_vectors.size(-1) > 1:
                from ..utils.lanczos import _postprocess_lanczos_root_inv_decomp

                inv_root = _postprocess_lanczos_root_inv_decomp(self, inv_root, initial_vectors, test_vectors)
            return RootLinearOperator(inv_root)
        raise RuntimeError(f"Unknown root inv decomposition method '{method}'")

    def zero_mean_mvn_samples(self, num_samples: int) -> Tensor:  # reference :2746-2793
        if settings.ciq_samples.on():  # A^{1/2} z by contour integral quadrature (:2759-2776)
            from ..utils.contour_integral_quad import contour_integral_quad

            # (the samples ride in the column dimension of one shifted-MINRES run; the reference stacks them in a
            # leading batch dimension, which gives the same independent solves)
            base_samples = torch.randn(*self.batch_shape, self.size(-1), num_samples, dtype=self.dtype,
                                       device=self.device)
            solves, weights, _, _ = contour_integral_quad(
                self, base_samples, inverse=False, num_contour_quadrature=settings.num_contour_quadrature.value())
            return (solves * weights).sum(0).permute(-1, *range(self.dim() - 1)).contiguous()
        if self.size()[-2:] == torch.Size([1, 1]):
            covar_root = self.to_dense().sqrt()
        else:
            covar_root = self.root_decomposition().root
        base_samples = torch.randn(*self.batch_shape, covar_root.size(-1), num_samples, dtype=self.dtype,
                                   device=self.device)
        return covar_root.matmul(base_samples).permute(-1, *range(self.dim() - 1)).contiguous()

    # ------------------------------------------------------------------ torch dispatch (reference :2981-3009)
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if kwargs is None:
            kwargs = {}

        def unsupported():
            name = func.__name__.replace("linalg_", "linalg.")
            arg_classes = ", ".join(arg.__class__.__name__ for arg in args)
            kwarg_classes = ", ".join(f"{key}={val.__class__.__name__}" for key, val in kwargs.items())
            return NotImplementedError(f"torch.{name}({arg_classes}, {kwarg_classes}) is not implemented.")

        ok_types = all(issubclass(t, (torch.Tensor, LinearOperator)) for t in types)
        if not isinstance(args[0], cls):
            if func not in _HANDLED_SECOND_ARG_FUNCTIONS or not ok_types:
                raise unsupported()
            method = getattr(cls, _HANDLED_SECOND_ARG_FUNCTIONS[func])  # by NAME: subclass overrides win
            return method(args[1], args[0], *args[2:], **kwargs)
        if func not in _HANDLED_FUNCTIONS or not ok_types:
            raise unsupported()
        method = getattr(cls, _HANDLED_FUNCTIONS[func])
        return method(*args, **kwargs)

    def __repr__(self):
        return f"<{self.__class__.__name__} of size {tuple(self.shape)}>"


class _TriangularFactor:
    """Minimal stand-in for CholLinearOperator(TriangularLinearOperator(L)) on the N <= max_cholesky_size
    branch (reference: chol_linear_operator.py:121, triangular_linear_operator.py:72-91): exact solves and
    logdet through ATen.  Plumbing for cfg1, not part of the HIP hot path."""

    def __init__(self, factor: Tensor, upper: bool = False):
        self.factor = factor
        self.upper = upper

    def to_dense(self):
        return self.factor

    def _cholesky_solve(self, rhs, upper: bool = False):
        is_vec = rhs.dim() == 1
        if is_vec:
            rhs = rhs.unsqueeze(-1)
        res = cholesky_solve(rhs, self.factor, upper=self.upper)
        return res.squeeze(-1) if is_vec else res

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        inv_quad_term, logdet_term = None, None
        if inv_quad_rhs is not None:
            is_vec = inv_quad_rhs.dim() == 1
            r = inv_quad_rhs.unsqueeze(-1) if is_vec else inv_quad_rhs
            L = self.factor.mT if self.upper else self.factor
            half = torch.linalg.solve_triangular(L, r, upper=False)
            inv_quad_term = (half ** 2).sum(-2)
            if reduce_inv_quad:
                inv_quad_term = inv_quad_term.sum(-1)
        if logdet:
            logdet_term = self.factor.diagonal(dim1=-1, dim2=-2).pow(2).log().sum(-1)
        else:
            logdet_term = torch.zeros([], dtype=self.factor.dtype, device=self.factor.device)
        if inv_quad_term is None:
            inv_quad_term = torch.zeros([], dtype=self.factor.dtype, device=self.factor.device)
        return inv_quad_term, logdet_term


def to_dense(obj):
    if torch.is_tensor(obj):
        return obj
    if isinstance(obj, LinearOperator):
        return obj.to_dense()
    raise TypeError("object of class {} cannot be made into a Tensor".format(obj.__class__.__name__))


__all__ = ["LinearOperator", "to_dense"]

