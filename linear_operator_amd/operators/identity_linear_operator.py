"""IdentityLinearOperator: `precond_lt` when no preconditioner is built (reference:
operators/identity_linear_operator.py; used at _linear_operator.py:1774-1783, probes :262-266)."""
from __future__ import annotations

import torch
from torch import Tensor

from .diag_linear_operator import ConstantDiagLinearOperator


class IdentityLinearOperator(ConstantDiagLinearOperator):
    def __init__(self, diag_shape: int, batch_shape=torch.Size([]), dtype=torch.float, device=None):
        one = torch.tensor(1.0, dtype=dtype, device=device)
        from ._linear_operator import LinearOperator

        LinearOperator.__init__(self, diag_shape=diag_shape, batch_shape=batch_shape, dtype=dtype, device=device)
        self.diag_values = one.expand(*batch_shape, 1)
        self.diag_shape = diag_shape
        self._batch_shape = torch.Size(batch_shape)
        self._dtype = dtype
        self._device = device

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device if self._device is not None else torch.device("cpu")

    def _matmul(self, rhs: Tensor) -> Tensor:
        return rhs

    def _size(self) -> torch.Size:
        return torch.Size((*self._batch_shape, self.diag_shape, self.diag_shape))

    def zero_mean_mvn_samples(self, num_samples: int) -> Tensor:
        return torch.randn(num_samples, *self._batch_shape, self.diag_shape, dtype=self._dtype, device=self._device)


__all__ = ["IdentityLinearOperator"]
