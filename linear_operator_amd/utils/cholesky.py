"""psd_safe_cholesky: cholesky_ex with escalating jitter (reference: linear_operator/utils/cholesky.py:13-74).
Plumbing for the N <= max_cholesky_size dispatch branch; runs on whatever device the tensor lives on through
ATen (this is the reference's own exact path, not part of the HIP hot path)."""
from __future__ import annotations

import warnings

import torch

from .. import settings
from .errors import NanError, NotPSDError
from .warnings import NumericalWarning


def psd_safe_cholesky(A, upper=False, jitter=None, max_tries=None):
    L, info = torch.linalg.cholesky_ex(A)
    if not torch.any(info):
        return L.mT if upper else L
    if torch.isnan(A).any():
        raise NanError(f"cholesky_cpu: {int(torch.isnan(A).sum())} of {A.numel()} elements of the {tuple(A.shape)} tensor are NaN.")
    if jitter is None:
        jitter = settings.cholesky_jitter.value(A.dtype)
    if max_tries is None:
        max_tries = settings.cholesky_max_tries.value()
    Aprime = A.clone()
    prev = 0.0
    for i in range(max_tries):
        new = jitter * (10 ** i)
        diag_add = ((info > 0) * (new - prev)).unsqueeze(-1).expand(*Aprime.shape[:-1])
        Aprime.diagonal(dim1=-1, dim2=-2).add_(diag_add)
        prev = new
        warnings.warn(f"A not p.d., added jitter of {new:.1e} to the diagonal", NumericalWarning)
        L, info = torch.linalg.cholesky_ex(Aprime)
        if not torch.any(info):
            return L.mT if upper else L
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {new:.1e}.")
