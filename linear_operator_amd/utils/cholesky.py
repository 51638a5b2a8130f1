"""psd_safe_cholesky: cholesky_ex with escalating jitter (reference: linear_operator/utils/cholesky.py:13-74).
Plumbing for the N <= max_cholesky_size dispatch branch; runs on whatever device the tensor lives on through
ATen (this is the reference's own exact path, not part of the HIP hot path)."""
from __future__ import annotations

import warnings

import torch

from .. import settings
from .errors import NanError, NotPSDError
from .warnings import NumericalWarning


def _cholesky_ex(A):
    """torch.linalg.cholesky_ex.  On this ROCm stack (torch 2.10 + rocm 7.0, gfx950) the BATCHED float32 factorisation
    of matrices with 256 < n < 384 rows dies with `HIP error: unspecified launch failure` (n <= 256 and n >= 384 are
    fine, so are one matrix and float64; tools/probe/chol_batched.py) and takes the context with it.  Those shapes are
    factorised as blockdiag(A, I) of 384 rows -- chol(blockdiag(A, I)) = blockdiag(chol(A), I), differentiable -- and
    the leading block is returned."""
    n = A.shape[-1]
    if A.is_cuda and A.dtype == torch.float32 and A.dim() > 2 and A.shape[:-2].numel() > 1 and 256 < n < 384:
        pad = 384 - n
        Ap = torch.nn.functional.pad(A, (0, pad, 0, pad))
        eye_tail = torch.zeros(384, dtype=A.dtype, device=A.device)
        eye_tail[n:] = 1.0
        Lp, info = torch.linalg.cholesky_ex(Ap + torch.diag_embed(eye_tail))
        return Lp[..., :n, :n], info
    return torch.linalg.cholesky_ex(A)


def cholesky_solve(rhs, factor, upper=False):
    """torch.cholesky_solve.  Same stack, second hole: the BATCHED solve of ONE right-hand-side column against factors of
    more than 512 rows (float32 and float64; two or more columns, one matrix, and n <= 512 are fine --
    tools/probe/chol_solve_batched.py) ends in `unspecified launch failure`.  Such a column is solved twice side by side
    and the first copy returned."""
    if (rhs.is_cuda and rhs.dim() >= 2 and rhs.shape[-1] == 1 and factor.shape[-1] > 512
            and torch.broadcast_shapes(rhs.shape[:-2], factor.shape[:-2]).numel() > 1):
        return torch.cholesky_solve(rhs.expand(*rhs.shape[:-1], 2).contiguous(), factor, upper=upper)[..., :1]
    return torch.cholesky_solve(rhs, factor, upper=upper)


def psd_safe_cholesky(A, upper=False, jitter=None, max_tries=None):
    L, info = _cholesky_ex(A)
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
        L, info = _cholesky_ex(Aprime)
        if not torch.any(info):
            return L.mT if upper else L
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {new:.1e}.")
