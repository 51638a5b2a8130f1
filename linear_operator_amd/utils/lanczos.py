"""lanczos_tridiag / lanczos_tridiag_to_diag with the reference's signatures (linear_operator/utils/lanczos.py:9-189),
executed by csrc/lo_lanczos.hip and csrc/lo_eig.hip."""
from __future__ import annotations

import torch

from .. import settings
from .. import kernels as K
from .linear_cg import _lower_matmul_closure


def lanczos_tridiag(
    matmul_closure,
    max_iter,
    dtype,
    device,
    matrix_shape,
    batch_shape=torch.Size(),
    init_vecs=None,
    num_init_vecs=1,
    tol=1e-5,
):
    if not callable(matmul_closure) and not torch.is_tensor(matmul_closure):
        raise RuntimeError(
            "matmul_closure should be a function callable object that multiples a (Lazy)Tensor "
            "by a vector. Got a {} instead.".format(matmul_closure.__class__.__name__)
        )
    if init_vecs is None:  # lanczos.py:31-33: one random block shared (expanded) over the batch
        init_vecs = torch.randn(matrix_shape[-1], num_init_vecs, dtype=dtype, device=device)
        init_vecs = init_vecs.expand(*batch_shape, matrix_shape[-1], num_init_vecs)
    elif settings.debug.on():
        if dtype != init_vecs.dtype:
            raise RuntimeError(f"Supplied dtype {dtype} and init_vecs.dtype {init_vecs.dtype} do not agree!")
        if device != init_vecs.device:
            raise RuntimeError(f"Supplied device {device} and init_vecs.device {init_vecs.device} do not agree!")
        if torch.Size(batch_shape) != init_vecs.shape[:-2]:
            raise RuntimeError(f"batch_shape {batch_shape} and init_vecs.shape {init_vecs.shape} do not agree!")
        if matrix_shape[-1] != init_vecs.size(-2):
            raise RuntimeError(f"matrix_shape {matrix_shape} and init_vecs.shape {init_vecs.shape} do not agree!")
    num_iter = min(max_iter, matrix_shape[-1])  # :57
    if settings.verbose_linalg.on():
        settings.verbose_linalg.logger.debug(
            f"Running Lanczos on a {matrix_shape} matrix with a {init_vecs.shape} RHS for {num_iter} iterations."
        )
    if init_vecs.dtype == torch.float64:  # the reference is dtype-generic; fp64 runs csrc/lo_lanczos_f64.hip
        if torch.is_tensor(matmul_closure):
            return K.lanczos_tridiag_f64(matmul_closure, None, init_vecs.contiguous(), num_iter, tol=tol)
        return K.lanczos_tridiag_f64(None, None, init_vecs.contiguous(), num_iter, tol=tol,
                                     matvec_closure=matmul_closure)
    desc = _lower_matmul_closure(matmul_closure, init_vecs.shape[:-2])
    closure = None
    if desc is None:
        closure = matmul_closure.matmul if torch.is_tensor(matmul_closure) else matmul_closure
    return K.lanczos_tridiag(desc, init_vecs.contiguous(), num_iter, tol=tol, matvec_closure=closure)


def lanczos_tridiag_to_diag(t_mat, tridiagonal=True):
    """t_mat [P, *batch, k, k] -> (evals [P,*batch,k], evecs [P,*batch,k,k]); negative eigenvalues -> 1 with their
    eigenvector columns zeroed (lanczos.py:185-187).  On device for k <= 32 (the reference moves the matrices to
    the CPU in that case, :179-180); larger k uses torch.linalg.eigh on the device like the reference (:182).
    `tridiagonal=False`: the matrices are full symmetric (Diagonalization's jitter), dense eigensolver."""
    if settings.verbose_linalg.on():
        settings.verbose_linalg.logger.debug(f"Running symeig on a matrix of size {t_mat.shape}.")
    if tridiagonal and t_mat.size(-1) <= 32 and t_mat.is_cuda and t_mat.dtype == torch.float32:
        lead = t_mat.shape[:-2]
        t3 = t_mat.reshape(1, -1, *t_mat.shape[-2:])
        evals, evecs, _ = K.tridiag_eigh_slq(t3, 1, want_evecs=True, want_logdet=False)
        return evals.reshape(*lead, -1), evecs.reshape(*lead, *t_mat.shape[-2:])
    evals, evecs = torch.linalg.eigh(t_mat)
    mask = evals.ge(0)
    evecs = evecs * mask.type_as(evecs).unsqueeze(-2)
    evals = evals.masked_fill_(~mask, 1)
    return evals, evecs


def _postprocess_lanczos_root_inv_decomp(linear_op, inv_roots, initial_vectors, test_vectors):
    """Pick, among the inverse roots obtained from each initial vector (inv_roots [P, *batch, N, k]), the one whose
    implied solve R R^T t reproduces the test vectors best under the operator: argmin_p sum || A R_p R_p^T t - t ||_2
    (reference utils/lanczos.py:192-223)."""
    n_cand = initial_vectors.size(-1)
    t = test_vectors.unsqueeze(0)
    cand_solves = inv_roots.matmul(inv_roots.mT.matmul(t))  # [P, *batch, N, T]
    nd = linear_op.dim()
    stacked = cand_solves.permute(*range(1, nd + 1), 0).contiguous()
    stacked = stacked.view(*linear_op.batch_shape, linear_op.matrix_shape[-1], -1)
    back = linear_op.matmul(stacked)
    back = back.view(*linear_op.batch_shape, linear_op.matrix_shape[-1], -1, n_cand).permute(-1, *range(0, nd))
    err = (back - t).norm(2, dim=-2)
    err = err.view(err.size(0), -1).sum(-1)
    best = err.min(0)[1]
    return inv_roots[best].squeeze(0)
