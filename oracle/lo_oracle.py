"""CPU ORACLE -- test infrastructure only, never a product path.

A numpy restatement of the reference's iterative solve / logdet hot path
(cornellius-gp/linear_operator @ /root/reference).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the *checker* -- the shipped package
(``linear_operator_amd``) never imports it and has no CPU fallback.

Parity is PINNED: ``tests/golden/make_golden.py`` runs the real reference (imported
from /root/reference in the build container) on seeded inputs and commits its
outputs as ``tests/golden/*.npz``; ``tests/test_oracle_vs_golden.py`` checks every
function below against those vectors (pivots bit-exact, floats <= 1e-5 rel).

Every function cites the reference file:line it restates.  Arithmetic is done in
the dtype of the inputs (fp32 for the benchmark configs, fp64 for the reference's
own unit-test recipes).  Where the order of floating point operations decides an
*integer* result (pivot selection in pivoted Cholesky) the order is spelled out:
sequential left-to-right sums, products rounded before summation, no FMA -- the
HIP kernels use exactly the same order, so pivots and L agree bit for bit.
"""
from __future__ import annotations

import math

import numpy as np

# ----------------------------------------------------------------------------------
# structured matvecs  (the `_matmul`s that feed CG)
# ----------------------------------------------------------------------------------


def matvec_lowrank_diag(C, d, v):
    """y = C (C^T v) + d o v.

    reference: AddedDiagLinearOperator._matmul (operators/added_diag_linear_operator.py:72-76)
    -> RootLinearOperator._matmul (operators/root_linear_operator.py:68-72)
    -> DenseLinearOperator._t_matmul/_matmul (operators/dense_linear_operator.py:84-88, 60-64).
    C [*B,N,R], d [*B,N], v [*B,N,c].
    """
    t = np.matmul(np.swapaxes(C, -1, -2), v)
    y = np.matmul(C, t)
    return y if d is None else y + d[..., None] * v  # (d None: the plain RootLinearOperator)


def matvec_dense_diag(K, d, v):
    """y = K v + d o v.  reference: added_diag_linear_operator.py:72-76 + dense_linear_operator.py:60-64."""
    y = np.matmul(K, v)
    return y if d is None else y + d[..., None] * v  # (d None: the plain DenseLinearOperator)


def matvec_kron(K1, K2, v):
    """y = (K1 (x) K2) v  ==  vec(K1 V K2^T) with V = v.reshape(n1, n2, c).

    reference: module-level _matmul in operators/kronecker_product_linear_operator.py:34-45
    (per factor: view [n_i, -1], factor matmul, view/transpose(-3,-2)/reshape).
    """
    n1, n2 = K1.shape[-1], K2.shape[-1]
    bshape = v.shape[:-2]
    c = v.shape[-1]
    V = v.reshape(*bshape, n1, n2 * c)
    W = np.matmul(K1, V).reshape(*bshape, n1, n2, c)  # [i1, j2, c]
    W = np.swapaxes(W, -3, -2).reshape(*bshape, n2, n1 * c)  # [j2, i1*c]
    Y = np.matmul(K2, W).reshape(*bshape, n2, n1, c)  # [i2, i1, c]
    return np.swapaxes(Y, -3, -2).reshape(*bshape, n1 * n2, c)


def matvec_kron_diag(K1, K2, d, v):
    """y = (K1 (x) K2) v + d o v (added_diag_linear_operator.py:72-76 over the Kron matvec)."""
    return matvec_kron(K1, K2, v) + d[..., None] * v


# ----------------------------------------------------------------------------------
# linear_cg  (HOT LOOP A)
# ----------------------------------------------------------------------------------


class CGInfo:
    """Side information the reference only exposes through its warning text."""

    def __init__(self):
        self.iterations = 0  # number of loop bodies executed (k+1 at exit)
        self.matvecs = 0  # closure calls (1 + iterations, linear_cg.py:186,248)
        self.tolerance_reached = False
        self.mean_residual = float("nan")
        self.skipped = False  # linear_cg.py:207-208
        self.last_tridiag_iter = 0


def linear_cg(
    matmul_closure,
    rhs,
    n_tridiag=0,
    tolerance=1.0,
    eps=1e-10,
    stop_updating_after=1e-10,
    max_iter=1000,
    max_tridiag_iter=20,
    initial_guess=None,
    preconditioner=None,
    terminate_cg_by_size=False,
):
    """Batched modified preconditioned CG, restating linear_operator/utils/linear_cg.py:98-359.

    Returns (x, t_mat_or_None, CGInfo).  Defaults are the reference's settings defaults
    (settings.py: cg_tolerance=1, max_cg_iterations=1000, max_lanczos_quadrature_iterations=20).
    """
    info = CGInfo()
    rhs = np.asarray(rhs)
    is_vector = rhs.ndim == 1  # :134-136
    if is_vector:
        rhs = rhs[:, None]
    dt = rhs.dtype
    if initial_guess is None:  # :143-149
        initial_guess = np.zeros_like(rhs)
    else:
        initial_guess = np.asarray(initial_guess, dtype=dt)
        if initial_guess.ndim == 1:
            is_vector = True
            initial_guess = initial_guess[:, None]
    precond = preconditioner is not None
    if max_tridiag_iter > max_iter:  # :159-160
        raise RuntimeError("Getting a tridiagonalization larger than the number of CG iterations run is not possible!")
    if not callable(matmul_closure):  # :163-166
        raise RuntimeError("matmul_closure must be a tensor, or a callable object!")

    num_rows = rhs.shape[-2]
    n_iter = min(max_iter, num_rows) if terminate_cg_by_size else max_iter  # :170
    n_tridiag_iter = min(max_tridiag_iter, num_rows)  # :171
    eps = dt.type(eps)  # :172
    stop_after = dt.type(stop_updating_after)

    def colnorm(a):
        return np.sqrt(np.sum(a * a, axis=-2, keepdims=True, dtype=dt))

    rhs_norm = colnorm(rhs)  # :177
    rhs_is_zero = rhs_norm < eps
    rhs_norm = np.where(rhs_is_zero, dt.type(1), rhs_norm)
    rhs = rhs / rhs_norm  # :182
    initial_guess = initial_guess / rhs_norm

    residual = rhs - matmul_closure(initial_guess)  # :186
    info.matvecs = 1
    batch_shape = residual.shape[:-2]
    ncols = rhs.shape[-1]
    result = np.ascontiguousarray(np.broadcast_to(initial_guess, residual.shape)).copy()  # :190

    if np.isnan(residual).any():  # :199-200
        raise RuntimeError("NaNs encountered when trying to perform matrix-vector multiplication")

    residual_norm = colnorm(residual)  # :204
    has_converged = residual_norm < stop_after

    if has_converged.all() and not n_tridiag:  # :207-208
        n_iter = 0
        info.skipped = True
    else:
        precond_residual = preconditioner(residual) if precond else residual.copy()  # :213
        curr_conjugate_vec = precond_residual.copy()
        residual_inner_prod = np.sum(precond_residual * residual, axis=-2, keepdims=True, dtype=dt)

    if n_tridiag:  # :224-236
        t_mat = np.zeros((n_tridiag_iter, n_tridiag_iter) + tuple(batch_shape) + (n_tridiag,), dtype=dt)
        prev_alpha_reciprocal = np.empty(tuple(batch_shape) + (n_tridiag,), dtype=dt)
        prev_beta = np.empty_like(prev_alpha_reciprocal)
    update_tridiag = True
    last_tridiag_iter = 0
    tolerance_reached = False
    k = -1
    one = dt.type(1)
    zero = dt.type(0)

    for k in range(n_iter):  # :245
        mvms = matmul_closure(curr_conjugate_vec)  # :248
        info.matvecs += 1
        alpha = np.sum(curr_conjugate_vec * mvms, axis=-2, keepdims=True, dtype=dt)  # :250-251 / :65-66
        is_zero = alpha < eps  # :254 (negative curvature is zeroed too)
        alpha = np.where(is_zero, one, alpha)
        alpha = residual_inner_prod / alpha
        alpha = np.where(is_zero, zero, alpha)
        alpha = np.where(has_converged, zero, alpha)  # :260
        residual = residual - alpha * mvms  # :264 / :78
        precond_residual = preconditioner(residual) if precond else residual.copy()  # :268 / :82
        result = result + alpha * curr_conjugate_vec  # :31
        beta = residual_inner_prod.copy()  # :34
        residual_inner_prod = np.sum(residual * precond_residual, axis=-2, keepdims=True, dtype=dt)  # :35-36
        is_zero = beta < eps  # :39
        beta = np.where(is_zero, one, beta)
        beta = residual_inner_prod / beta
        beta = np.where(is_zero, zero, beta)
        curr_conjugate_vec = curr_conjugate_vec * beta + precond_residual  # :46

        residual_norm = colnorm(residual)  # :298
        residual_norm = np.where(rhs_is_zero, zero, residual_norm)  # :299
        has_converged = residual_norm < stop_after  # :300

        if (
            k >= min(10, max_iter - 1)
            and bool(residual_norm.mean(dtype=dt) < tolerance)
            and not (n_tridiag and k < min(n_tridiag_iter, max_iter - 1))
        ):  # :302-308
            tolerance_reached = True
            break

        if n_tridiag and k < n_tridiag_iter and update_tridiag:  # :311-332
            alpha_tridiag = alpha[..., 0, :n_tridiag]
            beta_tridiag = beta[..., 0, :n_tridiag]
            a_is_zero = alpha_tridiag == 0
            alpha_reciprocal = one / np.where(a_is_zero, one, alpha_tridiag)
            if k == 0:
                t_mat[k, k] = alpha_reciprocal
            else:
                t_mat[k, k] = alpha_reciprocal + prev_beta * prev_alpha_reciprocal
                off = np.sqrt(prev_beta) * prev_alpha_reciprocal
                t_mat[k, k - 1] = off
                t_mat[k - 1, k] = off
                if t_mat[k - 1, k].max() < 1e-6:  # :326
                    update_tridiag = False
            last_tridiag_iter = k
            prev_alpha_reciprocal = alpha_reciprocal.copy()
            prev_beta = beta_tridiag.copy()

    result = result * rhs_norm  # :335
    info.iterations = k + 1 if n_iter > 0 else 0
    info.tolerance_reached = tolerance_reached
    info.mean_residual = float(residual_norm.mean(dtype=dt))
    info.last_tridiag_iter = last_tridiag_iter
    info.warned = (not tolerance_reached) and n_iter > 0  # :337
    if is_vector:
        result = result[..., 0]
    if n_tridiag:
        t_mat = t_mat[: last_tridiag_iter + 1, : last_tridiag_iter + 1]  # :353
        nb = len(batch_shape)
        perm = (t_mat.ndim - 1,) + tuple(range(2, 2 + nb)) + (0, 1)  # :354-357
        return result, np.ascontiguousarray(np.transpose(t_mat, perm)), info
    return result, None, info


# ----------------------------------------------------------------------------------
# pivoted Cholesky  (HOT LOOP B)
# ----------------------------------------------------------------------------------


def _seq_dot_rows(A, Bm):
    """sum_r A[..., r] * B[..., r], sequential in r, product rounded first (no FMA)."""
    R = A.shape[-1]
    acc = A[..., 0] * Bm[..., 0]
    for r in range(1, R):
        acc = acc + A[..., r] * Bm[..., r]
    return acc


class LowRankRowSource:
    """diag / row access for RootLinearOperator(C): root_linear_operator.py:22-28 (diag), :37-50 (rows)."""

    def __init__(self, C):
        self.C = C
        self.n = C.shape[-2]
        self.batch_shape = C.shape[:-2]
        self.dtype = C.dtype

    def diag(self):
        return _seq_dot_rows(self.C, self.C)

    def row(self, idx):  # idx [*B] -> [*B, N]
        Cr = np.take_along_axis(self.C, idx[..., None, None], axis=-2)  # [*B,1,R]
        return _seq_dot_rows(np.broadcast_to(Cr, self.C.shape), self.C)


class DenseRowSource:
    """dense_linear_operator.py:37-40 (diag), :47-50 (rows)."""

    def __init__(self, K):
        self.K = K
        self.n = K.shape[-1]
        self.batch_shape = K.shape[:-2]
        self.dtype = K.dtype

    def diag(self):
        return np.diagonal(self.K, axis1=-2, axis2=-1).copy()

    def row(self, idx):
        return np.take_along_axis(self.K, idx[..., None, None], axis=-2)[..., 0, :]


class KronRowSource:
    """kronecker_product_linear_operator.py:20-27,188-191 (diag), :198-216 (rows)."""

    def __init__(self, K1, K2):
        self.K1, self.K2 = K1, K2
        self.n1, self.n2 = K1.shape[-1], K2.shape[-1]
        self.n = self.n1 * self.n2
        self.batch_shape = K1.shape[:-2]
        self.dtype = K1.dtype

    def diag(self):
        d1 = np.diagonal(self.K1, axis1=-2, axis2=-1)
        d2 = np.diagonal(self.K2, axis1=-2, axis2=-1)
        return (d1[..., :, None] * d2[..., None, :]).reshape(*self.batch_shape, self.n)

    def row(self, idx):
        i1 = idx // self.n2
        i2 = idx % self.n2
        r1 = np.take_along_axis(self.K1, i1[..., None, None], axis=-2)[..., 0, :]
        r2 = np.take_along_axis(self.K2, i2[..., None, None], axis=-2)[..., 0, :]
        return (r1[..., :, None] * r2[..., None, :]).reshape(*self.batch_shape, self.n)


class SumRowSource:
    """diag / row access for SumLinearOperator(*terms): `_diagonal` = sum of the terms' diagonals and rows through
    `_get_indices` = sum of the terms' entries, both Python `sum()` left to right (sum_linear_operator.py:31-32,39-41)."""

    def __init__(self, *sources):
        self.sources = sources
        self.n = sources[0].n
        self.batch_shape = sources[0].batch_shape
        self.dtype = sources[0].dtype

    def diag(self):
        acc = self.sources[0].diag()
        for s in self.sources[1:]:
            acc = acc + s.diag()
        return acc

    def row(self, idx):
        acc = self.sources[0].row(idx)
        for s in self.sources[1:]:
            acc = acc + s.row(idx)
        return acc


def matvec_sum(matvecs, d, v):
    """SumLinearOperator._matmul (sum_linear_operator.py:47-51) of the closures `matvecs` (left to right) plus the
    diagonal term of an enclosing AddedDiagLinearOperator (added_diag_linear_operator.py:72-76); d may be None."""
    acc = matvecs[0](v)
    for mv in matvecs[1:]:
        acc = acc + mv(v)
    if d is not None:
        acc = acc + d[..., None] * v
    return acc


def pivoted_cholesky(src, rank, error_tol=1e-3):
    """Greedy partial pivoted Cholesky, restating functions/_pivoted_cholesky.py:14-105.

    src: a *RowSource (diag(), row(idx)).  Returns (L [*B,N,m], permutation [*B,N] int64).
    Batch-global loop condition (:57), first-max tie-break (:61-63, torch.max on CPU returns the
    first maximal index), sequential-in-j Schur update with products rounded before the sum (:83-89).
    """
    batch_shape = tuple(src.batch_shape)
    N = src.n
    dt = src.dtype
    diag = src.diag().astype(dt).copy()  # :25-30
    max_iter = min(rank, N)  # :33
    L = np.zeros(batch_shape + (max_iter, N), dtype=dt)  # :36-42
    orig_error = diag.max(axis=-1)  # :43
    errors = np.abs(diag).sum(axis=-1, dtype=dt) / orig_error  # :44
    perm = np.broadcast_to(np.arange(N, dtype=np.int64), batch_shape + (N,)).copy()  # :47-48
    m = 0
    while m == 0 or (m < max_iter and errors.max() > error_tol):  # :57
        permuted_diags = np.take_along_axis(diag, perm[..., m:], axis=-1)  # :61
        max_idx = permuted_diags.argmax(axis=-1)  # first max
        max_val = np.take_along_axis(permuted_diags, max_idx[..., None], axis=-1)[..., 0]
        max_idx = max_idx + m
        old_pi_m = perm[..., m].copy()  # :67-70
        new_pi_m = np.take_along_axis(perm, max_idx[..., None], axis=-1)[..., 0]
        perm[..., m] = new_pi_m
        np.put_along_axis(perm, max_idx[..., None], old_pi_m[..., None], axis=-1)
        pi_m = perm[..., m]
        L_m = L[..., m, :]
        piv = np.sqrt(max_val).astype(dt)
        np.put_along_axis(L_m, pi_m[..., None], piv[..., None], axis=-1)  # :73-74
        if m + 1 < N:  # :77
            row = src.row(pi_m)  # :79
            pi_i = perm[..., m + 1 :]
            L_m_new = np.take_along_axis(row, pi_i, axis=-1)  # :82
            if m > 0:  # :83-89
                acc = None
                for j in range(m):
                    Lj = L[..., j, :]
                    upd = np.take_along_axis(Lj, pi_m[..., None], axis=-1)  # [*B,1]
                    prev = np.take_along_axis(Lj, pi_i, axis=-1)
                    term = upd * prev
                    acc = term if acc is None else acc + term
                L_m_new = L_m_new - acc
            L_m_new = L_m_new / piv[..., None]  # :91
            np.put_along_axis(L_m, pi_i, L_m_new, axis=-1)  # :92
            cur = np.take_along_axis(diag, pi_i, axis=-1)  # :94-95
            newd = cur - L_m_new * L_m_new
            np.put_along_axis(diag, pi_i, newd, axis=-1)
            errors = np.abs(newd).sum(axis=-1, dtype=dt) / orig_error  # :99
        m += 1
    return np.ascontiguousarray(np.swapaxes(L[..., :m, :], -1, -2)), perm  # :105


# ----------------------------------------------------------------------------------
# pivoted-Cholesky / QR Woodbury preconditioner
# ----------------------------------------------------------------------------------


class Preconditioner:
    """P = L L^T + D, restating AddedDiagLinearOperator._preconditioner / _init_cache*
    (operators/added_diag_linear_operator.py:95-184)."""

    def __init__(self, L, d):
        # L [*B,N,k], d [*B,N]
        dt = L.dtype
        N, k = L.shape[-2:]
        noise = d[..., None]  # :146
        self.constant_diag = bool(np.array_equal(noise, noise[..., :1, :] * np.ones_like(noise)))  # :149-150
        eye = np.broadcast_to(np.eye(k, dtype=dt), L.shape[:-2] + (k, k))
        if self.constant_diag:  # :161-172
            noise = noise[..., :1, :]
            Q, R = np.linalg.qr(np.concatenate([L, np.sqrt(noise) * eye], axis=-2))
            self.Q = np.ascontiguousarray(Q[..., :N, :])
            logdet = 2 * np.log(np.abs(np.diagonal(R, axis1=-2, axis2=-1))).sum(-1)
            logdet = logdet + (N - k) * np.log(noise[..., 0, 0])
        else:  # :174-184
            sq = np.sqrt(noise)
            Q, R = np.linalg.qr(np.concatenate([L / sq, eye], axis=-2))
            self.Q = np.ascontiguousarray(Q[..., :N, :] / sq)
            logdet = 2 * np.log(np.abs(np.diagonal(R, axis1=-2, axis2=-1))).sum(-1)
            logdet = logdet - np.log(1.0 / noise).sum(axis=(-1, -2))
        self.R = R
        self.noise = noise.astype(dt)
        self.logdet = logdet.astype(dt)
        self.L = L
        self.d = d

    def apply(self, r):
        """precondition_closure, added_diag_linear_operator.py:135-140."""
        qqt = np.matmul(self.Q, np.matmul(np.swapaxes(self.Q, -1, -2), r))
        if self.constant_diag:
            return (1 / self.noise) * (r - qqt)
        return r / self.noise - qqt


# ----------------------------------------------------------------------------------
# Lanczos (HOT LOOP C), tridiagonal eigensolve, SLQ
# ----------------------------------------------------------------------------------


def lanczos_tridiag(matmul_closure, max_iter, init_vecs, tol=1e-5):
    """Lanczos with full re-orthogonalisation, restating utils/lanczos.py:9-164.

    init_vecs [*B,N,P] must be supplied (the reference's randn default is not reproducible).
    Returns (q_mat [P,*B,N,k'], t_mat [P,*B,k',k']) -- leading P squeezed iff P == 1 (:159-161).
    """
    init_vecs = np.asarray(init_vecs)
    dt = init_vecs.dtype
    batch_shape = init_vecs.shape[:-2]
    N, P = init_vecs.shape[-2:]
    num_iter = min(max_iter, N)  # :57
    q_mat = np.zeros((num_iter,) + tuple(batch_shape) + (N, P), dtype=dt)  # :69-77
    t_mat = np.zeros((num_iter, num_iter) + tuple(batch_shape) + (P,), dtype=dt)

    def vnorm(a, keepdims=False):
        return np.sqrt(np.sum(a * a, axis=-2, keepdims=keepdims, dtype=dt))

    q0 = init_vecs / vnorm(init_vecs)[..., None, :]  # :81
    q_mat[0] = q0
    r_vec = matmul_closure(q0)  # :85
    alpha0 = np.sum(q0 * r_vec, axis=-2, dtype=dt)
    r_vec = r_vec - alpha0[..., None, :] * q0  # :89
    beta0 = vnorm(r_vec)
    t_mat[0, 0] = alpha0
    if num_iter > 1:
        t_mat[0, 1] = beta0
        t_mat[1, 0] = beta0
        q_mat[1] = r_vec / beta0[..., None, :]  # :98
    k = 0
    for k in range(1, num_iter):  # :101
        q_prev = q_mat[k - 1]
        q_curr = q_mat[k]
        beta_prev = t_mat[k, k - 1][..., None, :]
        r_vec = matmul_closure(q_curr) - q_prev * beta_prev  # :108
        alpha_curr = np.sum(q_curr * r_vec, axis=-2, keepdims=True, dtype=dt)
        t_mat[k, k] = alpha_curr[..., 0, :]
        if k + 1 < num_iter:  # :114
            r_vec = r_vec - alpha_curr * q_curr
            Qk = q_mat[: k + 1]
            corr = np.sum(r_vec[None] * Qk, axis=-2, keepdims=True, dtype=dt)  # :118
            corr = np.sum(Qk * corr, axis=0, dtype=dt)
            r_vec = r_vec - corr
            r_norm = vnorm(r_vec, keepdims=True)
            r_vec = r_vec / r_norm
            beta_curr = r_norm[..., 0, :]
            t_mat[k, k + 1] = beta_curr
            t_mat[k + 1, k] = beta_curr
            inner = np.sum(Qk * r_vec[None], axis=-2, dtype=dt)  # :131
            could = False
            for _ in range(10):  # :133-142 (signed compare)
                if not np.sum(inner > tol):
                    could = True
                    break
                corr = np.sum(r_vec[None] * Qk, axis=-2, keepdims=True, dtype=dt)
                corr = np.sum(Qk * corr, axis=0, dtype=dt)
                r_vec = r_vec - corr
                r_norm = vnorm(r_vec, keepdims=True)
                r_vec = r_vec / r_norm
                inner = np.sum(Qk * r_vec[None], axis=-2, dtype=dt)
            q_mat[k + 1] = r_vec  # :145
            if np.sum(np.abs(beta_curr) > 1e-6) == 0 or not could:  # :147
                break
    num_iter = k + 1  # :151
    nb = len(batch_shape)
    qp = (q_mat.ndim - 1,) + tuple(range(1, 1 + nb)) + (q_mat.ndim - 2, 0)
    q_out = np.ascontiguousarray(np.transpose(q_mat[:num_iter], qp))
    tp = (t_mat.ndim - 1,) + tuple(range(2, 2 + nb)) + (0, 1)
    t_out = np.ascontiguousarray(np.transpose(t_mat[:num_iter, :num_iter], tp))
    if P == 1:  # squeeze_(0) only acts on a size-1 dim
        q_out, t_out = q_out[0], t_out[0]
    return q_out, t_out


def lanczos_tridiag_to_diag(t_mat):
    """eigh of each tridiagonal + clamp, restating utils/lanczos.py:167-189."""
    evals, evecs = np.linalg.eigh(t_mat)
    mask = evals >= 0
    evecs = evecs * mask[..., None, :].astype(evecs.dtype)  # zero eigenvector COLUMNS (:186)
    evals = np.where(mask, evals, evals.dtype.type(1))  # :187
    return evals, evecs


def slq_logdet(n, evals, evecs):
    """StochasticLQ.to_dense with funcs=[log], restating utils/stochastic_lq.py:67-82.

    evals [P,*B,k], evecs [P,*B,k,k] -> logdet [*B]  (scaled by n / P only, :80).
    """
    P = evals.shape[0]
    dt = evals.dtype
    res = np.zeros(evals.shape[1:-1], dtype=dt)
    for j in range(P):
        first = evecs[j][..., 0, :]
        dots = np.sum(first * first * np.log(evals[j]), axis=-1, dtype=dt)
        res = res + dt.type(n / float(P)) * dots
    return res


# ----------------------------------------------------------------------------------
# orchestration: solve / inv_quad_logdet with the default preconditioner
# ----------------------------------------------------------------------------------


def solve(matmul_closure, row_src, d, rhs, tolerance=1.0, max_iter=1000, rank=15, min_precond_size=2000,
          precond_tol=1e-3):
    """`_solve` dispatch for N > max_cholesky_size, restating functions/_solve.py:19-22 +
    LinearOperator._solve (operators/_linear_operator.py:781-803)."""
    N = rhs.shape[-2]
    pre = None
    if rank > 0 and N >= min_precond_size:
        L, _ = pivoted_cholesky(row_src, rank, precond_tol)
        pre = Preconditioner(L, d)
    x, _, info = linear_cg(matmul_closure, rhs, tolerance=tolerance, max_iter=max_iter,
                           preconditioner=pre.apply if pre else None)
    return x, info, pre


def inv_quad_logdet(matmul_closure, row_src, d, inv_quad_rhs, probes, tolerance=1.0, max_iter=1000,
                    max_tridiag_iter=20, rank=15, min_precond_size=2000, precond_tol=1e-3):
    """InvQuadLogdet.forward with injected (already normalised) probe vectors, restating
    functions/_inv_quad_logdet.py:112-153 and LinearOperator.inv_quad_logdet
    (operators/_linear_operator.py:1773-1804).  probes [*B,N,P] unit columns."""
    N = probes.shape[-2]
    P = probes.shape[-1]
    pre = None
    logdet_p = 0.0
    if rank > 0 and N >= min_precond_size:
        L, _ = pivoted_cholesky(row_src, rank, precond_tol)
        pre = Preconditioner(L, d)
        logdet_p = pre.logdet
    rhs = np.concatenate([probes, inv_quad_rhs], axis=-1) if inv_quad_rhs is not None else probes
    solves, t_mat, info = linear_cg(matmul_closure, rhs, n_tridiag=P, tolerance=tolerance, max_iter=max_iter,
                                    max_tridiag_iter=max_tridiag_iter,
                                    preconditioner=pre.apply if pre else None)
    if np.isnan(t_mat).any():  # :141-142
        logdet = np.full(probes.shape[:-2], np.nan, dtype=probes.dtype)
    else:
        evals, evecs = lanczos_tridiag_to_diag(t_mat)
        logdet = slq_logdet(N, evals, evecs)
    inv_quad = None
    if inv_quad_rhs is not None:
        inv_quad = np.sum(solves[..., P:] * inv_quad_rhs, axis=-2)  # :151-153
    return inv_quad, logdet + logdet_p, solves, t_mat, info, pre


# ----------------------------------------------------------------------------------
# LowRankRootAddedDiagLinearOperator closed forms
# ----------------------------------------------------------------------------------


def woodbury_chol_cap_mat(C, d):
    """Cholesky factor of the capacitance matrix I_R + C^T D^-1 C, restating
    LowRankRootAddedDiagLinearOperator.chol_cap_mat (operators/low_rank_root_added_diag_linear_operator.py:36-47).
    C [*B,N,R], d [*B,N]."""
    V = np.swapaxes(C, -1, -2)
    cap = np.eye(C.shape[-1], dtype=C.dtype) + V @ (C / d[..., None])
    return np.linalg.cholesky(cap)


def woodbury_solve(C, d, rhs):
    """(C C^T + D)^-1 rhs = D^-1 rhs - D^-1 C cap^-1 C^T D^-1 rhs, restating `_solve`
    (low_rank_root_added_diag_linear_operator.py:62-89): two triangular solves with chol_cap_mat."""
    from scipy.linalg import solve_triangular

    chol = woodbury_chol_cap_mat(C, d)
    Dir = rhs / d[..., None]
    res = np.swapaxes(C, -1, -2) @ Dir
    out = np.empty_like(res)
    for idx in np.ndindex(res.shape[:-2]):
        y = solve_triangular(chol[idx], res[idx], lower=True)
        out[idx] = solve_triangular(chol[idx].T, y, lower=False)
    return Dir - (C @ out) / d[..., None]


def woodbury_logdet(C, d):
    """logdet(C C^T + D) = 2 sum log diag(chol_cap_mat) + sum log d, restating `_logdet` (:97-103)."""
    chol = woodbury_chol_cap_mat(C, d)
    return 2.0 * np.log(np.diagonal(chol, axis1=-2, axis2=-1)).sum(-1) + np.log(d).sum(-1)


# ----------------------------------------------------------------------------------
# RootDecomposition.forward
# ----------------------------------------------------------------------------------


def root_decomposition(matmul_closure, init_vecs, max_iter, tridiagonal_jitter=1e-6):
    """Lanczos root / inverse root, restating RootDecomposition.forward (functions/_root_decomposition.py:49-85):
    Q, T <- Lanczos; T += jitter * min(diag T) * I (:67-70); (lambda, V) <- eigh with the negative-eigenvalue clamp;
    Q <- Q V; root = Q o sqrt(lambda); inverse = Q / sqrt(lambda).
    init_vecs [*B,N,P] -> (root, inverse) [P,*B,N,k] (leading P squeezed iff P == 1)."""
    q_mat, t_mat = lanczos_tridiag(matmul_closure, max_iter, init_vecs)
    single = t_mat.ndim == init_vecs.ndim  # one probe vector: no leading P (:62-64)
    if single:
        q_mat, t_mat = q_mat[None], t_mat[None]
    k = t_mat.shape[-1]
    mins = np.diagonal(t_mat, axis1=-2, axis2=-1).min(axis=-1)[..., None, None]
    jit = (t_mat.dtype.type(tridiagonal_jitter) * mins) * np.eye(k, dtype=t_mat.dtype)
    evals, evecs = lanczos_tridiag_to_diag(t_mat + jit)
    q_mat = q_mat @ evecs
    s = np.sqrt(evals)[..., None, :]
    root, inverse = q_mat * s, q_mat / s
    if single:
        root, inverse = root[0], inverse[0]
    return root, inverse


def diagonalization(matmul_closure, init_vecs, max_iter, tridiagonal_jitter=1e-6):
    """Partial eigendecomposition A ~= Q diag(lambda) Q^T, restating Diagonalization.forward
    (functions/_diagonalization.py:30-58): Q, T <- Lanczos; T += jitter_mat where jitter_mat is
    `diag_embed(jitter * min(diag T))` -- a [...,1,1] tensor -- EXPANDED to T's shape, i.e. the constant is added to
    EVERY entry of T, not only its diagonal (:48-50; reproduced as written); (lambda, V) <- eigh of that full
    symmetric matrix with the negative-eigenvalue clamp (lanczos_tridiag_to_diag); Q <- Q V.
    init_vecs [*B,N,1] -> (evals [*B,k], Q [*B,N,k])."""
    q_mat, t_mat = lanczos_tridiag(matmul_closure, max_iter, init_vecs)
    mins = np.diagonal(t_mat, axis1=-2, axis2=-1).min(axis=-1)[..., None, None]
    jit = np.broadcast_to(t_mat.dtype.type(tridiagonal_jitter) * mins, t_mat.shape)
    evals, evecs = lanczos_tridiag_to_diag(t_mat + jit)
    return evals, q_mat @ evecs


def kron_added_diag_eig(K1, K2):
    """Per-factor eigendecompositions of K1 (x) K2 (kronecker_product_linear_operator.py:338-360, each factor through
    LinearOperator._symeig :878-901: eigh in fp64, eigenvalues clamped at 0): evals [*B, n1 n2] = l1_i l2_j in the
    Kronecker order, and the factors' eigenvector matrices."""
    l1, q1 = np.linalg.eigh(K1.astype(np.float64))
    l2, q2 = np.linalg.eigh(K2.astype(np.float64))
    l1, l2 = np.maximum(l1, 0.0), np.maximum(l2, 0.0)
    evals = (l1[..., :, None] * l2[..., None, :]).reshape(*l1.shape[:-1], -1)
    return evals, q1, q2


def kron_added_diag_solve(K1, K2, sigma2, rhs):
    """(K1 (x) K2 + sigma2 I)^-1 rhs with a constant diagonal, restating
    KroneckerProductAddedDiagLinearOperator._solve (kronecker_product_added_diag_linear_operator.py:147-161): in fp64,
    Q (Q^T rhs / (lambda + sigma2)) with Q = Q1 (x) Q2 applied factor by factor.  sigma2 [*B,1], rhs [*B, n1 n2, c]."""
    evals, q1, q2 = kron_added_diag_eig(K1, K2)
    n1, n2, c = K1.shape[-1], K2.shape[-1], rhs.shape[-1]

    def kron_apply(a, b, v):  # (a (x) b) v
        v4 = v.reshape(*v.shape[:-2], n1, n2, c)
        return np.einsum("...ij,...ab,...jbc->...iac", a, b, v4).reshape(v.shape)

    t = kron_apply(np.swapaxes(q1, -1, -2), np.swapaxes(q2, -1, -2), rhs.astype(np.float64))
    t = t / (evals + sigma2.astype(np.float64))[..., None]
    return kron_apply(q1, q2, t)


def kron_added_diag_logdet(K1, K2, sigma2):
    """logdet(K1 (x) K2 + sigma2 I) = sum log(lambda + sigma2)  (:86-90)."""
    evals, _, _ = kron_added_diag_eig(K1, K2)
    return np.log(evals + sigma2.astype(np.float64)).sum(-1)


# ----------------------------------------------------------------------------------
# MINRES with shifts
# ----------------------------------------------------------------------------------


class MinresInfo:
    iterations = 0
    converged = False


def minres(matmul_closure, rhs, eps=1e-25, shifts=None, value=None, max_iter=1000, preconditioner=None,
           tolerance=1e-4):
    """Restates linear_operator.utils.minres.minres (utils/minres.py:10-207 and the update block :210-282):
    solutions of (value * K + shift_q I) x = rhs for all shifts at once -- one preconditioned Lanczos recurrence
    (alpha, beta, z, q) shared by the shifts, one QR (Givens) recurrence and one pair of search vectors per shift.
    rhs [*B,N,c]; shifts None | [Q] | [Q,*B]; returns ([Q,*B,N,c] or [*B,N,c] when there is one shift, info).
    Stop: every 10th iteration, mean over everything of ||search update|| / ||solution|| < tolerance (:178-183)."""
    dt = rhs.dtype
    info = MinresInfo()
    if shifts is None:
        shifts = np.zeros((), dtype=dt)  # :43-44
    shifts = np.asarray(shifts, dtype=dt)
    squeeze = rhs.ndim == 1
    if squeeze:
        rhs = rhs[..., None]
    rhs_norm = np.linalg.norm(rhs, axis=-2, keepdims=True).astype(dt)  # :52-55
    rhs_is_zero = rhs_norm < 1e-10
    rhs_norm = np.where(rhs_is_zero, dt.type(1), rhs_norm)
    rhs = rhs / rhs_norm
    max_iter = min(max_iter, rhs.shape[-2] + 1)  # :60
    eps = dt.type(eps)
    val = None if value is None else dt.type(value)

    def mm(v):
        p = matmul_closure(v).astype(dt)
        return p if val is None else p * val

    prod = mm(rhs)  # :66-68 (its values are not used: shape only)
    pad = prod.ndim - shifts.ndim + 1  # _pad_with_singletons(shifts, 0, ...)  :71
    shifts = shifts.reshape(shifts.shape + (1,) * pad)
    nq = shifts.shape[0]
    solution = np.zeros((nq,) + prod.shape, dtype=dt)
    z2 = np.zeros_like(prod)
    z1 = np.broadcast_to(rhs, prod.shape).copy()
    q1 = z1.copy() if preconditioner is None else preconditioner(z1).astype(dt)
    beta_prev = np.sqrt((z1 * q1).sum(-2, keepdims=True)).astype(dt)  # :80
    with np.errstate(invalid="ignore", divide="ignore"):  # an all-zero column: 0 / 0 here, masked at the end (:201)
        z1 = z1 / beta_prev
        q1 = q1 / beta_prev
    sc_shape = solution.shape[:-2] + (1, rhs.shape[-1])
    cos2, sin2 = np.ones(sc_shape, dt), np.zeros(sc_shape, dt)
    cos1, sin1 = np.ones(sc_shape, dt), np.zeros(sc_shape, dt)
    s2 = np.zeros_like(solution)
    s1 = np.zeros_like(solution)
    scale_prev = np.broadcast_to(beta_prev, sc_shape).copy()  # :113
    for i in range(max_iter + 2):  # :134
        prod = mm(q1)
        alpha = (prod * q1).sum(-2, keepdims=True).astype(dt)  # :141-142
        zc = prod - alpha * z1 - beta_prev * z2  # :144
        qc = zc.copy() if preconditioner is None else preconditioner(zc).astype(dt)
        beta = np.maximum(np.sqrt((zc * qc).sum(-2, keepdims=True)), eps).astype(dt)  # :147-150
        zc = zc / beta
        qc = qc / beta
        # ---- QR / solution update for every shift (:236-282)
        subsub = sin2 * beta_prev
        sub = cos2 * beta_prev
        alpha_s = alpha + shifts
        diag = alpha_s * cos1 - sin1 * sub
        sub = sub * cos1 + sin1 * alpha_s
        radius = np.sqrt(diag * diag + beta * beta)
        cosc = diag / radius
        sinc = beta / radius
        diag = diag * cosc + sinc * beta
        scale_curr = -(scale_prev * sinc)
        scale_prev = scale_prev * cosc
        sc = (q1 - sub * s1 - subsub * s2) / diag
        upd = sc * scale_prev
        solution = solution + upd
        info.iterations = i + 1
        if (i + 1) % 10 == 0:  # :178-183
            with np.errstate(invalid="ignore", divide="ignore"):
                conv = (np.linalg.norm(upd, axis=-2) / np.linalg.norm(solution, axis=-2)).mean()
            if conv < tolerance:
                info.converged = True
                break
        z2, z1 = z1, zc  # :186-198
        q1 = qc
        beta_prev = beta
        cos2, cos1 = cos1, cosc
        sin2, sin1 = sin1, sinc
        s2, s1 = s1, sc
        scale_prev = scale_curr
    solution = np.where(rhs_is_zero, dt.type(0), solution)  # :201
    if squeeze:
        solution = solution[..., 0]
        rhs_norm = rhs_norm[..., 0]
    if shifts.size == 1:
        solution = solution[0]  # :208-210
    return solution * rhs_norm, info


def contour_integral_quad(matmul_closure, rhs, inverse=False, num_contour_quadrature=15, max_lanczos_iter=20,
                          shift_offset=0.0, minres_tolerance=1e-4, max_iter=1000):
    """Restates contour_integral_quad without a preconditioner (utils/contour_integral_quad.py:14-156):
    spectrum ends from the tridiagonal of a 20-step CG run on the first column (:56-96), quadrature nodes and weights
    from the Jacobi elliptic functions (:100-126, scipy), all shifted solves by one MINRES call with value = -1
    (:137-144), one more product with K unless `inverse` (:147-148).
    rhs [*B,N,c] -> (solves [Q,*B,N,c], weights [Q,*B,1,1], no_shift_solves, shifts [Q+1,*B])."""
    from scipy.special import ellipj, ellipk

    dt = rhs.dtype
    init = np.ascontiguousarray(rhs[..., :1])
    _, t_mat, _ = linear_cg(matmul_closure, init, n_tridiag=1, max_iter=max_lanczos_iter, tolerance=1e-5,
                            max_tridiag_iter=max_lanczos_iter)
    t_mat = t_mat[0]  # squeeze(0): one tridiagonal column
    eigs = np.linalg.eigvalsh(t_mat)
    if eigs.min() <= 0:
        raise RuntimeError("oracle: the diagonal fallback of :95-96 is not restated")
    max_eig, min_eig = eigs.max(-1), eigs.min(-1)
    k2 = min_eig / max_eig
    nq = num_contour_quadrature
    flat_shifts = np.zeros((nq + 1, k2.size), dtype=dt)
    flat_weights = np.zeros((nq, k2.size), dtype=dt)
    for i, (sub_k2, sub_min) in enumerate(zip(k2.reshape(-1).tolist(), min_eig.reshape(-1).tolist())):
        Kp = ellipk(1 - sub_k2)
        t = 1j * (np.arange(1, nq + 1) - 0.5) * Kp / nq
        sn, cn, dn, _ = ellipj(np.imag(t), 1 - sub_k2)
        cn = 1.0 / cn
        dn = dn * cn
        sn = 1j * sn * cn
        w = np.sqrt(sub_min) * sn
        flat_shifts[1:, i] = np.real(np.power(w, 2)).astype(dt)
        constant = -2 * Kp * np.sqrt(sub_min) / (math.pi * nq)
        flat_weights[:, i] = (cn * dn).astype(dt) * dt.type(constant)
    weights = flat_weights.reshape((nq,) + k2.shape + (1, 1))
    shifts = flat_shifts.reshape((nq + 1,) + k2.shape) - dt.type(shift_offset)
    solves, _ = minres(matmul_closure, rhs, shifts=shifts, value=-1, max_iter=max_iter, tolerance=minres_tolerance)
    no_shift = solves[0]
    solves = solves[1:]
    if not inverse:
        solves = np.stack([matmul_closure(sv) for sv in solves])
    return solves, weights, no_shift, shifts


def sqrt_inv_matmul(matmul_closure, rhs, lhs=None, num_contour_quadrature=15):
    """SqrtInvMatmul.forward (functions/_sqrt_inv_matmul.py:18-51): A^{-1/2} rhs, or (lhs A^{-1/2} rhs,
    -sum(lhs o no_shift_solves^T)) = (..., diag(lhs A^-1 lhs^T)) when lhs is given (the no-shift solve has value -1)."""
    if lhs is None:
        solves, weights, _, _ = contour_integral_quad(matmul_closure, rhs, inverse=True,
                                                       num_contour_quadrature=num_contour_quadrature)
        return (solves * weights).sum(0)
    terms = np.concatenate([rhs, np.swapaxes(lhs, -1, -2)], axis=-1)
    solves, weights, no_shift, _ = contour_integral_quad(matmul_closure, terms, inverse=True,
                                                          num_contour_quadrature=num_contour_quadrature)
    c = rhs.shape[-1]
    res = lhs @ (solves[..., :c] * weights).sum(0)
    inv_quad = -(np.swapaxes(no_shift[..., c:], -1, -2) * lhs).sum(-1)
    return res, inv_quad


# ----------------------------------------------------------------------------------
# Backward passes
# ----------------------------------------------------------------------------------


def bilinear_derivative_dense(U, V):
    """d/dK sum_d u_d^T K v_d = U V^T  (operators/dense_linear_operator.py:69-71)."""
    return U @ np.swapaxes(V, -1, -2)


def bilinear_derivative_diag(U, V, constant=False):
    """Diag: sum_d U o V (operators/diag_linear_operator.py:37-45); ConstantDiag: its sum over N, [*B,1] (:337-344)."""
    res = (U * V).sum(-1)
    return res.sum(-1, keepdims=True) if constant else res


def bilinear_derivative_root(C, U, V):
    """K = C C^T: d/dC sum_d u_d^T C C^T v_d = U (V^T C) + V (U^T C) -- what the generic autograd version
    (operators/_linear_operator.py:336-393) yields for RootLinearOperator._matmul (root_linear_operator.py:68-72)."""
    return U @ (np.swapaxes(V, -1, -2) @ C) + V @ (np.swapaxes(U, -1, -2) @ C)


def root_decomposition_backward(q_mat, evals, root_grad=None, inverse_grad=None):
    """RootDecomposition.backward (functions/_root_decomposition.py:104-171) for ONE probe vector: with
    R_inv = Q / sqrt(lambda), the factors handed to `_bilinear_derivative` are
    left = grad_R - R_inv grad_Rinv^T R_inv  and  right = R_inv / 2.   q_mat [*B,N,k] (= Q V), evals [*B,k]."""
    inverse = q_mat / np.sqrt(evals)[..., None, :]
    left = np.zeros_like(inverse)
    if root_grad is not None:
        left = left + root_grad
    if inverse_grad is not None:
        left = left - inverse @ np.swapaxes(inverse_grad, -1, -2) @ inverse
    return left, inverse / 2.0


def diagonalization_backward(q_mat, evals, evals_grad, evecs_grad):
    """Diagonalization.backward (functions/_diagonalization.py:62-88): dense dL/dM =
    Q (K~^T o (Q^T dL/dQ)) Q^T + Q diag(dL/dlambda) Q^T,  K~_ij = 1_{i != j} / (lambda_i - lambda_j + 1e-10)."""
    kmat = 1.0 / (evals[..., :, None] - evals[..., None, :] + 1e-10)
    idx = np.arange(evals.shape[-1])
    kmat[..., idx, idx] = 0.0
    inner = np.swapaxes(kmat, -1, -2) * (np.swapaxes(q_mat, -1, -2) @ evecs_grad)
    term1 = q_mat @ inner @ np.swapaxes(q_mat, -1, -2)
    term2 = (q_mat * evals_grad[..., None, :]) @ np.swapaxes(q_mat, -1, -2)
    return term1 + term2


def solve_backward(solve_fn, right_solves, grad_output):
    """Solve.backward without a left tensor (functions/_solve.py:70-115): returns (rhs_grad, U, V) where
    (U, V) = ([L | R], -[R | L] / 2) are the factors handed to `_bilinear_derivative`, L = A^-1 grad_output."""
    left_solves = solve_fn(grad_output)
    U = np.concatenate([left_solves, right_solves], axis=-1)
    V = np.concatenate([right_solves, left_solves], axis=-1) * right_solves.dtype.type(-0.5)
    return left_solves, U, V


def inv_quad_backward(inv_quad_solves, grad_output):
    """InvQuad.backward (functions/_inv_quad.py:63-93): (rhs_grad, U, V)."""
    neg = -inv_quad_solves * grad_output[..., None, :]
    return -2.0 * neg, neg, inv_quad_solves


def inv_quad_logdet_backward(solves, probe_vectors, probe_vector_norms, num_probes, inv_quad_grad, logdet_grad,
                             precond_apply=None):
    """InvQuadLogdet.backward (functions/_inv_quad_logdet.py:163-226) without the preconditioner-tensor gradients:
    (rhs_grad, U, V) with U = [probe solves * norms * g / P | -iq_solves * g_iq], V = [P^-1 (probes * norms) | iq_solves]."""
    coef = 1.0 / probe_vectors.shape[-1]
    ld = logdet_grad[..., None, None]
    pvs = solves[..., :num_probes] * coef * probe_vector_norms * ld
    ppv = probe_vectors * probe_vector_norms
    if precond_apply is not None:
        ppv = precond_apply(ppv)
    iqs = solves[..., num_probes:]
    neg = -iqs * inv_quad_grad[..., None, :]
    U = np.concatenate([pvs, neg], axis=-1)
    V = np.concatenate([ppv, iqs], axis=-1)
    return -2.0 * neg, U, V


def bilinear_derivative_kron(K1, K2, U, V):
    """K = K1 (x) K2: (dK1, dK2) = (sum_d U_d K2 V_d^T, sum_d U_d^T K1 V_d) with U_d, V_d the [n1, n2] views of the
    columns -- the generic autograd `_bilinear_derivative` (operators/_linear_operator.py:336-393) applied to the
    Kronecker matvec (operators/kronecker_product_linear_operator.py:34-45)."""
    n1, n2 = K1.shape[-1], K2.shape[-1]
    bs = U.shape[:-2]
    D = U.shape[-1]
    Ud = np.moveaxis(U.reshape(*bs, n1, n2, D), -1, -3)  # [*B, D, n1, n2]
    Vd = np.moveaxis(V.reshape(*bs, n1, n2, D), -1, -3)
    dK1 = (Ud @ K2[..., None, :, :] @ np.swapaxes(Vd, -1, -2)).sum(-3)
    dK2 = (np.swapaxes(Ud, -1, -2) @ K1[..., None, :, :] @ Vd).sum(-3)
    return dK1, dK2
