"""Process-global flags and values of the hot path, with the reference's names, defaults and
context-manager protocol (linear_operator/settings.py: `_feature_flag` :58-93, `_value_context` :96-118,
`_dtype_value_context` :9-55).  GPyTorch subclasses these, so the class-attribute mechanics are kept:
values are read at call time inside the solver entry points, state lives on the class, `with flag(v):`
swaps it and restores it on exit.  Only the knobs the solve / logdet path reads are defined.
"""
from __future__ import annotations

import logging

import torch


class _feature_flag:
    _default = False
    _state = None

    @classmethod
    def is_default(cls):
        return cls._state is None

    @classmethod
    def on(cls):
        return cls._default if cls._state is None else cls._state

    @classmethod
    def off(cls):
        return not cls.on()

    @classmethod
    def _set_state(cls, state):
        cls._state = state

    def __init__(self, state=True):
        self.prev = type(self)._state
        self.state = state

    def __enter__(self):
        type(self)._set_state(self.state)

    def __exit__(self, *exc):
        type(self)._set_state(self.prev)
        return False


class _value_context:
    _global_value = None

    @classmethod
    def value(cls):
        return cls._global_value

    @classmethod
    def _set_value(cls, value):
        cls._global_value = value

    def __init__(self, value):
        self._orig_value = type(self).value()
        self._instance_value = value

    def __enter__(self):
        type(self)._set_value(self._instance_value)

    def __exit__(self, *exc):
        type(self)._set_value(self._orig_value)
        return False


class _dtype_value_context:
    _global_float_value = None
    _global_double_value = None
    _global_half_value = None

    @classmethod
    def value(cls, dtype):
        if torch.is_tensor(dtype):
            dtype = dtype.dtype
        table = {torch.float: cls._global_float_value, torch.double: cls._global_double_value,
                 torch.half: cls._global_half_value}
        if dtype not in table:
            raise RuntimeError(f"Unsupported dtype for {cls.__name__}.")
        return table[dtype]

    @classmethod
    def _set_value(cls, float_value, double_value, half_value):
        if float_value is not None:
            cls._global_float_value = float_value
        if double_value is not None:
            cls._global_double_value = double_value
        if half_value is not None:
            cls._global_half_value = half_value

    def __init__(self, float_value=None, double_value=None, half_value=None):
        cls = type(self)
        self._orig = (cls.value(torch.float), cls.value(torch.double), cls.value(torch.half))
        self._inst = (float_value, double_value, half_value)

    def __enter__(self):
        type(self)._set_value(*self._inst)

    def __exit__(self, *exc):
        type(self)._set_value(*self._orig)
        return False


# ---- dispatch: Cholesky vs CG (functions/_solve.py:17, _inv_quad.py:11-16, _linear_operator.py:1713) ----
class _fast_covar_root_decomposition(_feature_flag):
    _default = True


class _fast_log_prob(_feature_flag):
    _default = True


class _fast_solves(_feature_flag):
    _default = True


class fast_computations:
    """`with fast_computations(log_prob=False, solves=False)` -> exact Cholesky paths (settings.py:278-354)."""

    covar_root_decomposition = _fast_covar_root_decomposition
    log_prob = _fast_log_prob
    solves = _fast_solves

    def __init__(self, covar_root_decomposition=True, log_prob=True, solves=True):
        self.covar_root_decomposition = _fast_covar_root_decomposition(covar_root_decomposition)
        self.log_prob = _fast_log_prob(log_prob)
        self.solves = _fast_solves(solves)

    def __enter__(self):
        self.covar_root_decomposition.__enter__()
        self.log_prob.__enter__()
        self.solves.__enter__()

    def __exit__(self, *exc):
        self.covar_root_decomposition.__exit__()
        self.log_prob.__exit__()
        self.solves.__exit__()
        return False


class cholesky_jitter(_dtype_value_context):
    """Jitter added by psd_safe_cholesky on failure (settings.py:194-203)."""

    _global_float_value = 1e-6
    _global_double_value = 1e-8


class cholesky_max_tries(_value_context):
    _global_value = 3


class cg_tolerance(_value_context):
    """Mean relative residual at which linear_cg stops (settings.py:216-223; default 1)."""

    _global_value = 1


class debug(_feature_flag):
    _default = True


class deterministic_probes(_feature_flag):
    """The same base samples for the log-determinant probes of every call (reference settings.py:245-262, deprecated
    there): InvQuadLogdet colours them with a root of the preconditioner.  The base samples are cached HERE, in global
    scope, and dropped whenever the flag's state changes -- one model per context, as in the reference."""

    _default = False
    probe_vectors = None

    @classmethod
    def _set_state(cls, state):
        super()._set_state(state)
        cls.probe_vectors = None


class max_cg_iterations(_value_context):
    _global_value = 1000


class max_cholesky_size(_value_context):
    """N <= this -> Cholesky instead of CG (settings.py:394-402)."""

    _global_value = 800


class max_lanczos_quadrature_iterations(_value_context):
    _global_value = 20


class max_preconditioner_size(_value_context):
    """Rank of the pivoted-Cholesky preconditioner (settings.py:417-425)."""

    _global_value = 15


class max_root_decomposition_size(_value_context):
    _global_value = 100


class memory_efficient(_feature_flag):
    _default = False


class minres_tolerance(_value_context):
    """Relative update-norm tolerance that stops MINRES (settings.py:464-471); default 1e-4."""

    _global_value = 1e-4


class num_contour_quadrature(_value_context):
    """Quadrature points of contour integral quadrature (settings.py:474-481); default 15."""

    _global_value = 15


class ciq_samples(_feature_flag):
    """Draw samples with contour integral quadrature (settings.py:226-241); default off."""

    _default = False


class min_preconditioning_size(_value_context):
    """No preconditioner below this N (settings.py:453-461)."""

    _global_value = 2000


class num_trace_samples(_value_context):
    """Probe vectors for the SLQ logdet (settings.py:484-493)."""

    _global_value = 10


class preconditioner_tolerance(_value_context):
    _global_value = 1e-3


class skip_logdet_forward(_feature_flag):
    _default = False


class terminate_cg_by_size(_feature_flag):
    _default = False


class trace_mode(_feature_flag):
    _default = False


class tridiagonal_jitter(_value_context):
    _global_value = 1e-6


class _linalg_dtype_symeig(_value_context):
    _global_value = torch.double


class _linalg_dtype_cholesky(_value_context):
    _global_value = torch.double


class linalg_dtypes:
    """Precision of the dense symeig / Cholesky plumbing calls (settings.py:356-380); default torch.double."""

    def __init__(self, default=torch.double, symeig=None, cholesky=None):
        self.symeig = _linalg_dtype_symeig(default if symeig is None else symeig)
        self.cholesky = _linalg_dtype_cholesky(default if cholesky is None else cholesky)

    def __enter__(self):
        self.symeig.__enter__()
        self.cholesky.__enter__()

    def __exit__(self, *args):
        self.symeig.__exit__()
        self.cholesky.__exit__()
        return False


class verbose_linalg(_feature_flag):
    """Debug logging of every linear-algebra call (settings.py:587-605)."""

    _default = False
    logger = logging.getLogger("LinAlg (Verbose)")
    if not logger.handlers:
        _h = logging.StreamHandler()
        _h.setFormatter(logging.Formatter("%(name)s - %(levelname)s - %(message)s"))
        logger.addHandler(_h)
    logger.setLevel(logging.DEBUG)


__all__ = [
    "fast_computations", "linalg_dtypes", "cholesky_jitter", "cholesky_max_tries", "cg_tolerance", "debug", "deterministic_probes",
    "max_cg_iterations", "max_cholesky_size", "max_lanczos_quadrature_iterations", "max_preconditioner_size",
    "max_root_decomposition_size", "memory_efficient", "min_preconditioning_size", "minres_tolerance",
    "num_contour_quadrature", "ciq_samples", "num_trace_samples",
    "preconditioner_tolerance", "skip_logdet_forward", "terminate_cg_by_size", "trace_mode", "tridiagonal_jitter",
    "verbose_linalg",
]
