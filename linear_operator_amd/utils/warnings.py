"""Warning types of the hot path (reference: linear_operator/utils/warnings.py:5-18)."""


class NumericalWarning(RuntimeWarning):
    """Numerical trouble that does not stop the computation (e.g. CG did not reach the tolerance)."""


class PerformanceWarning(RuntimeWarning):
    """A slow path was taken."""
