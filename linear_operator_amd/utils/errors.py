"""Exception types of the hot path (reference: linear_operator/utils/errors.py:9-13)."""


class NanError(RuntimeError):
    pass


class NotPSDError(RuntimeError):
    pass
