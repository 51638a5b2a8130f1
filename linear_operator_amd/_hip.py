"""ctypes binding of liblo_amd.so (the C ABI declared in include/lo_amd.h).

This is the ONLY compute backend of the package: there is no CPU or eager-PyTorch fallback.  If the
shared library is missing or a tensor is not a contiguous fp32 HIP tensor the call raises.
PyTorch is used for device memory (torch.empty workspaces through the caching allocator) and for the
current HIP stream only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "liblo_amd.so")
_lib = None

LO_OP_LOWRANK_DIAG, LO_OP_DENSE_DIAG, LO_OP_KRON_DIAG, LO_OP_CALLBACK, LO_OP_SUM = 0, 1, 2, 3, 4
LO_MAX_TERMS = 4
LO_DIAG_NONE, LO_DIAG_FULL, LO_DIAG_CONST = 0, 1, 2
ABI_VERSION = 15

LO_ERR_UNSUPPORTED = -4
LO_FUSED_OK, LO_FUSED_EARLY_STOP, LO_FUSED_CONTINUE, LO_FUSED_TIMEOUT = 0, 1, 2, 3
_ERR = {-1: "bad argument", -2: "HIP launch/runtime failure", -3: "workspace too small", -4: "unsupported shape"}

EXPORTS = [
    "lo_abi_version", "lo_target_arch",
    "lo_matvec_workspace_bytes", "lo_matvec_f32",
    "lo_cg_workspace_bytes", "lo_cg_solve_f32", "lo_cg_set_onchip", "lo_cg_plan_f32", "lo_cg_last_executed",
    "lo_resident_status_get", "lo_resident_inject_timeouts",
    "lo_solve_fused_supported", "lo_solve_fused_workspace_bytes", "lo_solve_fused_f32", "lo_solve_fused_perm",
    "lo_cg_f64_workspace_bytes", "lo_cg_solve_f64", "lo_minres_f64_workspace_bytes", "lo_minres_f64",
    "lo_pivoted_cholesky_workspace_bytes", "lo_pivoted_cholesky_f32",
    "lo_pivoted_cholesky_cb_workspace_bytes", "lo_pivoted_cholesky_cb_f32",
    "lo_pivoted_cholesky_f64_workspace_bytes", "lo_pivoted_cholesky_f64",
    "lo_pivoted_cholesky_cb_f64_workspace_bytes", "lo_pivoted_cholesky_cb_f64",
    "lo_precond_build_workspace_bytes", "lo_precond_build_f32", "lo_precond_build_strided_f32",
    "lo_precond_apply_workspace_bytes", "lo_precond_apply_f32",
    "lo_precond_root_form_workspace_bytes", "lo_precond_root_form_f32",
    "lo_precond_root_form_rs_workspace_bytes", "lo_precond_root_form_rs_f32", "lo_precond_eigform_f32",
    "lo_precond_kron_root_workspace_bytes", "lo_precond_kron_root_f32",
    "lo_lanczos_workspace_bytes", "lo_lanczos_tridiag_f32", "lo_lanczos_permute_f32",
    "lo_root_from_lanczos_f32", "lo_root_from_lanczos_native_f32",
    "lo_lanczos_f64_workspace_bytes", "lo_lanczos_tridiag_f64",
    "lo_tridiag_eigh_slq_workspace_bytes", "lo_tridiag_eigh_slq_f32",
    "lo_bilinear_dense_f32", "lo_bilinear_diag_f32", "lo_bilinear_root_workspace_bytes", "lo_bilinear_root_f32",
    "lo_bilinear_kron_workspace_bytes", "lo_bilinear_kron_f32", "lo_root_apply_add_f32",
    "lo_minres_workspace_bytes", "lo_minres_f32",
    "lo_probe_vectors_workspace_bytes", "lo_probe_vectors_f32", "lo_iql_backward_factors_f32",
    "lo_prof_enable", "lo_prof_report", "lo_hbm_triad_f32", "lo_hbm_copy_f32", "lo_hbm_stream_dev", "lo_peer_gather_set",
]


class HipExtensionError(RuntimeError):
    pass


STOP_REDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))


class OpDesc(C.Structure):
    pass


OpDesc._fields_ = [("kind", C.c_int32), ("diag_mode", C.c_int32), ("B", C.c_int64), ("N", C.c_int64), ("R", C.c_int64),
                   ("n2", C.c_int64), ("A0", C.c_void_p), ("A1", C.c_void_p), ("d", C.c_void_p),
                   ("nterms", C.c_int32), ("reserved", C.c_int32), ("terms", C.POINTER(OpDesc))]


class PrecondDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("ldq", C.c_int32), ("constant_diag", C.c_int32), ("reserved", C.c_int32),
                ("Q", C.c_void_p), ("dinv", C.c_void_p), ("F", C.c_void_p), ("EF", C.c_void_p), ("E", C.c_void_p),
                ("rf_ld", C.c_int32), ("generation", C.c_int32),
                ("kron_a", C.c_void_p), ("kron_b", C.c_void_p), ("kron_F", C.c_void_p), ("RS", C.c_void_p),
                ("RSD", C.c_void_p)]


class CgParams(C.Structure):
    _fields_ = [("c", C.c_int64), ("n_tridiag", C.c_int32), ("max_iter", C.c_int32), ("max_tridiag_iter", C.c_int32),
                ("floor_max_iter", C.c_int32), ("tolerance", C.c_float), ("eps", C.c_float),
                ("stop_updating_after", C.c_float), ("pad", C.c_float), ("stop_reduce", STOP_REDUCE_CB),
                ("stop_reduce_user", C.c_void_p)]


class CgInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("matvecs", C.c_int32), ("tolerance_reached", C.c_int32),
                ("nan_detected", C.c_int32), ("skipped", C.c_int32), ("last_tridiag_iter", C.c_int32),
                ("mean_residual", C.c_float), ("reserved", C.c_float)]


class CgPlan(C.Structure):
    """lo_cg_plan (include/lo_amd.h): the engine selection of lo_cg_solve_f32."""
    _fields_ = [(n, C.c_int32) for n in (
        "resident", "resident_iterations", "lockstep_cols", "lockstep_group", "serial_engine", "serial_group", "lean",
        "needs_q", "streaming_precond", "poll_chunk", "first_stop_iteration", "reserved", "rspace", "reserved2")]


class ResidentStatus(C.Structure):
    """lo_resident_status (include/lo_amd.h): the gate of the resident kernels."""
    _fields_ = [(n, C.c_int32) for n in ("user_disabled", "timeouts", "cooldown", "backoff", "rearms", "fused_timeouts")]


ENGINE_NAMES = {0: "none", 1: "gen1", 2: "gen2", 3: "root"}
STREAM_PRE_NAMES = {0: "none", 1: "two_pass", 2: "closure", 3: "fused_q", 4: "fused_kron", 5: "fused_cols",
                    6: "fused_cols_nopre"}


class FusedInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_int32), ("matvecs", C.c_int32),
                ("tolerance_reached", C.c_int32), ("nan_detected", C.c_int32), ("skipped", C.c_int32),
                ("rank", C.c_int32), ("mean_residual", C.c_float)]


class CgParamsF64(C.Structure):
    _fields_ = [("c", C.c_int64), ("n_tridiag", C.c_int32), ("max_iter", C.c_int32), ("max_tridiag_iter", C.c_int32),
                ("floor_max_iter", C.c_int32), ("tolerance", C.c_double), ("eps", C.c_double),
                ("stop_updating_after", C.c_double)]


class CgInfoF64(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("matvecs", C.c_int32), ("tolerance_reached", C.c_int32),
                ("nan_detected", C.c_int32), ("skipped", C.c_int32), ("last_tridiag_iter", C.c_int32),
                ("mean_residual", C.c_double)]


class MinresParamsF64(C.Structure):
    _fields_ = [("c", C.c_int64), ("n_shifts", C.c_int32), ("max_iter", C.c_int32), ("has_value", C.c_int32),
                ("shifts_per_member", C.c_int32), ("value", C.c_double), ("tolerance", C.c_double), ("eps", C.c_double)]


class MinresInfoF64(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("matvecs", C.c_int32), ("converged", C.c_int32), ("pad", C.c_int32),
                ("conv", C.c_double)]


class MinresParams(C.Structure):
    _fields_ = [("c", C.c_int64), ("n_shifts", C.c_int32), ("max_iter", C.c_int32), ("has_value", C.c_int32),
                ("shifts_per_member", C.c_int32), ("value", C.c_float), ("tolerance", C.c_float), ("eps", C.c_float),
                ("pad", C.c_float)]


class MinresInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("matvecs", C.c_int32), ("converged", C.c_int32), ("conv", C.c_float)]


ROWFETCH_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)
MATVEC_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load liblo_amd.so (after torch, so that both share one HIP runtime).  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipExtensionError(
            f"liblo_amd.so not found at {_LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C linear_operator_amd/csrc`). linear_operator_amd has no CPU fallback."
        )
    lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.lo_abi_version.restype = C.c_int
    lib.lo_target_arch.restype = C.c_char_p
    if lib.lo_abi_version() != ABI_VERSION:
        raise HipExtensionError(f"liblo_amd.so ABI {lib.lo_abi_version()} != binding {ABI_VERSION}; rebuild")
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise HipExtensionError(f"liblo_amd.so does not export {missing}; rebuild it")
    sz = C.c_size_t
    P = C.POINTER
    lib.lo_matvec_workspace_bytes.restype = sz
    lib.lo_matvec_workspace_bytes.argtypes = [P(OpDesc), C.c_int64]
    lib.lo_matvec_f32.restype = C.c_int
    lib.lo_matvec_f32.argtypes = [P(OpDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, sz, C.c_void_p]
    lib.lo_cg_workspace_bytes.restype = sz
    lib.lo_cg_workspace_bytes.argtypes = [P(OpDesc), P(PrecondDesc), P(CgParams)]
    lib.lo_cg_solve_f32.restype = C.c_int
    lib.lo_cg_solve_f32.argtypes = [P(OpDesc), MATVEC_CB, C.c_void_p, P(PrecondDesc), MATVEC_CB, C.c_void_p,
                                    P(CgParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz,
                                    P(CgInfo), C.c_void_p]
    lib.lo_root_from_lanczos_native_f32.restype = C.c_int
    lib.lo_root_from_lanczos_native_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lo_pivoted_cholesky_f64_workspace_bytes.restype = sz
    lib.lo_pivoted_cholesky_f64_workspace_bytes.argtypes = [P(OpDesc), C.c_int32]
    lib.lo_pivoted_cholesky_f64.restype = C.c_int
    lib.lo_pivoted_cholesky_f64.argtypes = [P(OpDesc), C.c_int32, C.c_double, C.c_void_p, C.c_void_p, P(C.c_int32),
                                            C.c_void_p, sz, C.c_void_p]
    lib.lo_pivoted_cholesky_cb_f64_workspace_bytes.restype = sz
    lib.lo_pivoted_cholesky_cb_f64_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32]
    lib.lo_pivoted_cholesky_cb_f64.restype = C.c_int
    lib.lo_pivoted_cholesky_cb_f64.argtypes = [C.c_int64, C.c_int64, C.c_void_p, ROWFETCH_CB, C.c_void_p, C.c_int32,
                                               C.c_double, C.c_void_p, C.c_void_p, P(C.c_int32), C.c_void_p, sz,
                                               C.c_void_p]
    lib.lo_cg_plan_f32.restype = C.c_int
    lib.lo_cg_plan_f32.argtypes = [P(OpDesc), P(PrecondDesc), C.c_int, C.c_int, P(CgParams), C.c_int, P(CgPlan)]
    lib.lo_cg_last_executed.restype = C.c_int
    lib.lo_cg_last_executed.argtypes = [P(CgPlan)]
    lib.lo_solve_fused_supported.restype = C.c_int
    lib.lo_solve_fused_supported.argtypes = [P(OpDesc), C.c_int32, P(CgParams)]
    lib.lo_solve_fused_workspace_bytes.restype = sz
    lib.lo_solve_fused_workspace_bytes.argtypes = [P(OpDesc), C.c_int32, P(CgParams)]
    lib.lo_solve_fused_f32.restype = C.c_int
    lib.lo_solve_fused_f32.argtypes = [P(OpDesc), C.c_int32, C.c_float, P(CgParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz,
                                       P(FusedInfo), C.c_void_p]
    lib.lo_solve_fused_perm.restype = C.c_int
    lib.lo_solve_fused_perm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    lib.lo_cg_f64_workspace_bytes.restype = sz
    lib.lo_cg_f64_workspace_bytes.argtypes = [C.c_int64, C.c_int64, P(CgParamsF64)]
    lib.lo_cg_solve_f64.restype = C.c_int
    lib.lo_cg_solve_f64.argtypes = [C.c_void_p, C.c_void_p, MATVEC_CB, C.c_void_p, MATVEC_CB, C.c_void_p,
                                    P(CgParamsF64), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, sz, P(CgInfoF64), C.c_void_p]
    lib.lo_lanczos_f64_workspace_bytes.restype = sz
    lib.lo_lanczos_f64_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32]
    lib.lo_lanczos_tridiag_f64.restype = C.c_int
    lib.lo_lanczos_tridiag_f64.argtypes = [C.c_void_p, C.c_void_p, MATVEC_CB, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                           P(C.c_int32), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lo_minres_f64_workspace_bytes.restype = sz
    lib.lo_minres_f64_workspace_bytes.argtypes = [C.c_int64, C.c_int64, P(MinresParamsF64)]
    lib.lo_minres_f64.restype = C.c_int
    lib.lo_minres_f64.argtypes = [C.c_void_p, C.c_void_p, MATVEC_CB, C.c_void_p, MATVEC_CB, C.c_void_p,
                                  P(MinresParamsF64), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, sz, P(MinresInfoF64), C.c_void_p]
    lib.lo_minres_workspace_bytes.restype = sz
    lib.lo_minres_workspace_bytes.argtypes = [P(OpDesc), P(PrecondDesc), P(MinresParams)]
    lib.lo_minres_f32.restype = C.c_int
    lib.lo_minres_f32.argtypes = [P(OpDesc), MATVEC_CB, C.c_void_p, P(PrecondDesc), MATVEC_CB, C.c_void_p,
                                  P(MinresParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, P(MinresInfo),
                                  C.c_void_p]
    lib.lo_pivoted_cholesky_workspace_bytes.restype = sz
    lib.lo_pivoted_cholesky_workspace_bytes.argtypes = [P(OpDesc), C.c_int32]
    lib.lo_pivoted_cholesky_f32.restype = C.c_int
    lib.lo_pivoted_cholesky_f32.argtypes = [P(OpDesc), C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                            P(C.c_int32), C.c_void_p, sz, C.c_void_p]
    lib.lo_pivoted_cholesky_cb_workspace_bytes.restype = sz
    lib.lo_pivoted_cholesky_cb_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32]
    lib.lo_pivoted_cholesky_cb_f32.restype = C.c_int
    lib.lo_pivoted_cholesky_cb_f32.argtypes = [C.c_int64, C.c_int64, C.c_void_p, ROWFETCH_CB, C.c_void_p, C.c_int32,
                                               C.c_float, C.c_void_p, C.c_void_p, P(C.c_int32), C.c_void_p, sz,
                                               C.c_void_p]
    lib.lo_precond_build_workspace_bytes.restype = sz
    lib.lo_precond_build_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32]
    lib.lo_precond_build_f32.restype = C.c_int
    lib.lo_precond_build_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p]
    lib.lo_precond_build_strided_f32.restype = C.c_int
    lib.lo_precond_build_strided_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32,
                                                 C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, sz, C.c_void_p]
    lib.lo_precond_root_form_workspace_bytes.restype = sz
    lib.lo_precond_root_form_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32]
    lib.lo_precond_root_form_f32.restype = C.c_int
    lib.lo_precond_root_form_f32.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                             C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                             C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, sz, C.c_void_p]
    lib.lo_precond_root_form_rs_workspace_bytes.restype = sz
    lib.lo_precond_root_form_rs_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32]
    lib.lo_precond_eigform_f32.restype = C.c_int
    lib.lo_precond_eigform_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.lo_precond_root_form_rs_f32.restype = C.c_int
    lib.lo_precond_root_form_rs_f32.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                                C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                                C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, sz, C.c_void_p]
    lib.lo_precond_kron_root_workspace_bytes.restype = sz
    lib.lo_precond_kron_root_workspace_bytes.argtypes = [C.c_int64]
    lib.lo_precond_kron_root_f32.restype = C.c_int
    lib.lo_precond_kron_root_f32.argtypes = [P(OpDesc), C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                             C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz,
                                             C.c_void_p]
    lib.lo_precond_apply_workspace_bytes.restype = sz
    lib.lo_precond_apply_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int64]
    lib.lo_precond_apply_f32.restype = C.c_int
    lib.lo_precond_apply_f32.argtypes = [P(PrecondDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_void_p, sz, C.c_void_p]
    lib.lo_root_from_lanczos_f32.restype = C.c_int
    lib.lo_root_from_lanczos_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lo_lanczos_permute_f32.restype = C.c_int
    lib.lo_lanczos_permute_f32.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.lo_lanczos_workspace_bytes.restype = sz
    lib.lo_lanczos_workspace_bytes.argtypes = [P(OpDesc), C.c_int64, C.c_int32]
    lib.lo_lanczos_tridiag_f32.restype = C.c_int
    lib.lo_lanczos_tridiag_f32.argtypes = [P(OpDesc), MATVEC_CB, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                           C.c_float, C.c_void_p, C.c_void_p, P(C.c_int32), C.c_void_p, sz,
                                           C.c_void_p]
    lib.lo_tridiag_eigh_slq_f32.restype = C.c_int
    lib.lo_tridiag_eigh_slq_workspace_bytes.restype = sz
    lib.lo_tridiag_eigh_slq_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.lo_tridiag_eigh_slq_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p]
    lib.lo_cg_set_onchip.restype = C.c_int
    lib.lo_cg_set_onchip.argtypes = [C.c_int]
    lib.lo_resident_status_get.restype = C.c_int
    lib.lo_resident_status_get.argtypes = [P(ResidentStatus)]
    lib.lo_resident_inject_timeouts.restype = C.c_int
    lib.lo_resident_inject_timeouts.argtypes = [C.c_int32]
    lib.lo_prof_enable.restype = C.c_int
    lib.lo_prof_enable.argtypes = [C.c_int]
    lib.lo_prof_report.restype = C.c_int
    lib.lo_prof_report.argtypes = [C.c_char_p, sz]
    lib.lo_bilinear_dense_f32.restype = C.c_int
    lib.lo_bilinear_dense_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.lo_bilinear_diag_f32.restype = C.c_int
    lib.lo_bilinear_diag_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                         C.c_void_p, sz, C.c_void_p]
    lib.lo_bilinear_root_workspace_bytes.restype = sz
    lib.lo_bilinear_root_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    lib.lo_bilinear_root_f32.restype = C.c_int
    lib.lo_bilinear_root_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p]
    lib.lo_bilinear_kron_workspace_bytes.restype = sz
    lib.lo_bilinear_kron_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    lib.lo_bilinear_kron_f32.restype = C.c_int
    lib.lo_bilinear_kron_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p]
    lib.lo_root_apply_add_f32.restype = C.c_int
    lib.lo_root_apply_add_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                          C.c_void_p]
    lib.lo_probe_vectors_workspace_bytes.restype = sz
    lib.lo_probe_vectors_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    lib.lo_probe_vectors_f32.restype = C.c_int
    lib.lo_probe_vectors_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p]
    lib.lo_iql_backward_factors_f32.restype = C.c_int
    lib.lo_iql_backward_factors_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_float, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lo_hbm_triad_f32.restype = C.c_int
    lib.lo_hbm_triad_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, sz, C.c_void_p]
    lib.lo_hbm_copy_f32.restype = C.c_int
    lib.lo_hbm_copy_f32.argtypes = [C.c_void_p, C.c_void_p, sz, C.c_void_p]
    lib.lo_peer_gather_set.restype = C.c_int
    lib.lo_peer_gather_set.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_longlong]
    lib.lo_hbm_stream_dev.restype = C.c_int
    lib.lo_hbm_stream_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, sz,
                                      C.c_void_p]
    _lib = lib
    return lib


def prof_enable(on: bool):
    load().lo_prof_enable(1 if on else 0)


def prof_report() -> dict:
    """{kernel_class: (launch_count, total_ms)} since the last report (HIP events on the launch stream)."""
    buf = C.create_string_buffer(1 << 16)
    load().lo_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


def hbm_stream_gbs(device, mode: str = "triad", n_floats: int = 1 << 28, reps: int = 10, unroll=None, nt=None) -> float:
    """Achievable HBM rate of this box in GB/s: `triad` a = b + s c (12 bytes per element), `copy` a = b (8 bytes),
    `read` (8 bytes), over n-float arrays (1 GiB each by default: far beyond the 256 MiB Infinity Cache).  `unroll` /
    `nt` select a variant of the sweep aid instead of the library's default shape."""
    import torch
    lib = load()
    a, b, c = (torch.empty(n_floats, dtype=torch.float32, device=device) for _ in range(3))
    b.fill_(1.0)
    c.fill_(2.0)
    st = stream_ptr(device)
    code = {"triad": 0, "copy": 1, "read": 2}[mode]
    nbytes = {"triad": 12, "copy": 8, "read": 8}[mode]

    def launch():
        if unroll is not None:
            return lib.lo_hbm_stream_dev(code, unroll, 1 if nt else 0, ptr(a), ptr(b), ptr(c), 0.5, n_floats, st)
        if mode == "triad":
            return lib.lo_hbm_triad_f32(ptr(a), ptr(b), ptr(c), 0.5, n_floats, st)
        if mode == "copy":
            return lib.lo_hbm_copy_f32(ptr(a), ptr(b), n_floats, st)
        return lib.lo_hbm_stream_dev(2, 4, 1, ptr(a), ptr(b), ptr(c), 0.5, n_floats, st)

    check(launch(), "lo_hbm_stream")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize(device)
    return nbytes * n_floats * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def hbm_triad_gbs(device, n_floats: int = 1 << 28, reps: int = 10) -> float:
    return hbm_stream_gbs(device, "triad", n_floats, reps)


def check(rc: int, what: str):
    if rc != 0:
        raise HipExtensionError(f"liblo_amd {what} failed: {_ERR.get(rc, rc)}")


def require_hip(*tensors: Optional[torch.Tensor], dtype=torch.float32):
    """Every tensor must be a HIP tensor of `dtype` (fp32 everywhere but the fp64 linear_cg entry); anything else is an
    error (no CPU path exists)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipExtensionError(
                "linear_operator_amd's iterative solvers run only on MI355X device tensors (got a CPU tensor); "
                "there is no CPU fallback -- move the operator / right-hand side to 'cuda'."
            )
        if t.dtype != dtype:
            raise HipExtensionError(
                f"liblo_amd kernels are fp32 (linear_cg on dense tensors / closures also fp64); got {t.dtype} where "
                f"{dtype} was expected")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """The current HIP stream of `device` as a raw handle (torch._C._cuda_getCurrentRawStream: 0.3 us instead of the 4 us
    of building a torch.cuda.Stream object -- this sits on every solve's critical path)."""
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is None:
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    if device is None:
        idx = torch.cuda.current_device()
    else:
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return C.c_void_p(raw(idx))


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class DevArray:
    """Borrowed device pointer exposed through __cuda_array_interface__ so that torch can view it
    (used to hand the C engine's buffers to Python matvec closures without a copy)."""

    def __init__(self, p: int, shape, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(p), False),
                                         "version": 2, "strides": None}


def as_tensor(p: int, shape, device, typestr: str = "<f4") -> torch.Tensor:
    return torch.as_tensor(DevArray(p, shape, typestr), device=device)
