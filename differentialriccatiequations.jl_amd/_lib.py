"""ctypes binding of libdre_hip.so (include/dre_hip.h).  No torch types cross this boundary.

The library is built in-tree by `__graft_entry__.build()` / `make -C csrc`.  There is no CPU fallback:
if the shared object is missing the import fails loudly, and if no HIP device is usable
`dre_ctx_create` returns DRE_ERR_NODEVICE which is raised as `DREError`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRE_HIP_LIB") or os.path.join(_HERE, "libdre_hip.so")      # DRE_HIP_LIB: another build of the same library (A/B timing)


class DREError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdre_hip error {code}: {msg}")
        self.code = code


class AdiOptionsC(C.Structure):
    _fields_ = [
        ("maxiters", C.c_int32),
        ("reltol", C.c_double),
        ("abstol", C.c_double),
        ("ignore_initial_guess", C.c_int32),
        ("compression_interval", C.c_int32),
        ("compression", C.c_int32),
        ("shift_kind", C.c_int32),
        ("n_history", C.c_int32),
        ("nshifts", C.c_int32),
        ("shifts_re", C.POINTER(C.c_double)),
        ("shifts_im", C.POINTER(C.c_double)),
        ("compress_tolfac", C.c_double),
        ("compress_exact", C.c_int32),
        ("heuristic_kplus", C.c_int32),
        ("heuristic_kminus", C.c_int32),
        ("inner_solve", C.c_void_p),
        ("inner_user", C.c_void_p),
        ("shift_fn", C.c_void_p),
        ("shift_user", C.c_void_p),
    ]


# dre_block_solver_fn (include/dre_hip.h): int (*)(void* user, int n, int nrhs, double cA, double cE_re, double cE_im, const double* B, double* X_re, double* X_im)
# dre_shift_fn (include/dre_hip.h): int (*)(void* user, int restart, int n, int hist_cols, const double* hist, int ldh, int capacity, double* re, double* im, int* count)
SHIFT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int))
BLOCK_SOLVER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p)


_vp = C.c_void_p
_pvp = C.POINTER(C.c_void_p)
_pd = C.POINTER(C.c_double)
_pi64 = C.POINTER(C.c_int64)
_pi32 = C.POINTER(C.c_int32)
_pint = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol declared in include/dre_hip.h appears here
PROTOTYPES = {
    "dre_version": (C.c_int, []),
    "dre_ctx_create": (C.c_int, [C.c_int, _pvp]),
    "dre_ctx_destroy": (C.c_int, [_vp]),
    "dre_last_error": (C.c_char_p, [_vp]),
    "dre_ctx_sync": (C.c_int, [_vp]),
    "dre_ctx_info": (C.c_int, [_vp, _pi64]),
    "dre_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "dre_ctx_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "dre_prof_enable": (C.c_int, [_vp, C.c_int]),
    "dre_prof_reset": (C.c_int, [_vp]),
    "dre_prof_count": (C.c_int, [_vp, _pint]),
    "dre_prof_get": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, _pd, _pi64, _pd, _pd]),
    "dre_dense_upload": (C.c_int, [_vp, C.c_int, C.c_int, _pd, C.c_int, _pvp]),
    "dre_dense_create": (C.c_int, [_vp, C.c_int, C.c_int, _pvp]),
    "dre_dense_download": (C.c_int, [_vp, _vp, _pd, C.c_int]),
    "dre_dense_from_device": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, _pvp]),
    "dre_dense_to_device": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "dre_dense_shape": (C.c_int, [_vp, _pint, _pint]),
    "dre_dense_free": (C.c_int, [_vp, _vp]),
    "dre_pencil_create": (C.c_int, [_vp, C.c_int, _pi64, _pi64, _pd, _pi64, _pi64, _pd, C.c_int, C.c_int, _pvp]),
    "dre_pencil_create_host": (C.c_int, [C.c_int, _pi64, _pi64, _pd, _pi64, _pi64, _pd, C.c_int, C.c_int, _pvp]),
    "dre_pencil_free": (C.c_int, [_vp]),
    "dre_pencil_info": (C.c_int, [_vp, _pi64]),
    "dre_pencil_get_array": (C.c_int, [_vp, C.c_char_p, _pi64, C.c_int64, _pi64]),
    "dre_pencil_get_values": (C.c_int, [_vp, C.c_int, _pd, C.c_int64]),
    "dre_gemm": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, _vp, _vp, C.c_double, _vp]),
    "dre_spmm": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, C.c_double, _vp]),
    "dre_orthf": (C.c_int, [_vp, _vp, _pvp, _pvp]),
    "dre_sym_eig": (C.c_int, [_vp, _vp, C.c_double, _pvp, _pvp]),
    "dre_shift_factor": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _pvp]),
    "dre_shift_solve": (C.c_int, [_vp, _vp, _vp, _pvp, _pvp]),
    "dre_shift_solve_smw": (C.c_int, [_vp, _vp, C.c_double, _vp, _vp, _vp, _pvp, _pvp]),
    "dre_factor_growth": (C.c_int, [_vp, _vp, _pd]),
    "dre_factor_perturbed": (C.c_int, [_vp, _vp, _pi64]),
    "dre_factor_free": (C.c_int, [_vp, _vp]),
    "dre_ldlt_create": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, _pvp]),
    "dre_ldlt_zero": (C.c_int, [_vp, _vp, C.c_int, _pvp]),
    "dre_ldlt_free": (C.c_int, [_vp, _vp]),
    "dre_ldlt_info": (C.c_int, [_vp, _pint, _pint, _pint]),
    "dre_ldlt_add": (C.c_int, [_vp, _vp, _vp, _pvp]),
    "dre_ldlt_scale": (C.c_int, [_vp, _vp, C.c_double, _pvp]),
    "dre_ldlt_concatenate": (C.c_int, [_vp, _vp]),
    "dre_ldlt_compress": (C.c_int, [_vp, _vp]),
    "dre_ldlt_compress_tol": (C.c_int, [_vp, _vp, C.c_double]),
    "dre_ldlt_norm": (C.c_int, [_vp, _vp, _pd]),
    "dre_ldlt_canonicalize": (C.c_int, [_vp, _vp]),
    "dre_ldlt_destructure": (C.c_int, [_vp, _vp, _pd, _pd, C.c_int, _pd, C.c_int]),
    "dre_adi_default_options": (C.c_int, [C.POINTER(AdiOptionsC)]),
    "dre_gale_solve": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, C.POINTER(AdiOptionsC), _pvp]),
    "dre_ldlt_dot": (C.c_int, [_vp, _vp, _vp, _pd]),
    "dre_ldlt_compress_fast": (C.c_int, [_vp, _vp]),
    "dre_gare_residual": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp, C.c_double, _pvp]),
    "dre_ldlt_feedback": (C.c_int, [_vp, _vp, _vp, _vp, _pvp]),
    "dre_gale_apply": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _pvp]),
    "dre_adi_init": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, C.POINTER(AdiOptionsC), _pvp]),
    "dre_adi_step": (C.c_int, [_vp, _vp]),
    "dre_adi_solve": (C.c_int, [_vp, _vp]),
    "dre_adi_isdone": (C.c_int, [_vp, _pint]),
    "dre_adi_state": (C.c_int, [_vp, _pi64, _pd, _pd]),
    "dre_adi_finish": (C.c_int, [_vp, _vp, _pvp]),
    "dre_comm_unique_id": (C.c_int, [_vp, _vp]),
    "dre_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "dre_ctx_set_orthf": (C.c_int, [_vp, _vp, _vp]),
    "dre_comm_init_host": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "dre_comm_free": (C.c_int, [_vp]),
    "dre_comm_info": (C.c_int, [_vp, _pi64]),
    "dre_comm_allgather": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "dre_comm_allreduce_sum": (C.c_int, [_vp, _vp, C.c_size_t]),
    "dre_adi_snapshot": (C.c_int, [_vp, _vp, _pvp, _pvp]),
    "dre_adi_shifts": (C.c_int, [_vp, C.c_int64, _pi64, _pd, _pd]),
    "dre_adi_free": (C.c_int, [_vp]),
    "dre_heuristic_ritz": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, C.c_int, C.c_int, _pd, _pd, _pd, _pd]),
    "dre_gale_residual": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _pvp]),
    "dre_adi_result_info": (C.c_int, [_vp, _pi64, _pd]),
    "dre_adi_result_history": (C.c_int, [_vp, _pd, _pi32, _pd, _pd]),
    "dre_adi_result_take_x": (C.c_int, [_vp, _pvp]),
    "dre_adi_result_take_residual": (C.c_int, [_vp, _pvp]),
    "dre_adi_result_free": (C.c_int, [_vp]),
    "dre_gdre_solve": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(AdiOptionsC), _pvp]),
    "dre_gdre_result_info": (C.c_int, [_vp, _pi64]),
    "dre_gdre_result_times": (C.c_int, [_vp, _pd]),
    "dre_gdre_result_K": (C.c_int, [_vp, _vp, C.c_int, _pd, C.c_int]),
    "dre_gdre_result_K_device": (C.c_int, [_vp, _vp, _vp]),
    "dre_gdre_result_K_all": (C.c_int, [_vp, _vp, _pd]),
    "dre_gdre_result_X": (C.c_int, [_vp, C.c_int, _pvp]),
    "dre_gdre_result_gale": (C.c_int, [_vp, C.c_int, _pi64, _pd]),
    "dre_gdre_result_gale_history": (C.c_int, [_vp, C.c_int, _pi64, _pd, _pi32, _pd, _pd]),
    "dre_gdre_result_gales_all": (C.c_int, [_vp, _pi64, _pd, _pd, _pi32, _pd, _pd]),
    "dre_gdre_result_free": (C.c_int, [_vp]),
    "dre_host_eigvals": (C.c_int, [C.c_int, _pd, _pd, _pd]),
    "dre_host_gen_eigvals": (C.c_int, [C.c_int, _pd, _pd, _pd, _pd]),
    "dre_host_svd_left": (C.c_int, [C.c_int, C.c_int, _pd, _pd, _pd]),
}

_lib = None


def load():
    """Load libdre_hip.so and attach prototypes; raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_ptr, rc):
    if rc != 0:
        msg = load().dre_last_error(ctx_ptr)
        raise DREError(rc, msg.decode() if msg else "")
    return rc
