"""ctypes binding of ``libsd_downscale.so`` (the C ABI declared in ``include/sd_downscale.h``).

There is deliberately **no CPU fallback**: if the shared library is missing or a call fails the
error is raised to the caller.  Build the library with ``python __graft_entry__.py`` (or
``make -C scikit-downscale_amd``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SD_DOWNSCALE_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libsd_downscale.so"))

SD_OK = 0
BCSD_TAS, BCSD_PR = 0, 1
BCSD_RETURN_ANOMS, BCSD_QM_DETREND = 1, 2  # bits of the `return_anoms` argument of the fit entry points
CELL_OK, CELL_MASKED, CELL_NONFINITE, CELL_BAD_CLIMO, CELL_ONE_CLASS = 0, 1, 2, 3, 4
ANALOG_BEST, ANALOG_SAMPLE, ANALOG_WEIGHT, ANALOG_MEAN = 0, 1, 2, 3
QM_REGRESSOR, QM_EDCDF_DIFFERENCE, QM_EDCDF_RATIO = 0, 1, 2
CUNNANE_FORWARD, CUNNANE_INVERSE = 0, 1
EXTRAP_CODES = {None: 0, "1to1": 0, "min": 1, "max": 2, "both": 3}  # SD_EXTRAP_* (CunnaneTransformer: '1to1' clamps like None)
QM_EXTRAP_CODES = {None: 0, "min": 1, "max": 2, "both": 3, "1to1": 4}  # regressors (sd_qm_predict)
SYNTH_GAUSS, SYNTH_PRECIP = 0, 1

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_dbl = C.c_double

# name -> (argtypes); every function returns int unless noted
SIGNATURES = {
    "sd_version": [],
    "sd_device_count": [C.POINTER(_int)],
    "sd_ctx_create": [_int, C.POINTER(_p)],
    "sd_ctx_destroy": [_p],
    "sd_ctx_synchronize": [_p],
    "sd_ctx_release_cached": [_p],
    "sd_ctx_device_info": [_p, C.c_char_p, C.c_size_t, C.POINTER(_int), C.POINTER(_i64)],
    "sd_dev_alloc": [_p, C.c_size_t, C.POINTER(_p)],
    "sd_dev_free": [_p, _p],
    "sd_memcpy_h2d": [_p, _p, _p, C.c_size_t],
    "sd_memcpy_d2h": [_p, _p, _p, C.c_size_t],
    "sd_memcpy_d2d": [_p, _p, _p, C.c_size_t],
    "sd_convert_f32_to_f64_dev": [_p, _p, _i64, _p],
    "sd_convert_f64_to_f32_dev": [_p, _p, _i64, _p],
    "sd_timer_start": [_p],
    "sd_timer_stop": [_p, C.POINTER(C.c_float)],
    "sd_prof_enable": [_p, _int],
    "sd_prof_reset": [_p],
    "sd_prof_query": [_p, C.c_char_p, C.POINTER(_dbl), C.POINTER(_i64)],
    "sd_prof_names": [_p, C.c_char_p, C.c_size_t],
    "sd_synth_fill": [_p, _p, _i64, _i64, _i64, _i64, _i64, _int, C.c_uint64, C.c_uint32, _p, _dbl, _dbl, _dbl,
                      C.c_int32, _dbl],
    "sd_bcsd_fit": [_p, _int, _p, _p, _p, _int, _i64, _i64, _int, C.POINTER(_p)],
    "sd_bcsd_fit_dev": [_p, _int, _p, _p, _i64, _p, _int, _i64, _i64, _int, C.POINTER(_p)],
    "sd_bcsd_predict": [_p, _p, _p, _p, _i64, _p, _p],
    "sd_bcsd_predict_dev": [_p, _p, _p, _i64, _p, _i64, _p, _i64, _p],
    "sd_bcsd_fit_groups": [_p, _int, _p, _p, _p, _p, _int, _i64, _i64, _int, C.POINTER(_p)],
    "sd_bcsd_fit_groups_dev": [_p, _int, _p, _p, _i64, _p, _p, _int, _i64, _i64, _int, C.POINTER(_p)],
    "sd_bcsd_predict_trend": [_p, _p, _p, _p, _p, _int, _i64, _p, _p],
    "sd_bcsd_predict_trend_dev": [_p, _p, _p, _i64, _p, _p, _int, _i64, _p, _i64, _p],
    "sd_bcsd_fit_predict_dev": [_p, _int, _p, _p, _i64, _p, _int, _i64, _i64, _int, _p, _i64, _p, _i64, _p, _i64, _p],
    "sd_bcsd_state_info": [_p, C.POINTER(_int), C.POINTER(_int), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int)],
    "sd_bcsd_state_status": [_p, _p],
    "sd_bcsd_state_export": [_p, _p, _p, _p, _p, _p],
    "sd_bcsd_state_import": [_p, _int, _int, _i64, _i64, _int, _p, _p, _p, _p, _p, C.POINTER(_p)],
    "sd_bcsd_state_set_tails": [_p, _int, _int],
    "sd_bcsd_state_get_trend": [_p, _p],
    "sd_bcsd_state_set_trend": [_p, _p],
    "sd_bcsd_state_destroy": [_p],
    "sd_analog_fit": [_p, _p, _p, _i64, _int, _i64, C.POINTER(_p)],
    "sd_analog_fit_dev": [_p, _p, _p, _i64, _i64, _int, _i64, C.POINTER(_p)],
    "sd_analog_predict": [_p, _p, _p, _i64, _int, _int, _int, _dbl, _p, _p, _p, _p, _p],
    "sd_analog_predict_dev": [_p, _p, _p, _i64, _i64, _int, _int, _int, _dbl, _p, _p, _i64, _p, _p, _p],
    "sd_analog_fit_predict": [_p, _p, _p, _i64, _int, _i64, _p, _i64, _int, _int, _int, _dbl, _p, _p],
    "sd_analog_fit_predict_dev": [_p, _p, _p, _i64, _i64, _int, _i64, _p, _i64, _i64, _int, _int, _int, _dbl, _p, _i64, _p],
    "sd_analogreg_predict": [_p, _p, _p, _i64, _int, _int, _dbl, _p, _p],
    "sd_analogreg_predict_dev": [_p, _p, _p, _i64, _i64, _int, _int, _dbl, _p, _i64, _p],
    "sd_analog_state_info": [_p, C.POINTER(_i64), C.POINTER(_int), C.POINTER(_i64)],
    "sd_analog_state_destroy": [_p],
    "sd_linreg_fit": [_p, _p, _p, _i64, _int, _i64, _int, _dbl, C.POINTER(_p)],
    "sd_linreg_fit_dev": [_p, _p, _p, _i64, _i64, _int, _i64, _int, _dbl, C.POINTER(_p)],
    "sd_linreg_predict": [_p, _p, _p, _i64, _p, _p],
    "sd_linreg_predict_dev": [_p, _p, _p, _i64, _i64, _p, _i64, _p],
    "sd_linreg_state_info": [_p, C.POINTER(_i64), C.POINTER(_int), C.POINTER(_i64)],
    "sd_linreg_state_export": [_p, _p, _p, _p, _p, _p, _p],
    "sd_linreg_state_import": [_p, _i64, _int, _i64, _p, _p, _p, _p, _p, _p, C.POINTER(_p)],
    "sd_linreg_state_destroy": [_p],
    "sd_qm_fit": [_p, _p, _p, _i64, _i64, C.POINTER(_p)],
    "sd_qm_fit_dev": [_p, _p, _p, _i64, _i64, _i64, C.POINTER(_p)],
    "sd_qm_predict": [_p, _p, _int, _int, _int, _p, _i64, _p, _p],
    "sd_qm_predict_dev": [_p, _p, _int, _int, _int, _p, _i64, _i64, _p, _i64, _p],
    "sd_qm_cunnane": [_p, _p, _int, _int, _int, _p, _i64, _p, _p],
    "sd_qm_cunnane_dev": [_p, _p, _int, _int, _int, _p, _i64, _i64, _p, _i64, _p],
    "sd_qm_state_info": [_p, C.POINTER(_i64), C.POINTER(_i64)],
    "sd_qm_state_export": [_p, _p, _p, _p],
    "sd_qm_state_destroy": [_p],
    "sd_comm_unique_id": [_p],
    "sd_comm_create": [_p, _p, _int, _int, C.POINTER(_p)],
    "sd_comm_destroy": [_p],
    "sd_comm_info": [_p, C.POINTER(_int), C.POINTER(_int)],
    "sd_comm_rccl_info": [_p, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)],
    "sd_comm_barrier": [_p],
    "sd_comm_allreduce_max": [_p, _dbl, C.POINTER(_dbl)],
    "sd_comm_gather_field": [_p, _p, _i64, _p, _p, _int, _int],
    "sd_comm_wait": [_p],
}

_libs = {}
QT_TAIL_LOWER, QT_TAIL_UPPER = 1, 2  # sd_bcsd_state_set_tails
ABI_VERSION = 103  # include/sd_downscale.h: SD_VERSION


class EngineError(RuntimeError):
    """A call into the HIP engine failed (carries sd_last_error())."""


DEV_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libsd_downscale_dev.so")  # `make dev`: A/B switches (tests, tools)


def load(path=None):
    """Load (once per path) and return the ctypes library; raises if it has not been built."""
    path = LIB_PATH if path is None else path
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise EngineError(
            f"HIP engine library not found at {path}: build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _int
    lib.sd_last_error.argtypes = []
    lib.sd_last_error.restype = C.c_char_p
    got = lib.sd_version()
    if got != ABI_VERSION:  # a stale build would be called with misaligned arguments
        raise EngineError(f"{path} implements version {got} of the C ABI, this package binds version {ABI_VERSION} "
                          "(include/sd_downscale.h: SD_VERSION): rebuild it with `python __graft_entry__.py`")
    _libs[path] = lib
    return lib


def check(rc):
    if rc != SD_OK:
        msg = load().sd_last_error().decode("utf-8", "replace")
        if rc == 1:
            raise ValueError(f"sd_downscale: {msg}")
        if rc == 3:
            raise NotImplementedError(f"sd_downscale: {msg}")
        if rc == 4:
            raise MemoryError(f"sd_downscale: {msg}")
        raise EngineError(f"sd_downscale (code {rc}): {msg}")


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
