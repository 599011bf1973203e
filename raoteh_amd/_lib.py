"""
ctypes binding of libraoteh_hip.so (the C ABI in include/raoteh_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be
loaded, importing the compute modules raises ``HipLibraryError``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_int, c_int32,
                    c_int64, c_ubyte, c_void_p)

__all__ = ['HipLibraryError', 'RaotehHipError', 'lib', 'check', 'LIB_PATH']

LIB_PATH = os.environ.get(
    'RAOTEH_HIP_LIB',
    os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libraoteh_hip.so'))

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_HIP = -2
RT_ERR_UNSUPPORTED = -3
RT_ERR_NOMEM = -4
RT_ERR_RCCL = -5
RT_ERR_SINGULAR = -6
RT_ERR_ZERO_PROB = -7

RT_OBS_DENSE, RT_OBS_STATE, RT_OBS_MASK = 0, 1, 2
RT_K_EXPM, RT_K_PRUNE, RT_K_REDUCE, RT_K_COMBINE = 0, 1, 2, 3
RT_SITE_ZERO_PROB = 1
RT_SITE_NEGATIVE = 4


class HipLibraryError(ImportError):
    """libraoteh_hip.so is missing / not loadable (run __graft_entry__.build())."""


class RaotehHipError(RuntimeError):
    """A C-ABI call returned a negative RT_ERR_* code."""

    def __init__(self, code, message):
        RuntimeError.__init__(self, '%s (code %d)' % (message, code))
        self.code = code


_p_i64 = POINTER(c_int64)
_p_f64 = POINTER(c_double)
_p_i32 = POINTER(c_int32)

# name -> (restype, argtypes); every symbol declared in include/raoteh_hip.h
SIGNATURES = {
    'rt_version': (c_int, []),
    'rt_last_error': (c_char_p, []),
    'rt_device_count': (c_int, [POINTER(c_int)]),
    'rt_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
    'rt_ctx_destroy': (c_int, [c_void_p]),
    'rt_ctx_sync': (c_int, [c_void_p]),
    'rt_ctx_set_timing': (c_int, [c_void_p, c_int]),
    'rt_ctx_reset_timing': (c_int, [c_void_p]),
    'rt_ctx_kernel_time': (c_int, [c_void_p, c_int, POINTER(c_double),
                                   POINTER(c_int64), POINTER(c_char_p)]),
    'rt_expm': (c_int, [c_void_p, c_int64, c_int64, _p_f64, c_int64, _p_i64,
                        _p_f64, _p_f64, _p_i32]),
    'rt_lb_transition_matrix': (c_int, [c_void_p, c_int64, c_int64, _p_f64, _p_f64, _p_f64]),
    'rt_tmjp_get_inhomogeneous_mjp': (c_int, [c_int64, _p_i64, _p_i64, _p_i64, c_int64, _p_i64,
                                              _p_f64, c_double, c_double, c_int64, _p_i64,
                                              _p_f64]),
    'rt_mcy_get_node_to_pset': (c_int, [c_void_p, c_int64, c_int64, _p_i64, _p_i64, _p_i64,
                                        _p_i64, _p_i64]),
    'rt_get_node_to_set': (c_int, [c_void_p, c_int64, c_int64, _p_i64, _p_i64, _p_i64, _p_i64,
                                   _p_i64, _p_i64]),
    'rt_mcy_esd_get_node_to_pset': (c_int, [c_void_p, c_int64, c_int64, c_int64,
                                            _p_i64, _p_i64, _p_f64, _p_i64]),
    'rt_esd_get_node_to_set': (c_int, [c_void_p, c_int64, c_int64, c_int64,
                                       _p_i64, _p_i64, _p_f64, _p_i64]),
    'rt_mcy_esd_get_node_to_pmap': (c_int, [c_void_p, c_int64, c_int64, c_int64,
                                            _p_i64, _p_i64, _p_f64, _p_i64,
                                            _p_f64, _p_f64]),
    'rt_mcy_esd_passes': (c_int, [c_void_p, c_int64, c_int64, c_int64,
                                  _p_i64, _p_i64, _p_f64, _p_i64,
                                  _p_f64, _p_f64]),
    'rt_mc0_esd_get_node_to_distn': (c_int, [c_void_p, c_int64, c_int64, c_int64,
                                             _p_i64, _p_i64, _p_f64, _p_f64,
                                             _p_f64, _p_f64, _p_i32]),
    'rt_mc0_esd_get_joint_endpoint_distn': (c_int, [c_void_p, c_int64, c_int64,
                                                    c_int64, _p_i64, _p_i64,
                                                    _p_f64, _p_f64, _p_f64,
                                                    _p_f64]),
    'rt_mjp_esd_expectation_weights': (c_int, [c_void_p, c_int64, c_int64, c_int64,
                                               _p_i64, _p_i64, _p_f64, _p_f64, _p_i64,
                                               _p_f64, _p_f64, _p_i32]),
    'rt_mjp_esd_expectation_weights_obs': (c_int, [c_void_p, c_int64, c_int64, c_int64,
                                                   _p_i64, _p_i64, _p_f64, _p_f64, c_int64,
                                                   _p_i64, c_int, c_void_p, _p_f64, _p_f64,
                                                   _p_i32]),
    'rt_mjp_frechet_statistics': (c_int, [c_void_p, c_int64, c_int64, _p_f64, c_int64, _p_i64,
                                          _p_f64, _p_f64, _p_f64, _p_f64]),
    'rt_model_create': (c_int, [c_void_p, c_int64, c_int64, _p_i64, _p_i64,
                                POINTER(c_void_p)]),
    'rt_model_destroy': (c_int, [c_void_p]),
    'rt_model_set_rates': (c_int, [c_void_p, _p_f64, c_int64, _p_i64, _p_f64]),
    'rt_model_set_rates_spectral': (c_int, [c_void_p, _p_f64, _p_f64, _p_f64, _p_f64, _p_f64]),
    'rt_expm_spectral': (c_int, [c_void_p, c_int64, c_int64, _p_f64, _p_f64, _p_f64, _p_f64,
                                 _p_f64, _p_f64]),
    'rt_model_recompute_transitions': (c_int, [c_void_p]),
    'rt_model_set_transitions': (c_int, [c_void_p, _p_f64]),
    'rt_model_get_transitions': (c_int, [c_void_p, _p_f64]),
    'rt_model_get_expm_info': (c_int, [c_void_p, _p_i32]),
    'rt_model_set_root_distn': (c_int, [c_void_p, _p_f64]),
    'rt_model_schedule_depth': (c_int, [c_void_p]),
    'rt_model_get_schedule': (c_int, [c_void_p, _p_i32, c_int64, _p_i64]),
    'rt_build_schedule': (c_int, [c_int64, _p_i64, _p_i64, _p_i32, _p_i32]),
    'rt_set_option': (c_int, [c_char_p, c_int64]),
    'rt_ctx_set_option': (c_int, [c_void_p, c_char_p, c_int64]),
    'rt_sites_jit_compile_seconds': (c_double, [c_void_p]),
    'rt_sites_kernel_name': (c_char_p, [c_void_p]),
    'rt_debug_jit_global': (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    'rt_jit_source': (c_int, [c_int64, _p_i64, _p_i64, c_int64, c_int64, _p_i64, c_int64,
                              c_char_p, c_int64]),
    'rt_sites_create': (c_int, [c_void_p, c_int64, c_int, c_int64, _p_i64,
                                c_void_p, POINTER(c_void_p)]),
    'rt_sites_clone': (c_int, [c_void_p, POINTER(c_void_p)]),
    'rt_sites_jit_wait': (c_int, [c_void_p]),
    'rt_jit_wait_all': (c_int, []),
    'rt_expect_step': (c_int, [c_void_p, c_void_p, c_int, _p_f64, _p_f64, _p_f64, _p_i32]),
    'rt_sites_set_weights': (c_int, [c_void_p, _p_f64]),
    'rt_sites_destroy': (c_int, [c_void_p]),
    'rt_sites_device_bytes': (c_int64, [c_void_p]),
    'rt_prune': (c_int, [c_void_p, c_void_p]),
    'rt_step': (c_int, [c_void_p, c_void_p, c_int]),
    'rt_sites_get_logliks': (c_int, [c_void_p, _p_f64, _p_i32]),
    'rt_sites_get_totals': (c_int, [c_void_p, _p_f64]),
    'rt_comm_available': (c_int, []),
    'rt_comm_unique_id': (c_int, [POINTER(c_ubyte)]),
    'rt_comm_init': (c_int, [c_void_p, c_int, c_int, POINTER(c_ubyte)]),
    'rt_comm_destroy': (c_int, [c_void_p]),
    'rt_allreduce_totals': (c_int, [c_void_p, c_void_p]),
    'rt_allreduce_totals_group': (c_int, [c_void_p, POINTER(c_void_p), c_int64]),
    'rt_forest_passes': (c_int, [c_void_p, c_int64, c_int64, _p_i64, _p_i64, _p_i64, _p_f64,
                                 POINTER(ctypes.c_uint64), _p_f64]),
    'rt_forest_resample_states': (c_int, [c_void_p, c_int64, c_int64, _p_i64, _p_i64, _p_i64,
                                          _p_f64, _p_f64, POINTER(ctypes.c_uint64),
                                          ctypes.c_uint64, ctypes.c_uint64, _p_i32, _p_i32,
                                          _p_f64]),
    'rt_forest_resample_states_parents': (c_int, [c_void_p, c_int64, c_int64, _p_i64, _p_i32,
                                                  _p_f64, _p_f64, POINTER(ctypes.c_uint64),
                                                  ctypes.c_uint64, ctypes.c_uint64, _p_i32,
                                                  _p_i32]),
    'rt_chains_create': (c_int, [c_void_p, c_int64, _p_i32, _p_f64, c_int64, _p_f64, _p_f64,
                                 _p_f64, c_int64, POINTER(ctypes.c_uint64), ctypes.c_uint64,
                                 POINTER(c_void_p)]),
    'rt_chains_sweep': (c_int, [c_void_p, c_int64]),
    'rt_chains_get_sizes': (c_int, [c_void_p, _p_i64, _p_i64, _p_i64]),
    'rt_chains_get_statistics': (c_int, [c_void_p, _p_f64, _p_i64, _p_i32]),
    'rt_chains_get_rows': (c_int, [c_void_p, c_int64, _p_i64, _p_i32, _p_f64, _p_i32]),
    'rt_chains_snapshot': (c_int, [c_void_p]),
    'rt_chains_restore': (c_int, [c_void_p, POINTER(c_ubyte)]),
    'rt_chains_destroy': (c_int, [c_void_p]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises HipLibraryError."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            'HIP library not found at %s -- build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C raoteh_amd/csrc`; there is no CPU fallback' % LIB_PATH)
    try:
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise HipLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise HipLibraryError('%s does not export %s' % (LIB_PATH, name))
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    # background compiles must have finished before the process runs its exit handlers
    import atexit
    atexit.register(handle.rt_jit_wait_all)
    return _lib


def last_error():
    msg = lib().rt_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''


def check(code):
    """Raise for a negative return code (ValueError for bad arguments, as the
    reference raises ValueError for bad shapes, _density.py:88-101)."""
    if code >= 0:
        return code
    msg = last_error()
    if code == RT_ERR_INVALID:
        raise ValueError(msg)
    if code == RT_ERR_NOMEM:
        raise MemoryError(msg)
    raise RaotehHipError(code, msg)
