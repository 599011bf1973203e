"""ctypes binding of oracle/liboracle.so (oracle.c) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, 'liboracle.so')
        if not os.path.exists(path):
            subprocess.check_call(['make', '-C', HERE])
        _lib = ctypes.CDLL(path)
        _lib.orc_root.restype = ctypes.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def expm(Q, t):
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    n = Q.shape[0]
    P = np.empty((n, n))
    work = np.empty(7 * n * n)
    info = np.zeros(2, dtype=np.int32)
    rc = lib().orc_expm(n, _p(Q, ctypes.c_double), ctypes.c_double(t),
                        _p(P, ctypes.c_double), _p(work, ctypes.c_double),
                        _p(info, ctypes.c_int32))
    if rc:
        raise np.linalg.LinAlgError('singular Pade denominator')
    return P, tuple(info)


def batch_loglik(idx, ptr, esd, obs_nodes, obs_dense, root_w):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    esd = np.ascontiguousarray(esd, dtype=np.float64)
    on = np.ascontiguousarray(obs_nodes, dtype=np.int64)
    od = np.ascontiguousarray(obs_dense, dtype=np.float64)
    nsites = od.shape[0]
    w = None if root_w is None else np.ascontiguousarray(root_w, dtype=np.float64)
    ll = np.empty(nsites)
    st = np.empty(nsites, dtype=np.int32)
    rc = lib().orc_batch_loglik(
        ctypes.c_int64(esd.shape[0]), esd.shape[1], _p(idx, ctypes.c_int64),
        _p(ptr, ctypes.c_int64), _p(esd, ctypes.c_double), ctypes.c_int64(len(on)),
        _p(on, ctypes.c_int64), _p(od, ctypes.c_double), ctypes.c_int64(nsites),
        None if w is None else _p(w, ctypes.c_double), _p(ll, ctypes.c_double),
        _p(st, ctypes.c_int32))
    assert rc == 0
    return ll, st


def batch_loglik_faithful(idx, ptr, Q, node_q, t, obs_nodes, obs_dense, root_w):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    nq = np.ascontiguousarray(node_q, dtype=np.int64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    on = np.ascontiguousarray(obs_nodes, dtype=np.int64)
    od = np.ascontiguousarray(obs_dense, dtype=np.float64)
    nsites = od.shape[0]
    w = None if root_w is None else np.ascontiguousarray(root_w, dtype=np.float64)
    ll = np.empty(nsites)
    st = np.empty(nsites, dtype=np.int32)
    rc = lib().orc_batch_loglik_faithful(
        ctypes.c_int64(len(t)), Q.shape[1], _p(idx, ctypes.c_int64),
        _p(ptr, ctypes.c_int64), _p(Q, ctypes.c_double), _p(nq, ctypes.c_int64),
        _p(t, ctypes.c_double), ctypes.c_int64(len(on)), _p(on, ctypes.c_int64),
        _p(od, ctypes.c_double), ctypes.c_int64(nsites),
        None if w is None else _p(w, ctypes.c_double), _p(ll, ctypes.c_double),
        _p(st, ctypes.c_int32))
    assert rc == 0
    return ll, st
