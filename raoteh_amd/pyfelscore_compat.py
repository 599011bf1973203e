"""
Drop-in for the entry points of the third-party ``pyfelscore`` module that the
reference's likelihood path calls, backed by
libraoteh_hip.so.  A reference maintainer can write

    import raoteh_amd.pyfelscore_compat as pyfelscore

in raoteh/sampler/_mcy_dense.py / _mcx_dense.py / _tmjp_dense.py (see
INTEGRATION.md).  Arrays are mutated in place exactly as pyfelscore does.
"""
from __future__ import annotations

import numpy as np

from .device import get_context

__all__ = ['mcy_esd_get_node_to_pset', 'esd_get_node_to_set',
           'mcy_esd_get_node_to_pmap', 'mc0_esd_get_node_to_distn',
           'mc0_esd_get_joint_endpoint_distn', 'get_tolerance_rate_matrix',
           'get_mmpp_block', 'get_mmpp_block_zero_off_rate',
           'get_mmpp_frechet_all_positive', 'get_mmpp_frechet_diagonalizable_w_zero',
           'get_mmpp_frechet_defective_w_zero', 'get_tolerance_expectations']


def mcy_esd_get_node_to_pset(tree_csr_indices, tree_csr_indptr,
                             esd_transitions, state_mask):
    """call sites: _mcy_dense.py:168,270; _mcx_dense.py:145"""
    get_context().node_to_pset(tree_csr_indices, tree_csr_indptr,
                               esd_transitions, state_mask)


def esd_get_node_to_set(tree_csr_indices, tree_csr_indptr, esd_transitions,
                        state_mask):
    """call sites: _mcy_dense.py:175,277; _mcx_dense.py:152"""
    get_context().node_to_set(tree_csr_indices, tree_csr_indptr,
                              esd_transitions, state_mask)


def mcy_esd_get_node_to_pmap(tree_csr_indices, tree_csr_indptr,
                             esd_transitions, state_mask,
                             subtree_probability):
    """call sites: _mcy_dense.py:184,286; _mcx_dense.py:161"""
    get_context().node_to_pmap(tree_csr_indices, tree_csr_indptr,
                               esd_transitions, state_mask,
                               subtree_probability)


def get_tolerance_rate_matrix(t, Q, P):
    """Despite the name: P <- expm(t*Q) (_tmjp_dense.py:239,
    tests/test_expm.py:38-42)."""
    P[...] = get_context().expm(np.asarray(Q, dtype=float), [t])[0]


def mc0_esd_get_node_to_distn(tree_csr_indices, tree_csr_indptr, esd_transitions,
                              root_distn, subtree_probability,
                              node_to_distn_array):
    """call sites: _mc0_dense.py:381; _mcy_dense.py:195"""
    out, _ = get_context().node_to_distn(tree_csr_indices, tree_csr_indptr,
                                         esd_transitions, root_distn,
                                         subtree_probability)
    node_to_distn_array[...] = out


def mc0_esd_get_joint_endpoint_distn(tree_csr_indices, tree_csr_indptr,
                                     esd_transitions, subtree_probability,
                                     node_to_distn_array, joint_distns):
    """call site: _mcy_dense.py:205"""
    joint_distns[...] = get_context().joint_endpoint_distn(
        tree_csr_indices, tree_csr_indptr, esd_transitions, subtree_probability,
        node_to_distn_array)


# ---------------------------------------------------------------------------
# the 3-state tolerance process {0 -> 1: a, 1 -> 0: w, 1 -> 2: r} (_linalg.py:14-69,
# 92-118).  pyfelscore has closed forms for these; its source is not in the reference
# tree, so what is mirrored is the CONTRACT the call sites state: the numbers equal
# scipy.linalg.expm / expm_frechet of that matrix (tests/test_expm.py:20-42 for the
# blocks; _mjp.py:540-590 uses simple_expm_frechet exactly where it would otherwise use
# expm_frechet).  Here they come from the device's expm of the 3 x 3 matrix and of the
# nine 6 x 6 Frechet blocks [[tQ, tE_cd], [0, tQ]] (one launch), cached per (a, w, r, t)
# because the callers ask entry by entry.
# ---------------------------------------------------------------------------

def _tolerance_matrix(a, w, r):
    return np.array([[-a, a, 0.0], [w, -(w + r), r], [0.0, 0.0, 0.0]])


_frechet_cache = {}


def _tolerance_frechet(a, w, r, t):
    """L[c, d] = expm_frechet(tQ, t E_cd) for all nine directions, and expm(tQ)."""
    key = (float(a), float(w), float(r), float(t))
    hit = _frechet_cache.get(key)
    if hit is None:
        Q = _tolerance_matrix(*key[:3])
        blocks = np.zeros((10, 6, 6))
        for k in range(9):
            blocks[k, :3, :3] = blocks[k, 3:, 3:] = key[3] * Q
            blocks[k, k // 3, 3 + k % 3] = key[3]
        blocks[9, :3, :3] = key[3] * Q
        out = get_context().expm(blocks, np.ones(10))
        hit = (out[:9, :3, 3:].reshape(3, 3, 3, 3).copy(), out[9, :3, :3].copy())
        if len(_frechet_cache) > 4096:
            _frechet_cache.clear()
        _frechet_cache[key] = hit
    return hit


def get_mmpp_block(a, w, r, t):
    """Top-left 2 x 2 of expm(tQ) (_linalg.py:44; third row / column are rebuilt from
    row sums by the caller, :55-69)."""
    return _tolerance_frechet(a, w, r, t)[1][:2, :2].copy()


def get_mmpp_block_zero_off_rate(a, r, t):
    """The same with w = 0 (_linalg.py:46)."""
    return _tolerance_frechet(a, 0.0, r, t)[1][:2, :2].copy()


def get_mmpp_frechet_all_positive(a, w, r, t, ai, bi, ci, di):
    """Entry [ai, bi] of expm_frechet(tQ, t E_{ci,di}) (_linalg.py:111)."""
    return float(_tolerance_frechet(a, w, r, t)[0][ci, di, ai, bi])


def get_mmpp_frechet_diagonalizable_w_zero(a, r, t, ai, bi, ci, di):
    """w = 0, a != r (_linalg.py:114)."""
    return float(_tolerance_frechet(a, 0.0, r, t)[0][ci, di, ai, bi])


def get_mmpp_frechet_defective_w_zero(a, t, ai, bi, ci, di):
    """w = 0, r = a: the defective case (_linalg.py:117)."""
    return float(_tolerance_frechet(a, 0.0, a, t)[0][ci, di, ai, bi])


def get_tolerance_expectations(t, Q, P, J, expected_dwell_times, expected_transitions):
    """One edge of _tmjp_dense.py:320-340: adds the expected dwell times in tolerance
    states 0 (off) and 1 (on) to ``expected_dwell_times`` f64[2], the expected numbers of
    0 -> 1 and 1 -> 0 transitions to ``expected_transitions`` f64[2, 2] (off-diagonal
    entries; the diagonal is left alone), and returns the expected number of 1 -> 2
    (absorption) events: the reference's general formula (_mjp_dense.py:483-533) for this
    3 x 3 process, W = J / P on the support of J, one Frechet derivative on the device
    (rt_mjp_frechet_statistics).  pyfelscore's own source is absent from the reference
    tree: which entries it fills is taken from how _tmjp_dense.py:341-349 names the
    results (parity unpinned beyond that)."""
    Q = np.asarray(Q, dtype=float)
    P = np.asarray(P, dtype=float)
    J = np.asarray(J, dtype=float)
    if Q.shape != (3, 3) or P.shape != (3, 3) or J.shape != (3, 3):
        raise ValueError('expected 3 x 3 arrays')
    W = np.zeros((3, 3))
    live = J != 0
    W[live] = J[live] / P[live]
    dwell, trans = get_context().frechet_statistics(Q[None], [0], [float(t)], W[None])
    expected_dwell_times[0] += dwell[0]
    expected_dwell_times[1] += dwell[1]
    expected_transitions[0, 1] += trans[0, 1]
    expected_transitions[1, 0] += trans[1, 0]
    return float(trans[1, 2])
