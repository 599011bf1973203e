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
           'mc0_esd_get_joint_endpoint_distn', 'get_tolerance_rate_matrix']


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
