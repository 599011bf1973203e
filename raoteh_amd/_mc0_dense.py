"""
Root reduction and its error protocol -- host-side mirror of
raoteh/sampler/_mc0_dense.py:147-212 (``get_likelihood(root_pmap, root_distn)``).
The pmap itself comes from the HIP upward pass; this function only applies the
reference's zero-probability checks and the final n-term weighted sum.

``get_node_to_distn`` / ``get_node_to_distn_esd`` (downward pass, reference
:344-489) and ``get_joint_endpoint_distn`` (:217-270) run on the GPU through
rt_mc0_esd_get_node_to_distn / rt_mc0_esd_get_joint_endpoint_distn.
"""
from __future__ import annotations

import warnings

import networkx as nx
import numpy as np

from ._tree import TreeArrays, check_square_dense
from ._util import NumericalZeroProb, StructuralZeroProb

__all__ = ['get_likelihood', 'get_node_to_distn', 'get_node_to_distn_esd',
           'get_joint_endpoint_distn']


def get_likelihood(root_pmap, root_distn=None):
    """Root reduction with the reference's error protocol (_mc0_dense.py:167-212):
    shape mismatch -> ValueError; an all-zero prior, an all-zero pmap, or no state
    with both a positive weight and a positive subtree likelihood ->
    StructuralZeroProb (in that order); negative pmap entries are reported and
    clamped.  ``root_distn`` None = weights of one (_mjp_dense.py:389-393)."""
    weights = None
    if root_distn is not None:
        weights = np.asarray(root_distn)
        if np.shape(root_pmap) != weights.shape:
            raise ValueError('root shape mismatch: %s %s' % (np.shape(root_pmap),
                                                             weights.shape))
        if not np.any(weights != 0):
            raise StructuralZeroProb('no root state has nonzero prior likelihood')
    if root_pmap is None:
        raise ValueError('root_pmap is None')
    pmap = np.asarray(root_pmap)
    lowest = pmap.min()
    if lowest < 0:
        warnings.warn('root_pmap should have non-negative entries '
                      'but found minimum entry %s' % lowest)
        pmap = pmap.clip(min=0)
    support = pmap != 0                       # states the data allow at the root
    if not support.any():
        raise StructuralZeroProb(
            'all root states give a subtree likelihood of zero')
    if weights is not None:
        support = support & (weights != 0)    # ... that the prior allows as well
    if not support.any():
        raise StructuralZeroProb(
            'all root states have either zero prior likelihood '
            'or give a subtree likelihood of zero')
    return pmap.sum() if weights is None else weights.dot(pmap)


def get_node_to_distn(T, root, node_to_pmap, nstates, root_distn=None,
                      P_default=None):
    """Posterior marginal state distribution at every node
    (raoteh/sampler/_mc0_dense.py:400-489; the pyfelscore-accelerated twin is
    get_node_to_distn_esd, :344-395 -- both names map to the same kernel).
    Raises NumericalZeroProb where the reference's normaliser would."""
    from .device import get_context
    if P_default is not None:
        check_square_dense(P_default)
    if root_distn is not None and root_distn.shape[0] != nstates:
        raise ValueError('inconsistent root distribution')
    ta = TreeArrays(T, root)
    pmap = np.empty((ta.nnodes, nstates), dtype=np.float64)
    for i, na in enumerate(ta.preorder_nodes):
        if node_to_pmap[na].shape[0] != nstates:
            raise ValueError('inconsistent pmap')
        pmap[i] = node_to_pmap[na]
    if pmap.min() < -1e-6:                     # _util.py:131-135
        raise ValueError('expected non-negative entries but found %s' % pmap.min())
    esd = ta.esd_transitions(nstates, P_default=P_default)
    distn, status = get_context().node_to_distn(ta.indices, ta.indptr, esd,
                                                root_distn, pmap)
    if status[0]:
        raise NumericalZeroProb('the denominator is zero')
    return dict((na, distn[i]) for i, na in enumerate(ta.preorder_nodes))


get_node_to_distn_esd = get_node_to_distn


def get_joint_endpoint_distn(T, root, node_to_pmap, node_to_distn, nstates):
    """New nx.Graph whose BFS edges carry J = joint (parent state, child state)
    posterior as a 2d ndarray (raoteh/sampler/_mc0_dense.py:217-270)."""
    from .device import get_context
    ta = TreeArrays(T, root)
    pmap = np.empty((ta.nnodes, nstates), dtype=np.float64)
    distn = np.empty((ta.nnodes, nstates), dtype=np.float64)
    for i, na in enumerate(ta.preorder_nodes):
        pmap[i] = node_to_pmap[na]
        if node_to_distn[na].shape[0] != nstates:
            raise Exception('nstates inconsistency')
        distn[i] = node_to_distn[na]
    esd = ta.esd_transitions(nstates)
    J = get_context().joint_endpoint_distn(ta.indices, ta.indptr, esd, pmap, distn)
    T_aug = nx.Graph()
    for na, nb in nx.bfs_edges(T, root):
        T_aug.add_edge(na, nb, J=J[ta.node_to_index[nb]])
    return T_aug
