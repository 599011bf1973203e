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
    if root_distn is not None:
        if root_pmap.shape != root_distn.shape:
            raise ValueError('root shape mismatch: '
                             '%s %s' % (root_pmap.shape, root_distn.shape))
        prior_feasible_rstates = set(s for s, p in enumerate(root_distn) if p)
        if not prior_feasible_rstates:
            raise StructuralZeroProb(
                'no root state has nonzero prior likelihood')
    if root_pmap is None:
        raise ValueError('root_pmap is None')
    root_pmap_min = root_pmap.min()
    if root_pmap_min < 0:
        warnings.warn('root_pmap should have non-negative entries '
                      'but found minimum entry %s' % root_pmap_min)
        root_pmap = np.maximum(root_pmap, 0)
    if not root_pmap.sum():
        raise StructuralZeroProb(
            'all root states give a subtree likelihood of zero')
    feasible_rstates = set(s for s, p in enumerate(root_pmap) if p)
    if root_distn is not None:
        feasible_rstates.intersection_update(prior_feasible_rstates)
    if not feasible_rstates:
        raise StructuralZeroProb(
            'all root states have either zero prior likelihood '
            'or give a subtree likelihood of zero')
    if root_distn is not None:
        return root_distn.dot(root_pmap)
    return root_pmap.sum()


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
