"""
Type-y observations (node -> set of allowed states), dense transition matrices.
Same names, argument order and error behaviour as
raoteh/sampler/_mcy_dense.py (get_node_to_pmap :302-354, get_likelihood
:433-493); the three native passes run on the GPU through the C ABI
(rt_mcy_esd_get_node_to_pset / rt_esd_get_node_to_set /
rt_mcy_esd_get_node_to_pmap) instead of pyfelscore.
"""
from __future__ import annotations

import numpy as np

from . import _mc0_dense
from ._tree import TreeArrays
from ._util import StructuralZeroProb
from .device import get_context

__all__ = ['get_node_to_pmap', 'get_likelihood', 'kitchen_sink']


def _define_state_mask(node_to_allowed_states, preorder_nodes, nstates):
    """raoteh/sampler/_mcy_dense.py:43-54 (KeyError for a node missing from a
    given dict is the reference's behaviour)."""
    nnodes = len(preorder_nodes)
    if node_to_allowed_states is None:
        return np.ones((nnodes, nstates), dtype=np.int64)
    state_mask = np.zeros((nnodes, nstates), dtype=np.int64)
    for na_index, na in enumerate(preorder_nodes):
        allowed = [sa for sa in node_to_allowed_states[na] if 0 <= sa < nstates]
        state_mask[na_index, allowed] = 1
    return state_mask


def _check_root(T, root):
    if root not in T:
        raise ValueError('unrecognized root')


def _run_passes(ta, esd, state_mask, obs_likelihood=None):
    """The pass sequence of _mcy_dense.py:261-291 on the device."""
    pmap = np.empty(state_mask.shape, dtype=np.float64)
    get_context().passes(ta.indices, ta.indptr, esd, state_mask, pmap,
                         obs_likelihood=obs_likelihood)
    return pmap


def _esd_get_node_to_pmap(T, root, nstates, node_to_allowed_states=None,
                          P_default=None):
    ta = TreeArrays(T, root)
    state_mask = _define_state_mask(node_to_allowed_states, ta.preorder_nodes,
                                    nstates)
    esd = ta.esd_transitions(nstates, P_default=P_default)
    pmap = _run_passes(ta, esd, state_mask)
    node_to_pmap = dict((na, pmap[i]) for i, na in enumerate(ta.preorder_nodes))
    return state_mask, node_to_pmap


def get_node_to_pmap(T, root, nstates, node_to_allowed_states=None,
                     P_default=None, node_to_set=None):
    if len(T) == 1 and P_default is not None:
        _check_root(T, root)
        allowed_states = set(range(nstates))
        if node_to_allowed_states is not None:
            allowed_states &= set(node_to_allowed_states[root])
        root_pmap = np.array(
            [1 if s in allowed_states else 0 for s in range(nstates)],
            dtype=float)
        return {root: root_pmap}
    restriction = (node_to_set if node_to_set is not None
                   else node_to_allowed_states)
    _, node_to_pmap = _esd_get_node_to_pmap(
        T, root, nstates, node_to_allowed_states=restriction,
        P_default=P_default)
    return node_to_pmap


def get_likelihood(T, root, nstates, node_to_allowed_states=None,
                   root_distn=None, P_default=None):
    if len(T) == 1:
        _check_root(T, root)
        allowed_states = node_to_allowed_states[root]
        if not allowed_states:
            raise StructuralZeroProb('the tree has only a single node, '
                                     'and no state is allowed for the root')
        if root_distn is None:
            return 1
        pos_prob_states = set(s for s in allowed_states if root_distn[s])
        if not pos_prob_states:
            raise StructuralZeroProb(
                'the tree has only a single node, and every state with '
                'positive prior probability at the root is disallowed '
                'by a node state constraint')
        return sum(root_distn[s] for s in pos_prob_states)
    node_to_pmap = get_node_to_pmap(
        T, root, nstates, node_to_allowed_states=node_to_allowed_states,
        P_default=P_default)
    return _mc0_dense.get_likelihood(node_to_pmap[root], root_distn=root_distn)


def kitchen_sink(T, root, nstates, node_to_allowed_states=None, root_distn=None,
                 P_default=None):
    """State mask, upward messages, posterior node marginals and joint endpoint
    distributions in one go (raoteh/sampler/_mcy_dense.py:57-230).  Returns
    (state_mask, node_to_pmap, node_to_distn, edge_to_joint_distn)."""
    from ._util import NumericalZeroProb
    ta = TreeArrays(T, root)
    state_mask = _define_state_mask(node_to_allowed_states, ta.preorder_nodes,
                                    nstates)
    esd = ta.esd_transitions(nstates, P_default=P_default)
    pmap = _run_passes(ta, esd, state_mask)
    ctx = get_context()
    distn, status = ctx.node_to_distn(ta.indices, ta.indptr, esd, root_distn, pmap)
    if status[0]:
        raise NumericalZeroProb('the denominator is zero')
    J = ctx.joint_endpoint_distn(ta.indices, ta.indptr, esd, pmap, distn)
    node_to_pmap, node_to_distn, edge_to_joint_distn = {}, {}, {}
    for i, na in enumerate(ta.preorder_nodes):
        node_to_pmap[na] = pmap[i]
        node_to_distn[na] = distn[i]
        for j in range(ta.indptr[i], ta.indptr[i + 1]):
            nb_index = ta.indices[j]
            edge_to_joint_distn[(na, ta.preorder_nodes[nb_index])] = J[nb_index]
    return state_mask, node_to_pmap, node_to_distn, edge_to_joint_distn
