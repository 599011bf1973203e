"""
Type-x observations (node -> observed state).  Mirror of
raoteh/sampler/_mcx_dense.py (state mask :48-86, get_node_to_pmap,
get_likelihood :241-305) on the HIP passes.
"""
from __future__ import annotations

import numpy as np

from . import _mc0_dense
from ._mcy_dense import _check_root, _run_passes
from ._tree import TreeArrays
from ._util import StructuralZeroProb

__all__ = ['get_node_to_pmap', 'get_likelihood']


def _define_state_mask(preorder_nodes, nstates, node_to_state=None):
    nnodes = len(preorder_nodes)
    state_mask = np.ones((nnodes, nstates), dtype=np.int64)
    if node_to_state is not None:
        for na_index, na in enumerate(preorder_nodes):
            if na in node_to_state:
                state_mask[na_index] = 0
                state_mask[na_index, node_to_state[na]] = 1
    return state_mask


def get_node_to_pmap(T, root, nstates, node_to_state=None, P_default=None):
    if len(T) == 1 and P_default is not None:
        _check_root(T, root)
        if node_to_state is not None and root in node_to_state:
            allowed = {node_to_state[root]}
        else:
            allowed = set(range(nstates))
        return {root: np.array([1 if s in allowed else 0
                                for s in range(nstates)], dtype=float)}
    ta = TreeArrays(T, root)
    state_mask = _define_state_mask(ta.preorder_nodes, nstates, node_to_state)
    esd = ta.esd_transitions(nstates, P_default=P_default)
    pmap = _run_passes(ta, esd, state_mask)
    return dict((na, pmap[i]) for i, na in enumerate(ta.preorder_nodes))


def get_likelihood(T, root, nstates, node_to_state=None, root_distn=None,
                   P_default=None):
    if len(T) == 1:
        _check_root(T, root)
        if node_to_state is not None and root in node_to_state:
            allowed_states = {node_to_state[root]}
        else:
            allowed_states = set(range(nstates))
        if root_distn is None:
            return 1
        pos = set(s for s in allowed_states if root_distn[s])
        if not pos:
            raise StructuralZeroProb(
                'the tree has only a single node, and every state with '
                'positive prior probability at the root is disallowed '
                'by a node state constraint')
        return sum(root_distn[s] for s in pos)
    node_to_pmap = get_node_to_pmap(T, root, nstates,
                                    node_to_state=node_to_state,
                                    P_default=P_default)
    return _mc0_dense.get_likelihood(node_to_pmap[root], root_distn=root_distn)
