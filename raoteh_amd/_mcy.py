"""
Type-y observations (node -> set of allowed states), SPARSE API: transition
matrices are weighted nx.DiGraph objects over arbitrary sortable states,
results are dicts.  Same names, argument order and error behaviour as
raoteh/sampler/_mcy.py (get_node_to_set :323-345, get_node_to_pset :348-394,
get_node_to_pmap :563-608, get_likelihood :685-741); the three native passes
of its accelerated path (_esd_get_node_to_pset :274-320, _esd_get_node_to_set
:184-237, _esd_get_node_to_pmap :473-560) run on the GPU through the C ABI.
"""
from __future__ import annotations

import networkx as nx
import numpy as np

from . import _mc0
from ._sparse import SparseProblem
from ._util import StructuralZeroProb, get_first_element
from .device import get_context

__all__ = ['get_node_to_set', 'get_node_to_pset', 'get_node_to_pmap',
           'get_likelihood']


def _check_root(T, root):
    if root not in T:
        raise ValueError('unrecognized root')


def _check_matrices(T, root, P_default):
    bfs_edges = list(nx.bfs_edges(T, root))
    all_custom_P = all('P' in T[na][nb] for na, nb in bfs_edges)
    if (P_default is None) and (not all_custom_P):
        raise ValueError('expected a custom transition on each edge '
                         'when a default transition matrix is not available')


def _single_node(T, root, P_default, node_to_allowed_states):
    # _mcy.py:139-181, 240-271 restricted to a tree without edges
    _check_root(T, root)
    allowed = set(P_default)
    if node_to_allowed_states is not None and root in node_to_allowed_states:
        allowed &= set(node_to_allowed_states[root])
    return {root: allowed}


def _masks(T, root, node_to_allowed_states, P_default, forward):
    prob = SparseProblem(T, root, P_default=P_default)
    mask = prob.mask_from_allowed(node_to_allowed_states)
    ctx = get_context()
    ctx.node_to_pset(prob.ta.indices, prob.ta.indptr, prob.esd, mask)
    if forward:
        ctx.node_to_set(prob.ta.indices, prob.ta.indptr, prob.esd, mask)
    return prob, mask


def get_node_to_pset(T, root, node_to_allowed_states=None, P_default=None):
    _check_root(T, root)
    if len(T) == 1 and P_default is not None:
        return _single_node(T, root, P_default, node_to_allowed_states)
    _check_matrices(T, root, P_default)
    prob, mask = _masks(T, root, node_to_allowed_states, P_default, False)
    return prob.mask_to_dict(mask)


def get_node_to_set(T, root, node_to_allowed_states=None, P_default=None):
    _check_root(T, root)
    if len(T) == 1 and P_default is not None:
        return _single_node(T, root, P_default, node_to_allowed_states)
    _check_matrices(T, root, P_default)
    prob, mask = _masks(T, root, node_to_allowed_states, P_default, True)
    return prob.mask_to_dict(mask)


def get_node_to_pmap(T, root, node_to_allowed_states=None, P_default=None,
                     node_to_set=None):
    if len(T) == 1 and P_default is not None:
        _check_root(T, root)
        allowed_states = set(P_default)
        if node_to_allowed_states is not None:
            allowed_states &= set(node_to_allowed_states[root])
        return {root: dict((s, 1.0) for s in allowed_states)}
    _check_root(T, root)
    prob = SparseProblem(T, root, P_default=P_default)
    mask = prob.mask_from_allowed(node_to_allowed_states)
    pmap = np.empty(mask.shape, dtype=np.float64)
    # pset + set + pmap in one device call (rt_mcy_esd_passes)
    get_context().passes(prob.ta.indices, prob.ta.indptr, prob.esd, mask, pmap)
    if node_to_set is not None:
        mine = prob.mask_to_dict(mask)
        if mine != dict((k, set(v)) for k, v in node_to_set.items()):
            raise Exception('internal error %s %s' % (mine, node_to_set))
    return prob.pmap_to_dict(mask, pmap)


def get_likelihood(T, root, node_to_allowed_states=None, root_distn=None,
                   P_default=None):
    if len(T) == 1:
        if get_first_element(T) != root:
            raise Exception('the tree has only a single node, '
                            'but this node is not the root')
        allowed_states = node_to_allowed_states[root]
        if not allowed_states:
            raise StructuralZeroProb('the tree has only a single node, '
                                     'and no state is allowed for the root')
        if root_distn is None:
            return 1
        nonzero_prob_states = set(allowed_states) & set(root_distn)
        if not nonzero_prob_states:
            raise StructuralZeroProb(
                'the tree has only a single node, and every state with '
                'positive prior probability at the root is disallowed '
                'by a node state constraint')
        return sum(root_distn[s] for s in nonzero_prob_states)
    node_to_pmap = get_node_to_pmap(
        T, root, node_to_allowed_states=node_to_allowed_states,
        P_default=P_default)
    return _mc0.get_likelihood(node_to_pmap[root], root_distn=root_distn)
