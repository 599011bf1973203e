"""
Type-x observations (node -> known state), SPARSE API: same names and argument
order as raoteh/sampler/_mcx.py (get_node_to_pset :36-138, get_node_to_pmap
:141-213, get_likelihood :216-256).  An observed node allows exactly its state,
an unobserved one every state; from there the computation is the type-y one,
on the GPU (the reference's _mcx is the pure-Python twin of that path).

Corner case kept from the accelerated path rather than from _mcx.py:89-95: a
state that is not a node of the parent edge's transition digraph is removed by
the forward pass (``get_node_to_pmap``) but not yet by ``get_node_to_pset``.
"""
from __future__ import annotations

from . import _mc0, _mcy

__all__ = ['get_node_to_pset', 'get_node_to_pmap', 'get_likelihood']


def _allowed(T, node_to_state):
    if node_to_state is None:
        return None
    return dict((node, {state}) for node, state in node_to_state.items()
                if node in T)


def get_node_to_pset(T, root, node_to_state=None, P_default=None):
    if len(set(T)) == 1:
        if root not in T:
            raise ValueError('unrecognized root')
        if (node_to_state is not None) and (root in node_to_state):
            return {root: {node_to_state[root]}}
        return {root: set(P_default)}
    return _mcy.get_node_to_pset(
        T, root, node_to_allowed_states=_allowed(T, node_to_state),
        P_default=P_default)


def get_node_to_pmap(T, root, node_to_state=None, P_default=None,
                     node_to_set=None):
    return _mcy.get_node_to_pmap(
        T, root, node_to_allowed_states=_allowed(T, node_to_state),
        P_default=P_default, node_to_set=node_to_set)


def get_likelihood(T, root, node_to_state=None, root_distn=None,
                   P_default=None):
    node_to_pmap = get_node_to_pmap(T, root, node_to_state=node_to_state,
                                    P_default=P_default)
    return _mc0.get_likelihood(node_to_pmap[root], root_distn=root_distn)
