"""
Marshalling between the reference's sparse API (transition / rate matrices as
weighted ``nx.DiGraph`` over arbitrary sortable state labels, results as dicts)
and the dense arrays of the C ABI.

Follows raoteh/sampler/_mcy.py: the state space is the sorted union of the node
sets of every transition matrix in use (:487-493), matrices are densified over
it with absent edges as structural zeros (_get_esd_transitions, :77-105), masks
come from ``node_to_allowed_states`` with missing nodes unrestricted
(_define_state_mask, :108-122), and results go back to dicts keyed by state
(_state_mask_to_dict, :125-136; pmap conversion :537-559).
"""
from __future__ import annotations

import networkx as nx
import numpy as np

from ._tree import TreeArrays

__all__ = ['SparseProblem', 'digraph_to_dense', 'dense_to_digraph']


def digraph_to_dense(G, sorted_states):
    """Weighted nx.DiGraph -> f64[n, n] over sorted_states; states of
    sorted_states that are not nodes of G get zero rows and columns."""
    index = dict((s, i) for i, s in enumerate(sorted_states))
    M = np.zeros((len(sorted_states), len(sorted_states)), dtype=np.float64)
    for sa, sb, data in G.edges(data=True):
        if sa in index and sb in index:
            M[index[sa], index[sb]] = data['weight']
    return M


def dense_to_digraph(M, sorted_states, keep=None):
    """f64[n, n] -> weighted nx.DiGraph; ``keep[a][b]`` (bool) selects the
    entries that become edges (default: the non-zero ones)."""
    G = nx.DiGraph()
    for a, sa in enumerate(sorted_states):
        for b, sb in enumerate(sorted_states):
            if (keep[a][b] if keep is not None else M[a, b] != 0):
                G.add_edge(sa, sb, weight=float(M[a, b]))
    return G


class SparseProblem(object):
    """Tree arrays + dense transition matrices of one sparse-API call."""

    def __init__(self, T, root, P_default=None):
        self.ta = TreeArrays(T, root)
        state_set = set()
        for i in range(1, self.ta.nnodes):
            P = self.ta.edge_data[i].get('P', P_default)
            if P is None:
                raise ValueError('expected either a default transition matrix '
                                 'or a transition matrix on every edge')
            state_set.update(set(P))
        self.sorted_states = sorted(state_set)
        self.nstates = len(self.sorted_states)
        self.state_to_index = dict((s, i) for i, s in
                                   enumerate(self.sorted_states))
        esd = np.zeros((self.ta.nnodes, self.nstates, self.nstates),
                       dtype=np.float64)
        cache = {}
        for i in range(1, self.ta.nnodes):
            P = self.ta.edge_data[i].get('P', P_default)
            if id(P) not in cache:
                cache[id(P)] = digraph_to_dense(P, self.sorted_states)
            esd[i] = cache[id(P)]
        self.esd = esd

    def mask_from_allowed(self, node_to_allowed_states):
        mask = np.ones((self.ta.nnodes, self.nstates), dtype=np.int64)
        if node_to_allowed_states is not None:
            for i, na in enumerate(self.ta.preorder_nodes):
                if na in node_to_allowed_states:
                    allowed = node_to_allowed_states[na]
                    for j, s in enumerate(self.sorted_states):
                        if s not in allowed:
                            mask[i, j] = 0
        return mask

    def mask_to_dict(self, mask):
        return dict(
            (na, set(s for j, s in enumerate(self.sorted_states) if mask[i, j]))
            for i, na in enumerate(self.ta.preorder_nodes))

    def pmap_to_dict(self, mask, pmap):
        return dict(
            (na, dict((s, float(pmap[i, j]))
                      for j, s in enumerate(self.sorted_states) if mask[i, j]))
            for i, na in enumerate(self.ta.preorder_nodes))
