"""
Continuous-time front end, SPARSE API: rate matrices are weighted nx.DiGraph
objects without self loops (the diagonal is implied), as in
raoteh/sampler/_mjp.py (get_expm_augmented_tree :349-381, get_likelihood
:384-428, get_expected_history_statistics :431-595) and raoteh/sampler/_linalg.py (sparse_expm :31-39,
sparse_expm_naive :72-90).

Per edge: densify Q over ``sorted(Q)``, diagonal = -row sum (_linalg.py:73-81),
expm on the GPU (rt_expm; every edge of the tree in ONE launch), keep entry
(sa, sb) only if sb is reachable from sa in Q (:83-89: the structural zeros of
P).  Two deliberate differences from the reference as published:
 * under networkx >= 2 ``all_pairs_shortest_path_length`` returns an iterator
   and _linalg.py:86 would yield an EMPTY P; the reachability intended there is
   computed explicitly here;
 * the 3-state "tolerance" form goes through the same expm instead of the
   closed form pyfelscore.get_mmpp_block (:41-69); the reference's own
   tests/test_expm.py:20-42 pins the two to each other, and the sparsity
   pattern of the closed form IS the reachability pattern.
"""
from __future__ import annotations

import networkx as nx
import numpy as np

from . import _mcy
from ._sparse import digraph_to_dense, dense_to_digraph
from .device import get_context

__all__ = ['sparse_expm', 'get_expm_augmented_tree', 'get_likelihood',
           'get_expected_history_statistics']


def _dense_rate_matrix(Q):
    states = sorted(Q)
    Q_dense = digraph_to_dense(Q, states)
    np.fill_diagonal(Q_dense, 0.0)
    Q_dense = Q_dense - np.diag(np.sum(Q_dense, axis=1))
    return states, Q_dense


def _reachability(Q, states):
    keep = np.zeros((len(states), len(states)), dtype=bool)
    index = dict((s, i) for i, s in enumerate(states))
    for sa in states:
        for sb in nx.single_source_shortest_path_length(Q, sa):
            keep[index[sa], index[sb]] = True
    return keep


def sparse_expm(Q, t):
    """expm(Q*t) of one sparse rate matrix as a sparse transition matrix."""
    states, Q_dense = _dense_rate_matrix(Q)
    P = get_context().expm(Q_dense[None], np.array([float(t)]))[0]
    return dense_to_digraph(P, states, keep=_reachability(Q, states))


def get_expm_augmented_tree(T, root, Q_default=None):
    """_mjp.py:349-381; all edges exponentiated in one device launch."""
    edges = list(nx.bfs_edges(T, root))
    mats, owners, per_matrix = [], [], {}
    for na, nb in edges:
        edge = T[na][nb]
        Q = edge.get('Q', Q_default)
        if Q is None:
            raise ValueError('no rate matrix is available for this edge')
        if id(Q) not in per_matrix:
            states, Q_dense = _dense_rate_matrix(Q)
            per_matrix[id(Q)] = (states, Q_dense, _reachability(Q, states))
        owners.append(id(Q))
        mats.append(per_matrix[id(Q)][1])
    T_aug = nx.Graph()
    T_aug.add_nodes_from(T)
    # one launch per matrix order (edge-specific matrices may differ in size)
    by_n = {}
    for k, M in enumerate(mats):
        by_n.setdefault(M.shape[0], []).append(k)
    P_of = {}
    for n, ks in by_n.items():
        Qs = np.stack([mats[k] for k in ks])
        ts = np.array([float(T[edges[k][0]][edges[k][1]]['weight']) for k in ks])
        Ps = get_context().expm(Qs, ts)
        for k, P in zip(ks, Ps):
            P_of[k] = P
    for k, (na, nb) in enumerate(edges):
        states, _, keep = per_matrix[owners[k]]
        T_aug.add_edge(na, nb, weight=T[na][nb]['weight'],
                       P=dense_to_digraph(P_of[k], states, keep=keep))
    return T_aug


def get_likelihood(T, node_to_allowed_states, root, root_distn=None,
                   Q_default=None):
    """_mjp.py:384-428."""
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    T_aug = get_expm_augmented_tree(T, root, Q_default=Q_default)
    return _mcy.get_likelihood(T_aug, root,
                               node_to_allowed_states=node_to_allowed_states,
                               root_distn=root_distn, P_default=None)


def get_expected_history_statistics(T, node_to_allowed_states, root,
                                    root_distn=None, Q_default=None):
    """_mjp.py:431-595: (dict state -> expected dwell time, dict state ->
    posterior root probability, nx.DiGraph of expected transition counts on the
    edges of the rate matrices).  The state space is the sorted union of the
    nodes of Q_default and of the edge-specific matrices (:479-487); everything
    numerical is the dense path (_mjp_dense.get_expected_history_statistics,
    one Frechet block exponential per edge on the device).  The reference's
    closed forms for its 3-state "simple" matrices (_linalg.py:92-118, absent
    pyfelscore) are the same Frechet derivatives."""
    from . import _mjp_dense
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    full_state_set = set()
    if Q_default is not None:
        full_state_set.update(Q_default)
    for na, nb in nx.bfs_edges(T, root):
        Q = T[na][nb].get('Q', None)
        if Q is not None:
            full_state_set.update(Q)
    states = sorted(full_state_set)
    index = dict((s, i) for i, s in enumerate(states))
    nstates = len(states)

    def densify(Q):
        D = digraph_to_dense(Q, states)
        np.fill_diagonal(D, 0.0)
        return D - np.diag(np.sum(D, axis=1))
    dense_of = {}
    T_dense = nx.Graph()
    T_dense.add_nodes_from(T)
    for na, nb in nx.bfs_edges(T, root):
        edge = T[na][nb]
        Q = edge.get('Q', Q_default)
        if Q is None:
            raise ValueError('no rate matrix is available for this edge')
        if id(Q) not in dense_of:
            dense_of[id(Q)] = densify(Q)
        T_dense.add_edge(na, nb, weight=edge['weight'], Q=dense_of[id(Q)])
    allowed = dict((v, set(index[s] for s in ss if s in index))
                   for v, ss in node_to_allowed_states.items())
    for v in T:
        allowed.setdefault(v, set(range(nstates)))
    distn = None
    if root_distn is not None:
        distn = np.zeros(nstates)
        for s, p in root_distn.items():
            if s in index:
                distn[index[s]] = p
    dwell, init, trans = _mjp_dense.get_expected_history_statistics(
        T_dense, allowed, root, nstates, root_distn=distn)
    expected_transitions = nx.DiGraph()
    for c, d, dat in trans.edges(data=True):
        if c != d:                      # sparse rate matrices carry no diagonal
            expected_transitions.add_edge(states[c], states[d], weight=dat['weight'])
    return (dict((states[c], x) for c, x in dwell.items()),
            dict((states[i], float(p)) for i, p in enumerate(init) if p),
            expected_transitions)
