"""
Markov jump process likelihood on a tree, dense rate matrices -- the north-star
entry points with the reference's names, argument order and exceptions:

  custom_expm(Q, weight)                         raoteh/sampler/_mjp_dense.py:24-25
  get_expm_augmented_tree(T, root, Q_default)    raoteh/sampler/_mjp_dense.py:328-359
  get_likelihood(T, node_to_allowed_states, root, nstates, root_distn, Q_default)
                                                 raoteh/sampler/_mjp_dense.py:362-407

plus the batched forms the reference lacks (it loops over sites in Python,
examples/p53/p53.py:88-100):

  get_log_likelihoods(T, root, nstates, obs_nodes, data, kind, ...)
  get_total_log_likelihood(...)

Everything numerical runs in hand-written HIP kernels behind the C ABI.
"""
from __future__ import annotations

import networkx as nx
import numpy as np

from . import _mcy_dense
from ._tree import check_square_dense
from .device import TreeModel, get_context

__all__ = ['custom_expm', 'get_expm_augmented_tree', 'get_likelihood',
           'get_log_likelihoods', 'get_total_log_likelihood',
           'allowed_states_to_masks']


def custom_expm(Q, weight):
    check_square_dense(Q)
    return get_context().expm(Q, [weight])[0]


def get_expm_augmented_tree(T, root, Q_default=None):
    """New nx.Graph whose BFS edges carry ``weight`` and ``P = expm(Q*weight)``
    with ``Q = edge.get('Q', Q_default)``; all edges in one device launch."""
    T_aug = nx.Graph()
    edges = list(nx.bfs_edges(T, root))
    if not edges:
        return T_aug
    slots, mats, q_index, weights = {}, [], [], []
    for na, nb in edges:
        edge = T[na][nb]
        Q = edge.get('Q', Q_default)
        check_square_dense(Q)
        if id(Q) not in slots:
            slots[id(Q)] = len(mats)
            mats.append(np.ascontiguousarray(Q, dtype=np.float64))
        q_index.append(slots[id(Q)])
        weights.append(edge['weight'])
    if len(set(m.shape for m in mats)) != 1:
        raise ValueError('rate matrices of different shapes on one tree')
    P = get_context().expm(np.stack(mats), weights, q_index=q_index)
    for (na, nb), w, Pe in zip(edges, weights, P):
        T_aug.add_edge(na, nb, weight=w, P=Pe)
    return T_aug


def get_likelihood(T, node_to_allowed_states, root, nstates,
                   root_distn=None, Q_default=None):
    """One site, as the reference: returns a float likelihood, raises
    ValueError / StructuralZeroProb like _mjp_dense.get_likelihood."""
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    T_aug = get_expm_augmented_tree(T, root, Q_default=Q_default)
    if len(T) == 1:
        T_aug.add_node(root)
    return _mcy_dense.get_likelihood(
        T_aug, root, nstates, node_to_allowed_states=node_to_allowed_states,
        root_distn=root_distn, P_default=None)


def allowed_states_to_masks(sites, obs_nodes):
    """list of node->set dicts -> uint64[nsites, nobs] bit masks."""
    out = np.zeros((len(sites), len(obs_nodes)), dtype=np.uint64)
    for i, d in enumerate(sites):
        for k, v in enumerate(obs_nodes):
            m = 0
            for s in d[v]:
                m |= 1 << int(s)
            out[i, k] = m
    return out


def _build(T, root, nstates, obs_nodes, data, kind, root_distn, Q_default,
           model):
    if model is None:
        model = TreeModel(T, root, nstates)
        model.set_rates(Q_default=Q_default)
    model.set_root_distn(root_distn)
    batch = model.upload_sites(obs_nodes, data, kind=kind)
    return model, batch


def get_log_likelihoods(T, root, nstates, obs_nodes, data, kind='dense',
                        root_distn=None, Q_default=None, model=None,
                        compress=False):
    """Per-site log-likelihoods for a batch of independent sites sharing
    (T, Q, branch lengths).  Returns (loglik f64[nsites], status int32[nsites]);
    status 1 marks a zero-probability site (loglik = -inf), for which the
    reference would raise StructuralZeroProb.  compress=True evaluates every
    distinct site pattern once (alignment columns repeat a lot) and copies the
    result to its duplicates."""
    if compress:
        from .io import compress_patterns
        unique, inverse, _ = compress_patterns(data)
        ll, st = get_log_likelihoods(T, root, nstates, obs_nodes, unique, kind=kind,
                                     root_distn=root_distn, Q_default=Q_default,
                                     model=model)
        return ll[inverse], st[inverse]
    model, batch = _build(T, root, nstates, obs_nodes, data, kind, root_distn,
                          Q_default, model)
    return model.log_likelihoods(batch)


def get_total_log_likelihood(T, root, nstates, obs_nodes, data, kind='dense',
                             root_distn=None, Q_default=None, model=None,
                             compress=False):
    """Sum over sites of the log-likelihood (examples/p53/p53.py:98-99).  With
    compress=True: sum over distinct patterns of count * log-likelihood (the sum
    is taken on the host; zero-probability patterns give -inf)."""
    if compress:
        from .io import compress_patterns
        unique, _, counts = compress_patterns(data)
        ll, _ = get_log_likelihoods(T, root, nstates, obs_nodes, unique, kind=kind,
                                    root_distn=root_distn, Q_default=Q_default,
                                    model=model)
        return float(np.dot(counts, ll))
    model, batch = _build(T, root, nstates, obs_nodes, data, kind, root_distn,
                          Q_default, model)
    return model.total_log_likelihood(batch)[0]
