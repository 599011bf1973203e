"""
Markov jump process likelihood on a tree, dense rate matrices -- the north-star
entry points with the reference's names, argument order and exceptions:

  custom_expm(Q, weight)                         raoteh/sampler/_mjp_dense.py:24-25
  get_expm_augmented_tree(T, root, Q_default)    raoteh/sampler/_mjp_dense.py:328-359
  get_likelihood(T, node_to_allowed_states, root, nstates, root_distn, Q_default)
                                                 raoteh/sampler/_mjp_dense.py:362-407
  get_expected_history_statistics(T, node_to_allowed_states, root, nstates,
                                  root_distn, Q_default)
                                                 raoteh/sampler/_mjp_dense.py:410-539

plus the batched forms the reference lacks (it loops over sites in Python,
examples/p53/p53.py:88-100):

  get_log_likelihoods(T, root, nstates, obs_nodes, data, kind, ...)
  get_total_log_likelihood(...)
  get_expected_history_statistics_batch(T, root, nstates, sites, ...)

Everything numerical runs in hand-written HIP kernels behind the C ABI.
"""
from __future__ import annotations

import networkx as nx
import numpy as np

from . import _mc0_dense, _mcy_dense
from ._tree import TreeArrays, check_square_dense
from .device import TreeModel, get_context

__all__ = ['custom_expm', 'get_expm_augmented_tree', 'get_likelihood',
           'get_expected_history_statistics', 'get_expected_history_statistics_batch',
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


def allowed_states_to_masks(sites, obs_nodes, nstates=None):
    """list of node->set dicts -> uint64[nsites, nobs] bit masks; with nstates > 64
    uint64[nsites, nobs, ceil(nstates / 64)] (bit s % 64 of word s // 64)."""
    words = 1 if nstates is None else max(1, (int(nstates) + 63) // 64)
    out = np.zeros((len(sites), len(obs_nodes), words), dtype=np.uint64)
    for i, d in enumerate(sites):
        for k, v in enumerate(obs_nodes):
            m = 0
            for s in d[v]:
                m |= 1 << int(s)
            for w in range(words):
                out[i, k, w] = (m >> (64 * w)) & (2 ** 64 - 1)
    return out[:, :, 0] if words == 1 else out


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


# ---------------------------------------------------------------------------
# expected history statistics (reference :410-539)
# ---------------------------------------------------------------------------
#
# The reference calls scipy.linalg.expm_frechet n + nnz(Q) times per edge, once
# per direction E_cd, and contracts each result with W = J / P (J the joint
# endpoint posterior, P the transition matrix, over the entries with J != 0).
# The same numbers come from ONE Frechet derivative per edge, by the adjoint
# identity <W, L(A, E)> = <L(A^T, W), E>:
#     M = L(t Q^T, W)      dwell[c] += t M[c, c]      trans[c, d] += t Q[c, d] M[c, d]
# and L(A, W) is the upper right block of expm([[A, W], [0, A]]) -- a 2n x 2n
# matrix exponential per edge, all edges in one launch of the expm kernel
# (csrc/expm.hip), W scaled to unit size first so that it does not drive the
# scaling-and-squaring of the block.

def _edge_rates(T, root, Q_default):
    """BFS edges of the tree, the distinct rate matrices on them and, per edge, which
    one it carries and its branch length."""
    edges = list(nx.bfs_edges(T, root))
    mats, index_of, q_index, ts = [], {}, [], []
    for na, nb in edges:
        Q = T[na][nb].get('Q', Q_default)
        check_square_dense(Q)
        key = id(Q)
        if key not in index_of:
            index_of[key] = len(mats)
            mats.append(np.asarray(Q, dtype=np.float64))
        q_index.append(index_of[key])
        ts.append(float(T[na][nb]['weight']))
    return edges, mats, np.array(q_index, dtype=np.int64), np.array(ts)


def _history_statistics_from_weights(ctx, nstates, mats, q_index, ts, Ws):
    """(dwell, trans) from the per-edge weights: block assembly, the Frechet block
    exponentials and the contraction over the edges all run on the device
    (rt_mjp_frechet_statistics); n <= 64, i.e. the 61-state codon model included."""
    if not len(ts):
        return np.zeros(nstates), np.zeros((nstates, nstates))
    if 2 * nstates > 128:                   # RT_MAX_EXPM_STATES
        raise ValueError('expected history statistics need the expm kernel at order 2n = '
                         '%d; it covers order <= 128' % (2 * nstates))
    return ctx.frechet_statistics(np.stack(mats), q_index, ts, Ws)


def get_expected_history_statistics(T, node_to_allowed_states, root, nstates,
                                    root_distn=None, Q_default=None):
    """One site, as the reference (:410-539): returns (dict state -> expected dwell
    time, posterior root distribution as a 1d ndarray, nx.DiGraph whose edge
    (c, d) carries the expected number of c -> d transitions as ``weight`` for
    every nonzero Q[c, d] of some edge, the diagonal included as in the
    reference).  expm, the upward passes, the downward pass, the joint endpoint
    distributions and the Frechet block exponentials run on the device."""
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    T_aug = get_expm_augmented_tree(T, root, Q_default=Q_default)
    node_to_pmap = _mcy_dense.get_node_to_pmap(
        T_aug, root, nstates, node_to_allowed_states=node_to_allowed_states)
    node_to_distn = _mc0_dense.get_node_to_distn(
        T_aug, root, node_to_pmap, nstates, root_distn=root_distn)
    T_joint = _mc0_dense.get_joint_endpoint_distn(
        T_aug, root, node_to_pmap, node_to_distn, nstates)
    edges, mats, q_index, ts = _edge_rates(T, root, Q_default)
    Ws = np.zeros((len(edges), nstates, nstates))
    for e, (na, nb) in enumerate(edges):
        J, P = T_joint[na][nb]['J'], T_aug[na][nb]['P']
        live = J != 0
        Ws[e][live] = J[live] / P[live]
    dwell, trans = _history_statistics_from_weights(get_context(), nstates, mats, q_index,
                                                    ts, Ws)
    expected_transitions = nx.DiGraph()
    for Q in mats:
        for c, d in zip(*np.nonzero(Q)):
            if not expected_transitions.has_edge(int(c), int(d)):
                expected_transitions.add_edge(int(c), int(d), weight=float(trans[c, d]))
    return (dict((c, float(dwell[c])) for c in range(nstates)),
            node_to_distn[root], expected_transitions)


def get_expected_history_statistics_batch(T, root, nstates, sites=None, root_distn=None,
                                          Q_default=None, weights=None,
                                          obs_nodes=None, data=None, kind='state'):
    """Sum over sites of the reference's per-site statistics (what an EM step over
    an alignment needs; the reference loops over sites in Python): ``sites`` is a
    list of node_to_allowed_states dicts.  Returns (dwell f64[n], summed root
    posteriors f64[n], transitions f64[n, n]).  The batched passes give every
    site's joint endpoint posteriors; their site sum (taken on the device,
    rt_mjp_esd_expectation_weights) enters ONE Frechet block exponential per edge,
    whatever the number of sites.  ``weights``: optional
    per-site multiplicities (site patterns).  Instead of ``sites``, the array form
    of get_log_likelihoods: ``obs_nodes`` + ``data`` [nsites, len(obs_nodes)] of
    states (kind='state', a value >= nstates = unobserved) or allowed-set bit masks
    (kind='mask')."""
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    ctx = get_context()
    T_aug = get_expm_augmented_tree(T, root, Q_default=Q_default)
    ta = TreeArrays(T_aug, root)
    esd = ta.esd_transitions(nstates)
    w = None if weights is None else np.asarray(weights, dtype=np.float64)
    # upward passes, downward pass and the per-edge site sums of J / P in one call;
    # n*n numbers per edge come back, whatever the number of sites
    if sites is not None:
        # node -> allowed-set dicts become one 64-bit set per site and node that any
        # site restricts (a node missing from a site's dict is unrestricted there); the
        # reference-format mask array is built on the device
        if nstates > 64:
            raise ValueError('allowed-state sets hold at most 64 states')
        obs_nodes = [v for v in ta.preorder_nodes
                     if any(d is not None and v in d for d in sites)]
        full = np.uint64((1 << nstates) - 1) if nstates < 64 else np.uint64(2 ** 64 - 1)
        data = np.full((len(sites), len(obs_nodes)), full, dtype=np.uint64)
        memo = {}
        for k, d in enumerate(sites):
            if d is None:
                continue
            for j, v in enumerate(obs_nodes):
                if v in d:
                    key = frozenset(d[v])
                    m = memo.get(key)
                    if m is None:
                        m = memo[key] = np.uint64(sum(1 << int(s) for s in key))
                    data[k, j] = m
        kind = 'mask'
    else:
        data = np.asarray(data)
        if data.ndim != 2 or data.shape[1] != len(obs_nodes):
            raise ValueError('data must be [nsites, len(obs_nodes)]')
        if kind == 'state':
            data = np.where((data < 0) | (data >= min(nstates, 255)), 255, data).astype(np.uint8)
            if nstates > 255:
                raise ValueError("kind='state' holds at most 255 states")
    # compact observations: one byte / one 64-bit set per site and observed node
    cols = [ta.node_to_index[v] for v in obs_nodes]
    if len(cols) == 0:                       # nothing observed anywhere
        cols = [0]
        data = (np.full((data.shape[0], 1), 255, dtype=np.uint8) if kind == 'state' else
                np.full((data.shape[0], 1), 2 ** 64 - 1 if nstates == 64 else (1 << nstates) - 1,
                        dtype=np.uint64))
    W, root_post, status = ctx.expectation_weights_obs(
        ta.indices, ta.indptr, esd, root_distn, cols, data, kind, site_weights=w)
    if status.any():
        from ._util import NumericalZeroProb
        raise NumericalZeroProb('the denominator is zero (site %d)'
                                % int(np.nonzero(status)[0][0]))
    edges, mats, q_index, ts = _edge_rates(T, root, Q_default)
    Ws = W[[ta.node_to_index[nb] for _, nb in edges]] if edges else \
        np.zeros((0, nstates, nstates))
    dwell, trans = _history_statistics_from_weights(ctx, nstates, mats, q_index, ts, Ws)
    return dwell, root_post, trans
