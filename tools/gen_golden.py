#!/usr/bin/env python
"""
Generate tests/golden/*.json by running the REFERENCE's own pure-Python code.

Runs only in the build container (it reads /root/reference); the GPU box never
needs it.  Nothing from the reference is copied: the outputs are data (inputs +
expected outputs).  Import recipe: the two package
``__init__`` files import the long-gone ``numpy.testing.Tester``, so empty
package objects whose ``__path__`` points at the reference directories are
registered instead, and an empty module named ``pyfelscore`` lets ``_mcy`` and
``_mcz`` import (none of the functions used below calls into it: they are the
reference's "unaccelerated" twins, ``_mcy.py:396-470,611-682``,
``_mcz.py:94-166``, ``_mc0.py:89-138``).

Reference functions exercised:
  _mcx.get_likelihood / get_node_to_pmap      raoteh/sampler/_mcx.py:141-256
  _mc0.get_likelihood                          raoteh/sampler/_mc0.py:202-252
  _mc0.get_history_log_likelihood              raoteh/sampler/_mc0.py:141-199
  _mc0.get_node_to_set_unaccelerated           raoteh/sampler/_mc0.py:89-138
  _mcy.unaccelerated_get_node_to_pset/_pmap    raoteh/sampler/_mcy.py:396-470,611-682
  _mcz.get_node_to_pmap                        raoteh/sampler/_mcz.py:94-166
  _linalg.sparse_expm_naive                    raoteh/sampler/_linalg.py:72-90
  _conditional_expectation.get_jukes_cantor_*  raoteh/sampler/_conditional_expectation.py:15-33
  examples/code2x3/run.py do_blinking_process  :329-475 (builder of the config-5 model)
  _graph_transform.get_chunk_tree_type_b       raoteh/sampler/_graph_transform.py:298-375
  _mc0.get_node_to_distn                       raoteh/sampler/_mc0.py:382-462
  examples/p53/liwen.py:566-636,677-682        the 122-state switching model (fixture_switching)
  _tmjp.get_inhomogeneous_mjp                  raoteh/sampler/_tmjp.py:803-903 (the sparse twin of
                                               pyfelscore.tmjp_get_inhomogeneous_mjp)
expm per edge is ``scipy.linalg.expm(Q*t)`` exactly as ``_mjp_dense.py:24-25``.

usage: python tools/gen_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib
import itertools
import json
import os
import sys
import time
import types

import networkx as nx
import numpy as np
import scipy
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

REF = '/root/reference'


def import_reference():
    for name, path in (('raoteh', REF + '/raoteh'),
                       ('raoteh.sampler', REF + '/raoteh/sampler')):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    sys.modules.setdefault('pyfelscore', types.ModuleType('pyfelscore'))
    mods = {}
    for mod in ('_util', '_density', '_mc0', '_mcx', '_mcy', '_mcz',
                '_conditional_expectation'):
        mods[mod] = importlib.import_module('raoteh.sampler.' + mod)
    return mods


def dense_to_nx(P):
    """Dense transition matrix -> the weighted nx.DiGraph the sparse reference
    API uses; zero entries are absent edges (structural zeros)."""
    G = nx.DiGraph()
    n = P.shape[0]
    G.add_nodes_from(range(n))
    for i in range(n):
        for j in range(n):
            if P[i, j] != 0:
                G.add_edge(i, j, weight=float(P[i, j]))
    return G


def tree_json(T):
    return [[int(a), int(b), dict((k, v) for k, v in d.items()
                                  if k == 'weight')]
            for a, b, d in T.edges(data=True)]


def edges_in_insertion_order(T):
    return [[int(a), int(b), float(d.get('weight', 1.0))]
            for a, b, d in T.edges(data=True)]


def augmented(T, root, Q_default):
    """nx tree whose BFS edges carry P = expm(Q*t) as nx.DiGraph (for the
    reference's sparse API) and the same P dense (for the fixture)."""
    T_aug = nx.Graph()
    T_aug.add_nodes_from(T)
    dense = {}
    for na, nb in nx.bfs_edges(T, root):
        edge = T[na][nb]
        Q = edge.get('Q', Q_default)
        P = scipy.linalg.expm(np.asarray(Q) * edge['weight'])
        T_aug.add_edge(na, nb, weight=edge['weight'], P=dense_to_nx(P))
        dense[nb] = P
    return T_aug, dense


def ref_type_y(mods, T_aug, root, node_to_allowed_states):
    """Likelihood + pmaps for allowed-set observations through the reference's
    unaccelerated functions."""
    _mc0, _mcy = mods['_mc0'], mods['_mcy']
    pset = _mcy.unaccelerated_get_node_to_pset(
        T_aug, root, node_to_allowed_states=node_to_allowed_states)
    nset = _mc0.get_node_to_set_unaccelerated(T_aug, root, pset)
    pmap = _mcy.unaccelerated_get_node_to_pmap(
        T_aug, root, node_to_allowed_states=node_to_allowed_states,
        node_to_set=nset)
    return pset, nset, pmap


def pmap_json(pmap, nstates):
    return dict((str(int(v)), [float(m.get(s, 0.0)) for s in range(nstates)])
                for v, m in pmap.items())


def set_json(d):
    return dict((str(int(v)), sorted(int(s) for s in ss))
                for v, ss in d.items())


# ---------------------------------------------------------------------------
# fixtures
# ---------------------------------------------------------------------------

def fixture_test_mjp(mods):
    """tests/test_mjp.py:91-164 inputs; likelihood for every rooting and the
    16-term marginalisation."""
    _mcx, _mc0 = mods['_mcx'], mods['_mc0']
    T = nx.Graph()
    T.add_weighted_edges_from([(0, 1, 2.0), (0, 2, 3.0), (0, 3, 4.0),
                               (1, 4, 5.0), (1, 5, 6.0)])
    nstates = 4
    distn = {0: 0.1, 1: 0.2, 2: 0.3, 3: 0.4}
    Q = np.zeros((4, 4))
    for a, b, w in [(0, 1, 1.0 * distn[1]), (1, 0, 1.0 * distn[0]),
                    (1, 2, 2.0 * distn[2]), (2, 1, 2.0 * distn[1]),
                    (2, 3, 1.0 * distn[3]), (3, 2, 1.0 * distn[2]),
                    (3, 0, 2.0 * distn[0]), (0, 3, 2.0 * distn[3])]:
        Q[a, b] = w
    Q -= np.diag(Q.sum(axis=1))
    node_to_state = {2: 0, 3: 1, 4: 2, 5: 3}
    out = dict(edges=edges_in_insertion_order(T), nstates=nstates,
               root_distn=[distn[s] for s in range(4)], Q=Q.tolist(),
               node_to_state=dict((str(k), v) for k, v in node_to_state.items()),
               rootings=[])
    for root in range(6):
        T_aug, _ = augmented(T, root, Q)
        lk = _mcx.get_likelihood(T_aug, root, node_to_state=node_to_state,
                                 root_distn=distn)
        marg = 0.0
        for s0 in range(4):
            for s1 in range(4):
                nm = dict(node_to_state)
                nm[0] = s0
                nm[1] = s1
                try:
                    marg += _mcx.get_likelihood(T_aug, root, node_to_state=nm,
                                                root_distn=distn)
                except mods['_util'].StructuralZeroProb:
                    pass
        out['rootings'].append(dict(root=root, likelihood=lk,
                                    marginalised=marg))
    return out


def fixture_sum_to_one(mods):
    """tests/test_mjp.py:52-89 with a fixed seed: star tree, 3 states, all 81
    singleton assignments; the likelihoods sum to 1."""
    _mcx = mods['_mcx']
    rng = np.random.RandomState(20131004)
    T = nx.Graph()
    T.add_weighted_edges_from([(0, 1, 2.0), (0, 2, 3.0), (0, 3, 4.0)])
    n = 3
    w = rng.exponential(size=n)
    distn = w / w.sum()
    Q = rng.exponential(size=(n, n))
    np.fill_diagonal(Q, 0)
    Q -= np.diag(Q.sum(axis=1))
    T_aug, _ = augmented(T, 0, Q)
    liks = []
    for assignment in itertools.product(range(n), repeat=4):
        nts = dict(zip(range(4), assignment))
        lk = _mcx.get_likelihood(
            T_aug, 0, node_to_state=nts,
            root_distn=dict(enumerate(distn.tolist())))
        liks.append(lk)
    return dict(edges=edges_in_insertion_order(T), nstates=n, root=0,
                root_distn=distn.tolist(), Q=Q.tolist(),
                assignments=[list(a) for a in
                             itertools.product(range(n), repeat=4)],
                likelihoods=liks, total=float(np.sum(liks)))


def fixture_kat_history(mods):
    """tests/test_mc.py:131-150: history log likelihood == 4*log(0.5)."""
    _mc0 = mods['_mc0']
    T = nx.Graph()
    T.add_edge(0, 1)
    T.add_edge(0, 2)
    T.add_edge(0, 3)
    P = np.array([[0.5, 0.25, 0.25], [0.25, 0.5, 0.25], [0.25, 0.25, 0.5]])
    node_to_state = {0: 0, 1: 0, 2: 0, 3: 0}
    root_distn = {0: 0.5, 1: 0.5, 2: 0, 3: 0}
    ll = _mc0.get_history_log_likelihood(T, 0, node_to_state,
                                         root_distn=root_distn,
                                         P_default=dense_to_nx(P))
    return dict(edges=[[0, 1, 1.0], [0, 2, 1.0], [0, 3, 1.0]], nstates=3,
                P=P.tolist(), root=0, root_distn=[0.5, 0.5, 0.0],
                node_to_state={'0': 0, '1': 0, '2': 0, '3': 0},
                history_log_likelihood=ll,
                literal='4*log(0.5)', literal_value=4 * np.log(0.5))


def fixture_jukes_cantor(mods):
    """_conditional_expectation.py:25-33: p_ij(t) closed form vs the rate
    matrix of :15-23 (weights 1/(n-1))."""
    ce = mods['_conditional_expectation']
    rows = []
    for n in (3, 4, 7):
        for t in (0.01, 0.5, 2.0, 10.0):
            rows.append(dict(
                n=n, t=t,
                p_same=ce.get_jukes_cantor_probability(0, 0, t, n),
                p_diff=ce.get_jukes_cantor_probability(0, 1, t, n)))
    return dict(rows=rows)


def _random_sparse_P(rng, n):
    """Random row-stochastic matrix with one structural zero per row
    (the generator pattern of tests/test_mc.py:33-49)."""
    P = np.zeros((n, n))
    for i in range(n):
        jmiss = rng.randint(n)
        w = rng.exponential(size=n)
        w[jmiss] = 0
        P[i] = w / w.sum()
    return P


def fixture_random_sparse(mods, ncases=24):
    """Random small trees with a random sparse P on every edge, sparse root
    distribution, and one disallowed state per node (the set-up of
    tests/test_mc.py:52-102, seed fixed): per-node pset / set / pmap and the
    likelihood (or StructuralZeroProb) from the reference's unaccelerated
    type-y path; plus a type-z variant with random observation likelihoods."""
    _mc0, _mcz, _util = mods['_mc0'], mods['_mcz'], mods['_util']
    rng = np.random.RandomState(1234)
    cases = []
    for case in range(ncases):
        n = int(rng.randint(2, 6))
        nnodes = int(rng.randint(2, 9))
        # random recursive tree with shuffled, non-contiguous ids
        ids = (rng.permutation(30)[:nnodes] + 3).tolist()
        T = nx.Graph()
        T.add_node(ids[0])
        Ps = {}
        for k in range(1, nnodes):
            parent = ids[rng.randint(k)]
            T.add_edge(parent, ids[k], weight=1.0)
        root = ids[int(rng.randint(nnodes))]
        T_aug = nx.Graph()
        T_aug.add_nodes_from(T)
        for na, nb in nx.bfs_edges(T, root):
            P = _random_sparse_P(rng, n)
            Ps[nb] = P
            T_aug.add_edge(na, nb, P=dense_to_nx(P))
        w = rng.exponential(size=n)
        w[rng.randint(n)] = 0
        distn = w / w.sum()
        distn_dict = dict((i, float(p)) for i, p in enumerate(distn) if p)
        allowed = dict((v, set(range(n))) for v in T)
        for v in T:
            allowed[v].discard(int(rng.randint(n)))
        rec = dict(edges=[[int(a), int(b), 1.0] for a, b in T.edges()],
                   nodes=[int(v) for v in T], root=int(root), nstates=n,
                   root_distn=distn.tolist(),
                   P=dict((str(int(k)), v.tolist()) for k, v in Ps.items()),
                   allowed=set_json(allowed))
        pset, nset, pmap = ref_type_y(mods, T_aug, root, allowed)
        rec['pset'] = set_json(pset)
        rec['set'] = set_json(nset)
        rec['pmap'] = pmap_json(pmap, n)
        try:
            rec['likelihood'] = _mc0.get_likelihood(pmap[root],
                                                    root_distn=distn_dict)
            rec['zero'] = False
        except _util.StructuralZeroProb:
            rec['likelihood'] = 0.0
            rec['zero'] = True
        # downward pass + joint endpoint distributions (_mc0.py:255-308,382-462)
        if not rec['zero']:
            nd = _mc0.get_node_to_distn(T_aug, root, pmap, root_distn=distn_dict)
            rec['distn'] = pmap_json(nd, n)
            TJ = _mc0.get_joint_endpoint_distn(T_aug, root, pmap, nd)
            joint = {}
            for na, nb in nx.bfs_edges(T, root):
                Jm = np.zeros((n, n))
                for sa, sb, dat in TJ[na][nb]['J'].edges(data=True):
                    Jm[sa, sb] = dat['weight']
                joint[str(int(nb))] = Jm.tolist()
            rec['joint'] = joint
        # type-z: random likelihood for every (node, state)
        obs = dict((v, dict((s, float(rng.uniform(0.1, 1.0)))
                            for s in range(n))) for v in T)
        pmap_z = _mcz.get_node_to_pmap(T_aug, root,
                                       node_to_state_to_likelihood=obs,
                                       node_to_set=nset)
        rec['obs_lik'] = dict((str(int(v)), [m[s] for s in range(n)])
                              for v, m in obs.items())
        rec['pmap_z'] = pmap_json(pmap_z, n)
        cases.append(rec)
    return dict(cases=cases)


def fixture_sparse_api(mods, ncases=16):
    """The sparse (nx.DiGraph / dict) API with arbitrary, unsorted state labels:
    rate matrices as sparse digraphs without loops, per-edge P from the
    reference's own _linalg.sparse_expm_naive (raoteh/sampler/_linalg.py:72-90;
    networkx >= 2 returns an iterator from all_pairs_shortest_path_length, so
    the harness hands that function a dict, as SURVEY.md 8a row 3 asks), then
    the reference's pure-Python twins: _mcx (:36-256), _mcy unaccelerated
    (:396-470, 611-682), _mc0 (:89-138, 202-252), _mcz (:94-166)."""
    _mc0, _mcx, _mcy, _mcz, _util = (mods[k] for k in
                                     ('_mc0', '_mcx', '_mcy', '_mcz', '_util'))
    _linalg = importlib.import_module('raoteh.sampler._linalg')
    real_apspl = nx.all_pairs_shortest_path_length

    class _NxShim(object):
        def __getattr__(self, name):
            return getattr(nx, name)
        @staticmethod
        def all_pairs_shortest_path_length(G):
            return dict(real_apspl(G))
    _linalg.nx = _NxShim()

    rng = np.random.RandomState(4321)
    labels_pool = [3, 40, 7, 11, 25, 2]
    cases = []
    for case in range(ncases):
        n = int(rng.randint(2, 6))
        labels = [labels_pool[i] for i in rng.permutation(len(labels_pool))[:n]]
        nnodes = int(rng.randint(2, 9))
        ids = (rng.permutation(30)[:nnodes] + 3).tolist()
        T = nx.Graph()
        T.add_node(ids[0])
        for k in range(1, nnodes):
            T.add_edge(ids[rng.randint(k)], ids[k],
                       weight=float(rng.uniform(0.05, 0.6)))
        root = ids[int(rng.randint(nnodes))]

        def random_Q():
            Q = nx.DiGraph()
            Q.add_nodes_from(labels)
            for sa in labels:
                for sb in labels:
                    if sa != sb and rng.uniform() < 0.55:
                        Q.add_edge(sa, sb, weight=float(rng.exponential()))
            return Q
        Q_default = random_Q()
        edge_Q = {}
        for na, nb in nx.bfs_edges(T, root):
            if rng.uniform() < 0.4:
                edge_Q[nb] = random_Q()
        T_aug = nx.Graph()
        T_aug.add_nodes_from(T)
        P_json = {}
        for na, nb in nx.bfs_edges(T, root):
            Q = edge_Q.get(nb, Q_default)
            P = _linalg.sparse_expm_naive(Q, T[na][nb]['weight'])
            T_aug.add_edge(na, nb, weight=T[na][nb]['weight'], P=P)
            P_json[str(int(nb))] = [[int(sa), int(sb), float(d['weight'])]
                                    for sa, sb, d in P.edges(data=True)]
        w = rng.exponential(size=n)
        if n > 2:
            w[rng.randint(n)] = 0
        root_distn = dict((labels[i], float(p)) for i, p in enumerate(w / w.sum())
                          if p)
        allowed = dict((v, set(labels)) for v in T)
        for v in T:
            if rng.uniform() < 0.7:
                allowed[v].discard(labels[int(rng.randint(n))])
        node_to_state = dict((v, labels[int(rng.randint(n))]) for v in T
                             if T.degree(v) == 1 and v != root
                             and rng.uniform() < 0.8)
        rec = dict(
            labels=[int(x) for x in labels], root=int(root),
            nodes=[int(v) for v in T],
            edges=[[int(a), int(b), float(d['weight'])]
                   for a, b, d in T.edges(data=True)],
            Q_default=[[int(a), int(b), float(d['weight'])]
                       for a, b, d in Q_default.edges(data=True)],
            edge_Q=dict((str(int(k)), [[int(a), int(b), float(d['weight'])]
                                       for a, b, d in Q.edges(data=True)])
                        for k, Q in edge_Q.items()),
            P=P_json,
            root_distn=dict((str(k), v) for k, v in root_distn.items()),
            allowed=dict((str(int(v)), sorted(int(x) for x in ss))
                         for v, ss in allowed.items()),
            node_to_state=dict((str(int(v)), int(x))
                               for v, x in node_to_state.items()))

        def sets_json(d):
            return dict((str(int(v)), sorted(int(x) for x in ss))
                        for v, ss in d.items())

        def pmap_sparse_json(d):
            return dict((str(int(v)), dict((str(int(k)), float(x))
                                           for k, x in m.items()))
                        for v, m in d.items())

        # type y
        pset = _mcy.unaccelerated_get_node_to_pset(
            T_aug, root, node_to_allowed_states=allowed)
        nset = _mc0.get_node_to_set_unaccelerated(T_aug, root, pset)
        rec['y_pset'] = sets_json(pset)
        rec['y_set'] = sets_json(nset)
        try:
            pmap = _mcy.unaccelerated_get_node_to_pmap(
                T_aug, root, node_to_allowed_states=allowed, node_to_set=nset)
            rec['y_pmap'] = pmap_sparse_json(pmap)
            rec['y_likelihood'] = float(_mc0.get_likelihood(
                pmap[root], root_distn=root_distn))
            rec['y_zero'] = False
        except (_util.StructuralZeroProb, ValueError):
            # an empty node set: the reference's pure-Python pmap raises
            # 'internal error'; the likelihood is a structural zero
            rec['y_zero'] = True
        # type x
        try:
            rec['x_likelihood'] = float(_mcx.get_likelihood(
                T_aug, root, node_to_state=node_to_state,
                root_distn=root_distn))
            rec['x_pmap'] = pmap_sparse_json(_mcx.get_node_to_pmap(
                T_aug, root, node_to_state=node_to_state))
            rec['x_zero'] = False
        except (_util.StructuralZeroProb, ValueError):
            rec['x_zero'] = True
        # type z on the type-y support
        if not rec['y_zero']:
            obs = dict((v, dict((s, float(rng.uniform(0.1, 1.0)))
                                for s in allowed[v])) for v in T)
            pz = _mcz.get_node_to_pmap(T_aug, root,
                                       node_to_state_to_likelihood=obs,
                                       node_to_set=nset)
            rec['z_obs'] = pmap_sparse_json(obs)
            rec['z_pmap'] = pmap_sparse_json(pz)
        cases.append(rec)
    return dict(cases=cases)


def _reference_mjp(mods):
    """raoteh.sampler._mjp with the two harness adaptations the other fixtures use:
    _linalg's nx.all_pairs_shortest_path_length returns a dict (networkx >= 2
    hands back an iterator), and _mcy.get_node_to_pmap -- whose published body
    calls the absent pyfelscore -- is the reference's own unaccelerated twins
    (ref_type_y: _mcy.py:396-470, _mc0.py:89-138, _mcy.py:611-682)."""
    _linalg = importlib.import_module('raoteh.sampler._linalg')
    real_apspl = nx.all_pairs_shortest_path_length

    class _NxShim(object):
        def __getattr__(self, name):
            return getattr(nx, name)
        @staticmethod
        def all_pairs_shortest_path_length(G):
            return dict(real_apspl(G))
    _linalg.nx = _NxShim()
    _mjp = importlib.import_module('raoteh.sampler._mjp')

    class _McyTwins(object):
        def __getattr__(self, name):
            return getattr(mods['_mcy'], name)
        @staticmethod
        def get_node_to_pmap(T, root, node_to_allowed_states=None,
                             P_default=None, node_to_set=None):
            return ref_type_y(mods, T, root, node_to_allowed_states)[2]
    _mjp._mcy = _McyTwins()
    return _mjp


def _expectation_record(info, n):
    dwell, init, trans = info
    M = np.zeros((n, n))
    for sa, sb, dat in trans.edges(data=True):
        M[sa, sb] = dat['weight']
    return dict(dwell=[float(dwell.get(s, 0.0)) for s in range(n)],
                init=[float(init.get(s, 0.0)) for s in range(n)],
                trans=M.tolist())


def fixture_expectations(mods, ncases=10):
    """Expected history statistics (dwell time per state, posterior root
    distribution, expected transition counts) from the reference's
    _mjp.get_expected_history_statistics (raoteh/sampler/_mjp.py:431-595; the
    dense twin _mjp_dense.py:410-539 is pinned to it by tests/test_mjp.py:166-237).
     * 'jukes_cantor': the known-answer case of tests/test_mjp.py:166-237 -- a
       5-node path, end states (a, b), every root; expected dwell times from the
       closed form _conditional_expectation.py:25-47;
     * 'cases': random trees, random (sparse-ish) rate matrices with two
       edge-specific ones, allowed-state sets at the leaves, random root prior."""
    ce = mods['_conditional_expectation']
    _mjp = _reference_mjp(mods)
    n = 4
    t = 0.5
    T = nx.Graph()
    for k, f in enumerate((0.1, 0.2, 0.3, 0.4)):
        T.add_edge(k, k + 1, weight=f * t)
    Q = ce.get_jukes_cantor_rate_matrix(n)
    jc = []
    for a in range(n):
        for b in range(n):
            allowed = dict((v, set(range(n))) for v in T)
            allowed[0] = {a}
            allowed[4] = {b}
            closed = [ce.get_jukes_cantor_interaction(a, b, i, i, t, n) /
                      ce.get_jukes_cantor_probability(a, b, t, n) for i in range(n)]
            for root in (0, 2, 4):
                info = _mjp.get_expected_history_statistics(
                    T, allowed, root, Q_default=Q)
                rec = _expectation_record(info, n)
                rec.update(a=a, b=b, root=root, closed_form_dwell=closed)
                jc.append(rec)
    Qd = np.zeros((n, n))
    for sa, sb, dat in Q.edges(data=True):
        Qd[sa, sb] = dat['weight']
    Qd -= np.diag(Qd.sum(axis=1))
    jukes = dict(nstates=n, t=t, edges=edges_in_insertion_order(T), Q=Qd.tolist(),
                 rows=jc)

    rng = np.random.RandomState(97531)
    cases = []
    for case in range(ncases):
        n = int(rng.randint(4, 7))
        nnodes = int(rng.randint(4, 10))
        T = nx.Graph()
        T.add_node(0)
        for k in range(1, nnodes):
            T.add_edge(int(rng.randint(k)), k, weight=float(rng.uniform(0.05, 1.2)))
        root = int(rng.randint(nnodes))

        def random_rates():
            R = rng.exponential(size=(n, n))
            R[rng.uniform(size=(n, n)) < 0.2] = 0.0
            np.fill_diagonal(R, 0.0)
            for i in range(n):                   # a cycle keeps every state reachable
                if R[i, (i + 1) % n] == 0:
                    R[i, (i + 1) % n] = float(rng.uniform(0.2, 1.0))
            return R
        mats = [random_rates(), random_rates()]

        def to_digraph(R):
            G = nx.DiGraph()
            G.add_nodes_from(range(n))
            for i in range(n):
                for j in range(n):
                    if R[i, j]:
                        G.add_edge(i, j, weight=float(R[i, j]))
            return G
        graphs = [to_digraph(R) for R in mats]
        edge_q = {}
        for na, nb in T.edges():
            if rng.uniform() < 0.3:
                T[na][nb]['Q'] = graphs[1]
                edge_q[(na, nb)] = 1
            else:
                edge_q[(na, nb)] = 0
        allowed = dict((v, set(range(n))) for v in T)
        for v in T:
            if T.degree(v) == 1 or rng.uniform() < 0.15:
                k = int(rng.randint(1, 3))
                allowed[v] = set(int(x) for x in rng.permutation(n)[:k])
        w = rng.exponential(size=n)
        w /= w.sum()
        distn = dict((i, float(p)) for i, p in enumerate(w))
        info = _mjp.get_expected_history_statistics(
            T, allowed, root, root_distn=distn, Q_default=graphs[0])
        rec = _expectation_record(info, n)
        dense = []
        for R in mats:
            D = R.copy()
            D -= np.diag(D.sum(axis=1))
            dense.append(D.tolist())
        rec.update(nstates=n, root=root, root_distn=w.tolist(),
                   edges=[[int(a), int(b), float(d['weight']), edge_q[(a, b)]]
                          for a, b, d in T.edges(data=True)],
                   Q=dense, allowed=set_json(allowed))
        cases.append(rec)
    return dict(jukes_cantor=jukes, cases=cases)


def fixture_p53_mg94(mods):
    """The MG94 codon rate matrix the reference's p53 example builds
    (examples/p53/create_mg94.py:23-142 called as examples/p53/p53.py:44-47) on the
    genetic-code table shipped with it.  networkx >= 3 dropped to_numpy_matrix,
    which raoteh/sampler/_density.py:52 calls: the harness provides it."""
    sys.path.insert(0, REF + '/examples/p53')
    if not hasattr(nx, 'to_numpy_matrix'):
        nx.to_numpy_matrix = lambda G, **kw: np.asmatrix(nx.to_numpy_array(G, **kw))
    import create_mg94
    from raoteh_amd import io as rio
    here = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'p53')
    code = rio.read_genetic_code(os.path.join(here, 'universal.code.txt'))
    nt = dict(A=0.25039, C=0.30126, G=0.25952, T=0.18883)
    kappa, omega = 3.17632, 0.21925
    Qnx, distn, _, _ = create_mg94.create_mg94(
        nt['A'], nt['C'], nt['G'], nt['T'], kappa, omega, code,
        target_expected_rate=1.0)
    n = len(code)
    Q = np.zeros((n, n))
    for a, b, d in Qnx.edges(data=True):
        Q[a, b] = d['weight']
    return dict(Q_offdiagonal=Q.tolist(), distn=[float(distn[i]) for i in range(n)],
                kappa=kappa, omega=omega, nt=nt)


def fixture_config(mods, name, nsites, with_pmap=False):
    """A BASELINE.json-shaped configuration evaluated by the reference path:
    E x scipy.linalg.expm + _mcx.get_likelihood (type-x observations, C1-C3)
    or the unaccelerated type-y path (C5)."""
    from raoteh_amd import synth
    _mcx, _mc0 = mods['_mcx'], mods['_mc0']
    cfg = synth.make_config(name, nsites=nsites)
    T, root, leaves, n = cfg['T'], cfg['root'], cfg['leaves'], cfg['nstates']
    t0 = time.time()
    T_aug, dense = augmented(T, root, cfg['Q_default'])
    distn = cfg['root_distn']
    distn_dict = dict((i, float(p)) for i, p in enumerate(distn) if p)
    liks = []
    pmaps = []
    for site in range(nsites):
        if cfg['obs_kind'] == 'state':
            nts = dict((leaf, int(cfg['leaf_states'][site, k]))
                       for k, leaf in enumerate(leaves))
            pm = _mcx.get_node_to_pmap(T_aug, root, node_to_state=nts)
        else:
            allowed = synth.site_node_to_allowed_states(cfg, site)
            _, _, pm = ref_type_y(mods, T_aug, root, allowed)
        liks.append(_mc0.get_likelihood(pm[root], root_distn=distn_dict))
        if with_pmap:
            pmaps.append(pmap_json(pm, n))
    seconds = time.time() - t0
    rec = dict(config=name, nsites=nsites, nstates=n, root=int(root),
               leaves=[int(v) for v in leaves],
               edges=edges_in_insertion_order(T),
               root_distn=np.asarray(distn).tolist(),
               obs_kind=cfg['obs_kind'],
               leaf_states=cfg['leaf_states'][:nsites].tolist(),
               likelihoods=liks,
               log_likelihoods=[float(np.log(x)) for x in liks],
               reference_seconds=seconds)
    if cfg['Q_default'] is not None:
        rec['Q_default'] = np.asarray(cfg['Q_default']).tolist()
    else:
        rec['Q_edges'] = dict(
            (str(int(nb)), np.asarray(T[na][nb]['Q']).tolist())
            for na, nb in nx.bfs_edges(T, root))
        rec['leaf_allowed'] = [list(map(int, a)) for a in cfg['leaf_allowed']]
    # a few scipy expm outputs so expm parity is testable by itself
    keep = list(dense)[:3] if n > 20 else list(dense)
    rec['P_scipy'] = dict((str(int(k)), dense[k].tolist()) for k in keep)
    if with_pmap:
        rec['pmaps'] = pmaps
    return rec


def fixture_expm(mods):
    """scipy.linalg.expm(Q*t) on the matrix families the reference tests
    (tests/test_expm.py:44-82: 3-state tolerance forms a/b/c at
    t = 2^-5..2^5) plus HKY/MG94/blinking matrices at small and large t."""
    from raoteh_amd import synth
    rng = np.random.RandomState(1234)
    rows = []
    for t in np.logspace(-5, 5, 10, base=2):
        for form in 'abc':
            if form == 'a':
                a, w, r = rng.exponential(size=3)
                R = [(0, 1, a), (1, 0, w), (1, 2, r)]
            elif form == 'b':
                a, r = rng.exponential(size=2)
                R = [(0, 1, a), (1, 2, r)]
            else:
                a = rng.exponential()
                R = [(0, 1, a), (1, 2, a)]
            Q = np.zeros((3, 3))
            for i, j, x in R:
                Q[i, j] = x
            Q -= np.diag(Q.sum(axis=1))
            rows.append(dict(form=form, t=float(t), Q=Q.tolist(),
                             P=scipy.linalg.expm(Q * t).tolist()))
    Qh, _ = synth.hky85()
    Qm, _ = synth.mg94()
    c5 = synth.make_config('c5', nsites=1)
    Qb = c5['T'][0][1]['Q']
    for label, Q in (('hky85', Qh), ('blinking20', Qb), ('mg94', Qm)):
        for t in (1e-4, 0.05, 0.3, 1.0, 4.0, 40.0):
            if label == 'mg94' and t not in (0.05, 1.0, 40.0):
                continue
            rows.append(dict(form=label, t=t, Q=np.asarray(Q).tolist(),
                             P=scipy.linalg.expm(np.asarray(Q) * t).tolist()))
    return dict(rows=rows)


def fixture_blinking(mods):
    """The reference's own builder of the blinking compound process
    (examples/code2x3/run.py:329-461, ``do_blinking_process``) run on the
    (nprimary = 5, nparts = 2) model of config 5: its rate matrix, its root
    distribution and its per-node allowed compound-state sets, captured where
    the reference hands them to ``_mjp_dense.get_likelihood`` (run.py:473-475).
    run.py imports ``raoteh.sampler._mjp_dense`` (pyfelscore) and ``extras``:
    both are replaced by recording stand-ins for the import; nothing after the
    captured call is used."""
    import importlib.util
    import io
    import contextlib
    captured = []
    fake_mjp = types.ModuleType('raoteh.sampler._mjp_dense')

    def get_likelihood(T, node_to_allowed_states, root, nstates,
                       root_distn=None, Q_default=None):
        captured.append(dict(T=T, allowed=node_to_allowed_states, root=root,
                             nstates=nstates, root_distn=root_distn,
                             Q=Q_default))
        return 1.0
    fake_mjp.get_likelihood = get_likelihood
    fake_extras = types.ModuleType('extras')
    fake_extras.get_expected_ntransitions = lambda *a, **k: {}
    saved = dict((k, sys.modules.get(k)) for k in
                 ('raoteh.sampler._mjp_dense', 'extras'))
    sys.modules['raoteh.sampler._mjp_dense'] = fake_mjp
    sys.modules['raoteh.sampler']._mjp_dense = fake_mjp
    sys.modules['extras'] = fake_extras
    try:
        spec = importlib.util.spec_from_file_location(
            'code2x3_run', REF + '/examples/code2x3/run.py')
        run = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(run)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        if hasattr(sys.modules['raoteh.sampler'], '_mjp_dense'):
            del sys.modules['raoteh.sampler']._mjp_dense
    from raoteh_amd import synth
    nprimary, nparts = 5, 2
    primary_to_part = {0: 0, 1: 0, 2: 0, 3: 1, 4: 1}
    rng = np.random.RandomState(4)
    w = 0.5 + rng.exponential(size=nprimary)
    primary_distn = w / w.sum()
    S = 0.5 + rng.exponential(size=(nprimary, nprimary))
    S = (S + S.T) / 2
    Q_primary = synth._finish_rate_matrix(S * primary_distn[None, :], primary_distn)
    rows = []
    # a 7-node tree: leaves 3..6 observe primary states, the others nothing;
    # no tolerance ("disease") data anywhere, as in config 5
    preorder_nodes = [0, 1, 3, 4, 2, 5, 6]
    preorder_edges = [(0, 1), (1, 3), (1, 4), (0, 2), (2, 5), (2, 6)]
    for rate_on, rate_off, leaf_primary in ((1.3, 0.7, (0, 3, 4, 2)),
                                            (0.55, 2.4, (1, 1, 3, 0)),
                                            (0.5 + rng.exponential(),
                                             0.5 + rng.exponential(), (4, 2, 0, 3))):
        allowed_primary = dict((v, set(range(nprimary))) for v in preorder_nodes)
        for leaf, c in zip((3, 4, 5, 6), leaf_primary):
            allowed_primary[leaf] = {c}
        part_allowed = dict(((v, p), {0, 1}) for v in preorder_nodes
                            for p in range(nparts))
        del captured[:]
        with contextlib.redirect_stdout(io.StringIO()):
            run.do_blinking_process(
                Q_primary, primary_distn, preorder_nodes, preorder_edges, 0.1,
                primary_to_part, rate_on, rate_off, allowed_primary, part_allowed)
        cap = captured[0]
        rows.append(dict(
            rate_on=float(rate_on), rate_off=float(rate_off),
            leaf_primary=list(leaf_primary), leaves=[3, 4, 5, 6],
            Q=np.asarray(cap['Q']).tolist(),
            root_distn=np.asarray(cap['root_distn']).tolist(),
            nstates=int(cap['nstates']),
            allowed=dict((str(v), sorted(int(x) for x in ss))
                         for v, ss in cap['allowed'].items())))
    return dict(nprimary=nprimary, nparts=nparts,
                primary_to_part=dict((str(k), v) for k, v in primary_to_part.items()),
                Q_primary=Q_primary.tolist(), primary_distn=primary_distn.tolist(),
                rows=rows)


def fixture_forest(mods, ncases=24):
    """The Rao-Teh sweep core on a ragged batch of trees with ONE shared matrix: for
    random trees with random event nodes the reference's chunk tree
    (_graph_transform.get_chunk_tree_type_b, :298-375), a uniformized P = I + Q / omega
    of a random sparse rate matrix (_sample_mjp_dense.py:107-110 restated: the module
    itself imports pyfelscore-bound code), allowed-state sets on some chunk nodes, and
    the reference's outputs for P_default = that P on every edge: pset
    (_mcy.unaccelerated_get_node_to_pset), set (_mc0.get_node_to_set_unaccelerated),
    pmap (_mcy.unaccelerated_get_node_to_pmap) and, where the likelihood is positive,
    the exact posterior node marginals (_mc0.get_node_to_distn) the sampled states of
    _sample_mc0.resample_states must follow."""
    _mc0, _mcy, _util = mods['_mc0'], mods['_mcy'], mods['_util']
    _gt = importlib.import_module('raoteh.sampler._graph_transform')
    rng = np.random.RandomState(4321)
    cases = []
    for case in range(ncases):
        n = int(rng.choice([2, 3, 4, 6, 9]))
        # sparse rate matrix with a connected support (a cycle) plus random extras
        Q = np.zeros((n, n))
        for i in range(n):
            Q[i, (i + 1) % n] = rng.exponential() + 0.1
        extra = rng.uniform(size=(n, n)) < 0.25
        Q += extra * rng.exponential(size=(n, n))
        np.fill_diagonal(Q, 0.0)
        Q -= np.diag(Q.sum(axis=1))
        omega = 2.0 * (-np.diag(Q)).max()
        P = np.identity(n) + Q / omega
        P_nx = dense_to_nx(P)
        # a random tree, some of whose degree-2-ish nodes are events -> its chunk tree
        nnodes = int(rng.randint(3, 26))
        T = nx.Graph()
        T.add_node(0)
        for k in range(1, nnodes):
            T.add_edge(int(rng.randint(k)), k)
        events = set(int(v) for v in range(1, nnodes) if rng.uniform() < 0.6)
        chunk_tree, edge_to_chunk, event_to_edge = _gt.get_chunk_tree_type_b(T, 0, events)
        root = 0
        nodes = list(chunk_tree)
        allowed = dict((v, set(range(n))) for v in nodes)
        for v in nodes:
            u = rng.uniform()
            if u < 0.3:
                allowed[v] = {int(rng.randint(n))}
            elif u < 0.5:
                allowed[v] = set(int(x) for x in rng.choice(n, size=max(1, n // 2),
                                                             replace=False))
        w = rng.exponential(size=n)
        if rng.uniform() < 0.3:
            w[rng.randint(n)] = 0.0
        distn = w / w.sum()
        distn_dict = dict((i, float(p)) for i, p in enumerate(distn) if p)
        rec = dict(nstates=n, P=P.tolist(), Q=Q.tolist(), omega=float(omega),
                   tree_edges=[[int(a), int(b)] for a, b in T.edges()],
                   event_nodes=sorted(events),
                   chunk_nodes=[int(v) for v in nodes],
                   chunk_edges=[[int(a), int(b)] for a, b in nx.bfs_edges(chunk_tree, root)]
                   if len(nodes) > 1 else [],
                   root=root, allowed=set_json(allowed), root_distn=distn.tolist())
        if len(nodes) == 1:
            # a single chunk: the reference's passes need at least one edge
            rec['single'] = True
            cases.append(rec)
            continue
        pset = _mcy.unaccelerated_get_node_to_pset(
            chunk_tree, root, node_to_allowed_states=allowed, P_default=P_nx)
        nset = _mc0.get_node_to_set_unaccelerated(chunk_tree, root, pset, P_default=P_nx)
        pmap = _mcy.unaccelerated_get_node_to_pmap(
            chunk_tree, root, node_to_allowed_states=allowed, node_to_set=nset,
            P_default=P_nx)
        rec['pset'] = set_json(pset)
        rec['set'] = set_json(nset)
        rec['pmap'] = pmap_json(pmap, n)
        try:
            rec['likelihood'] = _mc0.get_likelihood(pmap[root], root_distn=distn_dict)
            rec['zero'] = False
            nd = _mc0.get_node_to_distn(chunk_tree, root, pmap, root_distn=distn_dict,
                                        P_default=P_nx)
            rec['distn'] = pmap_json(nd, n)
        except _util.StructuralZeroProb:
            rec['likelihood'] = 0.0
            rec['zero'] = True
        cases.append(rec)
    return dict(cases=cases)


def fixture_chunk_trees(mods, ncases=30):
    """Histories as the sampler holds them between steps 2 and 3 of a sweep: a base tree
    whose edges carry event nodes of degree two (the transitions kept from the last sweep
    and the fresh Poisson events, _sampler.py:376-381), and the reference's chunk tree of
    each (_graph_transform.get_chunk_tree_type_b, :298-375): which chunk every directed
    edge of the history lies in, and the chunk tree's edges."""
    _gt = importlib.import_module('raoteh.sampler._graph_transform')
    rng = np.random.RandomState(977)
    cases = []
    for case in range(ncases):
        N = int(rng.randint(2, 18))
        parent = [-1] + [int(rng.randint(max(0, v - 5), v)) for v in range(1, N)]
        counts = [0] + [int(rng.choice([1, 1, 1, 2, 3, 5])) for _ in range(1, N)]   # pieces
        T = nx.Graph()
        T.add_node(0)
        nxt = N
        pieces = []
        events = set()
        for v in range(1, N):
            prev = parent[v]
            for k in range(counts[v]):
                if k == counts[v] - 1:
                    nb = v
                else:
                    nb = nxt
                    nxt += 1
                    events.add(nb)
                T.add_edge(prev, nb)
                pieces.append([v, k, int(prev), int(nb)])
                prev = nb
        chunk_tree, edge_to_chunk, event_to_edge = _gt.get_chunk_tree_type_b(T, 0, events)
        cases.append(dict(
            parent=parent, counts=counts, pieces=pieces, event_nodes=sorted(events),
            edge_to_chunk=[[int(a), int(b), int(c)] for (a, b), c in edge_to_chunk.items()],
            chunk_nodes=sorted(int(v) for v in chunk_tree),
            chunk_edges=sorted([int(a), int(b)] for a, b in
                               (event_to_edge[e] for e in sorted(events)))))
    return dict(cases=cases)


def fixture_spectral(mods):
    """The reference's optional spectral path (examples/p53/qtop.py): its own random
    reversible rate matrices (random_reversible_rate_matrix, :346-379: the first state has
    zero stationary probability), the decomposition decompose_spectral_v2 (:142-150) and
    getp_spectral_v2 (:76-88) at several branch lengths, next to getp_rate_matrix
    (scipy.linalg.expm, :24-28) of the same matrix -- the two things qtop.py's own
    test_spectral_v2_expm (:587-609) compares; plus the MG94 codon matrix of p53.py.
    numpy.testing.run_module_suite no longer exists (qtop.py:16 imports it for its
    __main__): the harness provides a stand-in."""
    import numpy.testing as npt
    if not hasattr(npt, 'run_module_suite'):
        npt.run_module_suite = lambda *a, **k: None
    sys.path.insert(0, REF + '/examples/p53')
    import qtop
    cases = []
    np.random.seed(1234)                       # the seed of qtop.py:588
    for n, ts in ((4, (0.23,)), (4, (0.01, 1.7)), (7, (0.05, 0.4, 3.0)), (20, (0.1, 2.5)),
                  (33, (0.9,)), (61, (0.003, 1.4)), (64, (0.5,))):
        S, D = qtop.random_reversible_rate_matrix(n)
        if n == 4:                             # (its atol of 1e-15 is for the n = 4 of its tests)
            qtop.assert_SD_reversible_rate_matrix(S, D)
        Q = np.dot(S, np.diag(D))
        A, lam, B = qtop.decompose_spectral_v2(S, D)
        case = dict(n=n, D=D.tolist(), A=A.tolist(), lam=lam.tolist(), B=B.tolist(), t=list(ts),
                    P_spectral=[qtop.getp_spectral_v2(D, A, lam, B, t).tolist() for t in ts])
        if n <= 20:                            # (kept small: the large cases carry P_spectral only)
            case['S'] = S.tolist()
            case['P_expm'] = [qtop.getp_rate_matrix(Q, t).tolist() for t in ts]
        cases.append(case)
    # the codon matrix of the p53 example (tests/golden/p53_mg94.json holds the matrix itself):
    # S = Q diag(1 / distn), D = distn
    mg = fixture_p53_mg94(mods)
    Q = np.array(mg['Q_offdiagonal'])
    Q -= np.diag(Q.sum(axis=1))
    D = np.array(mg['distn'])
    S = Q / D[None, :]
    S = 0.5 * (S + S.T)
    A, lam, B = qtop.decompose_spectral_v2(S, D)
    ts = (0.004, 0.31)
    cases.append(dict(
        n=len(D), mg94=True, D=D.tolist(), A=A.tolist(), lam=lam.tolist(), B=B.tolist(),
        t=list(ts), P_spectral=[qtop.getp_spectral_v2(D, A, lam, B, t).tolist() for t in ts],
        P_expm=[qtop.getp_rate_matrix(np.dot(S, np.diag(D)), ts[0]).tolist()]))
    return dict(cases=cases)


def fixture_tmjp_inhomogeneous(mods, ncases=12):
    """pyfelscore.tmjp_get_inhomogeneous_mjp (_tmjp_dense.py:1039-1054) through its sparse
    twin _tmjp.get_inhomogeneous_mjp (_tmjp.py:803-903): random primary trajectories (trees
    whose edges carry a primary state), random sparse primary rate matrices, every tolerance
    class.  Recorded per case: the tree in the reference's preorder CSR, the primary state of
    the edge above every node, and per class the 3 x 3 rate matrix of every edge (dense,
    diagonal = minus the row sum, keyed by the child's preorder index) and the allowed
    tolerance states of every node."""
    _tmjp = importlib.import_module('raoteh.sampler._tmjp')
    _density = mods['_density']
    if not hasattr(nx, 'to_numpy_matrix'):      # networkx >= 3 (as in fixture_p53_mg94)
        nx.to_numpy_matrix = lambda G, **kw: np.asmatrix(nx.to_numpy_array(G, **kw))
    rng = np.random.RandomState(77)
    cases = []
    for case in range(ncases):
        nprimary = int(rng.choice([2, 3, 5, 6]))
        nparts = int(rng.randint(1, min(3, nprimary) + 1))
        primary_to_part = dict((i, int(rng.randint(nparts))) for i in range(nprimary))
        Qd = rng.exponential(size=(nprimary, nprimary)) * (rng.uniform(size=(nprimary, nprimary)) < 0.6)
        np.fill_diagonal(Qd, 0.0)
        Q_nx = nx.DiGraph()
        Q_nx.add_nodes_from(range(nprimary))
        for a in range(nprimary):
            for b in range(nprimary):
                if Qd[a, b]:
                    Q_nx.add_edge(a, b, weight=float(Qd[a, b]))
        nnodes = int(rng.randint(2, 15))
        T = nx.Graph()
        T.add_node(0)
        for k in range(1, nnodes):
            T.add_edge(int(rng.randint(k)), k, weight=float(0.1 + rng.exponential()),
                       state=int(rng.randint(nprimary)))
        root = 0
        rate_on, rate_off = float(0.2 + rng.exponential()), float(0.2 + rng.exponential())
        preorder = list(nx.dfs_preorder_nodes(T, root))
        T_bfs = nx.DiGraph()
        T_bfs.add_node(root)
        for na, nb in nx.bfs_edges(T, root):
            T_bfs.add_edge(na, nb)
        idx, ptr = _density.digraph_to_bool_csr(T_bfs, preorder)
        index = dict((v, i) for i, v in enumerate(preorder))
        pred = dict((b, a) for a, b in nx.bfs_edges(T, root))
        edge_state = [0] * nnodes
        for b, a in pred.items():
            edge_state[index[b]] = T[a][b]['state']
        per_class = []
        for tol in range(nparts):
            T_tol, allowed = _tmjp.get_inhomogeneous_mjp(
                primary_to_part, rate_on, rate_off, Q_nx, T, root, tol)
            mats = np.zeros((nnodes, 3, 3))
            for b, a in pred.items():
                M = np.zeros((3, 3))            # (the graph may lack a state: no nodelist)
                for sa, sb, d in T_tol[a][b]['Q'].edges(data=True):
                    M[sa, sb] = d['weight']
                mats[index[b]] = M - np.diag(M.sum(axis=1))
            arr = [[1 if t in allowed[v] else 0 for t in (0, 1)] for v in preorder]
            per_class.append(dict(tolerance_class=tol, tol_rate_matrices=mats.tolist(),
                                  node_to_allowed_tolerances=arr))
        cases.append(dict(nprimary=nprimary, primary_to_part=[primary_to_part[i] for i in range(nprimary)],
                          Q_primary_offdiagonal=Qd.tolist(), rate_on=rate_on, rate_off=rate_off,
                          tree_csr_indices=[int(x) for x in idx], tree_csr_indptr=[int(x) for x in ptr],
                          edge_to_primary_state=[int(x) for x in edge_state], classes=per_class))
    return dict(cases=cases)


def fixture_switching(mods, nsites=5):
    """The 122-state "switching" model of examples/p53/liwen.py: MG94 codon process
    (create_mg94, parameters of get_jeff_params_e :273-300) x {reference, default},
    built step by step as liwen.py:566-636 builds it (the same reference helpers:
    create_mg94.create_mg94, _density.rate_matrix_to_numpy_array / dict_to_numpy_array,
    _util.get_normalized_dict_distn), on the p53 tree re-rooted at the leaf 'Has'
    (:476-477), leaf sets {c, 61 + c} (:682).  The disease table liwen.py reads is not
    in the reference tree, so each column gets a seeded benign set of amino acids (the
    residues seen in the column plus a random half of the others).  Likelihoods and the
    posterior probability that the ORIGINAL root is in the reference process
    (:403-415) come from the reference's unaccelerated type-y functions
    (_mcy.py:396-470,611-682, _mc0.py:89-138,202-252,382-462) with P = scipy expm."""
    sys.path.insert(0, REF + '/examples/p53')
    if not hasattr(nx, 'to_numpy_matrix'):
        nx.to_numpy_matrix = lambda G, **kw: np.asmatrix(nx.to_numpy_array(G, **kw))
    import create_mg94
    from raoteh_amd import io as rio
    _util, _density, _mc0 = mods['_util'], mods['_density'], mods['_mc0']
    here = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'p53')
    genetic_code = rio.read_genetic_code(os.path.join(here, 'universal.code.txt'))
    codon_to_state = dict((c, s) for s, r, c in genetic_code)
    nstates = len(genetic_code)
    states = list(range(nstates))
    # get_jeff_params_e (liwen.py:273-300) on the tree shipped with the p53 example
    rho = 0.61610
    AG = 0.50862
    CT = 1 - AG
    A = AG * 0.49373
    G = AG - A
    T = CT * 0.38884
    C = CT - T
    kappa = 3.38714
    omega = 0.37767
    Q, primary_distn, state_to_residue, residue_to_part = create_mg94.create_mg94(
        A, C, G, T, kappa, omega, genetic_code, target_expected_rate=1.0)
    Q_dense = _density.rate_matrix_to_numpy_array(Q, nodelist=states)
    with open(os.path.join(here, 'p53S.const.tree')) as f:
        tree, original_root, leaf_name_pairs = rio.read_newick(f.read())
    name_to_leaf = dict((name, leaf) for leaf, name in leaf_name_pairs)
    root = name_to_leaf['Has']
    name_codons = rio.read_phylip(os.path.join(here, 'alignment.for.codeml.phylip'))
    names = [nm for nm, _ in name_codons]
    columns = list(zip(*[cod for _, cod in name_codons]))
    residues = sorted(set(r for s, r, c in genetic_code))
    rng = np.random.RandomState(20131205)
    ncompound = 2 * nstates
    compound_states = list(range(ncompound))
    sites = []
    t0 = time.time()
    for i in range(nsites):
        column = [c.upper() for c in columns[i]]
        seen = set(state_to_residue[codon_to_state[c]] for c in column)
        others = [r for r in residues if r not in seen]
        extra = [r for r in others if rng.random_sample() < 0.5]
        benign_residues = seen | set(extra)
        if i == 1:                     # a column whose observed residue may be lethal somewhere
            benign_residues = set(extra) | set(sorted(seen)[:1])
        benign_states = set(s for s, r, c in genetic_code if r in benign_residues)
        # liwen.py:599-627
        Q_compound = nx.DiGraph()
        for sa, sb in Q.edges():
            weight = Q[sa][sb]['weight']
            Q_compound.add_edge(nstates + sa, nstates + sb, weight=weight)
        for sa, sb in Q.edges():
            weight = Q[sa][sb]['weight']
            if sb in benign_states:
                Q_compound.add_edge(sa, sb, weight=weight)
        for s in range(nstates):
            Q_compound.add_edge(s, nstates + s, weight=rho)
        compound_weights = {}
        for s in range(ncompound):
            if (s in primary_distn) and (s in benign_states):
                compound_weights[s] = primary_distn[s]
        compound_distn = _util.get_normalized_dict_distn(compound_weights)
        Q_compound_dense = _density.rate_matrix_to_numpy_array(
            Q_compound, nodelist=compound_states)
        compound_distn_dense = _density.dict_to_numpy_array(
            compound_distn, nodelist=compound_states)
        # liwen.py:677-682
        node_to_allowed_states = dict((n, set(compound_states)) for n in tree)
        for name, codon in zip(names, column):
            leaf = name_to_leaf[name]
            codon_state = codon_to_state[codon]
            node_to_allowed_states[leaf] = {codon_state, nstates + codon_state}
        # get_codon_site_inferences (:367-415), compound process, unaccelerated functions
        tmp = nx.Graph()
        for a, b, d in tree.edges(data=True):
            tmp.add_edge(a, b, weight=d['weight'])
        T_aug, dense = augmented(tmp, root, Q_compound_dense)
        rec = dict(column=column, benign_residues=sorted(benign_residues),
                   benign_states=sorted(int(s) for s in benign_states),
                   compound_distn=compound_distn_dense.tolist())
        try:
            pset, nset, pmap = ref_type_y(mods, T_aug, root, node_to_allowed_states)
            lik = _mc0.get_likelihood(pmap[root], root_distn=compound_distn)
            node_to_distn = _mc0.get_node_to_distn(T_aug, root, pmap, root_distn=compound_distn)
            d0 = node_to_distn[original_root]
            rec.update(likelihood=float(lik), log_likelihood=float(np.log(lik)),
                       p_reference=float(sum(p for s, p in d0.items() if s < nstates)),
                       root_pmap=[float(pmap[root].get(s, 0.0)) for s in compound_states],
                       original_root_distn=[float(d0.get(s, 0.0)) for s in compound_states])
        except _util.StructuralZeroProb:
            rec.update(likelihood=0.0, log_likelihood=None, p_reference=None)
        if i < 2:                      # the compound rate matrix itself, sparse
            nz = np.nonzero(Q_compound_dense)
            rec['Q_compound_nonzero'] = [[int(a), int(b), float(Q_compound_dense[a, b])]
                                         for a, b in zip(*nz)]
        if i == 0:
            nb0 = sorted(dense)[0]
            rec['P_scipy_node'] = int(nb0)
            rec['P_scipy_rows'] = [0, 60, 61, 121]
            rec['P_scipy'] = [dense[nb0][r].tolist() for r in (0, 60, 61, 121)]
        sites.append(rec)
    return dict(nstates=nstates, ncompound=ncompound, rho=rho, kappa=kappa, omega=omega,
                nt=dict(A=A, C=C, G=G, T=T), root=int(root), original_root=int(original_root),
                names=names, Q_default_offdiagonal_checksum=float(np.abs(Q_dense).sum()),
                primary_distn=[float(primary_distn[s]) for s in states],
                sites=sites, reference_seconds=time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE),
                                                  'tests', 'golden'))
    ap.add_argument('--only', default=None,
                    help='comma-separated fixture names (default: all)')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    mods = import_reference()
    meta = dict(generator='tools/gen_golden.py',
                reference='/root/reference (argriffing/raoteh)',
                python=sys.version.split()[0], numpy=np.__version__,
                scipy=scipy.__version__, networkx=nx.__version__)
    makers = dict(
        test_mjp_rerooting=lambda: fixture_test_mjp(mods),
        sum_to_one=lambda: fixture_sum_to_one(mods),
        kat_history=lambda: fixture_kat_history(mods),
        jukes_cantor=lambda: fixture_jukes_cantor(mods),
        random_sparse=lambda: fixture_random_sparse(mods),
        sparse_api=lambda: fixture_sparse_api(mods),
        p53_mg94=lambda: fixture_p53_mg94(mods),
        expectations=lambda: fixture_expectations(mods),
        expm=lambda: fixture_expm(mods),
        config_c1=lambda: fixture_config(mods, 'c1', 4, with_pmap=True),
        config_c2=lambda: fixture_config(mods, 'c2', 6),
        config_c3=lambda: fixture_config(mods, 'c3', 2),
        config_c5=lambda: fixture_config(mods, 'c5', 4),
        blinking=lambda: fixture_blinking(mods),
        forest=lambda: fixture_forest(mods),
        chunk_trees=lambda: fixture_chunk_trees(mods),
        spectral=lambda: fixture_spectral(mods),
        switching=lambda: fixture_switching(mods),
        tmjp_inhomogeneous=lambda: fixture_tmjp_inhomogeneous(mods),
    )
    only = args.only.split(',') if args.only else list(makers)
    fixtures = dict((name, makers[name]()) for name in only)
    for name, fx in fixtures.items():
        fx['_meta'] = meta
        path = os.path.join(args.out, name + '.json')
        with open(path, 'w') as f:
            json.dump(fx, f)
        print('%-22s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
