"""
Pins the oracle (oracle/oracle_numpy.py) to the reference: every golden vector
in tests/golden/ was produced by the reference's own pure-Python code
(tools/gen_golden.py), plus the literal known answers of its test-suite.
CPU only.
"""
import itertools

import networkx as nx
import numpy as np
import pytest

from conftest import (load_golden, tree_from_edges, config_from_golden, expectation_cases,
                      switching_cases)
from oracle import oracle_numpy as orc

RTOL = 1e-12


def test_rerooting_known_answer():
    # reference tests/test_mjp.py:91-164; SURVEY known answer 0.002296828148732273
    fx = load_golden('test_mjp_rerooting')
    T = tree_from_edges(fx['edges'])
    Q = np.array(fx['Q'])
    distn = np.array(fx['root_distn'])
    n = fx['nstates']
    allowed = dict((v, set(range(n))) for v in T)
    for k, s in fx['node_to_state'].items():
        allowed[int(k)] = {s}
    for r in fx['rootings']:
        lk = orc.mjp_dense_get_likelihood(T, allowed, r['root'], n,
                                          root_distn=distn, Q_default=Q)
        assert lk == pytest.approx(r['likelihood'], rel=RTOL)
        assert lk == pytest.approx(0.002296828148732273, rel=1e-12)
        assert lk == pytest.approx(r['marginalised'], rel=1e-12)


def test_sum_to_one():
    # reference tests/test_mjp.py:52-89
    fx = load_golden('sum_to_one')
    T = tree_from_edges(fx['edges'])
    Q = np.array(fx['Q'])
    distn = np.array(fx['root_distn'])
    total = 0.0
    for assignment, want in zip(fx['assignments'], fx['likelihoods']):
        allowed = dict((v, {s}) for v, s in enumerate(assignment))
        lk = orc.mjp_dense_get_likelihood(T, allowed, 0, 3, root_distn=distn,
                                          Q_default=Q)
        assert lk == pytest.approx(want, rel=RTOL)
        total += lk
    assert total == pytest.approx(1.0, rel=1e-12)


def test_kat_four_log_half():
    # reference tests/test_mc.py:131-150: fully observed history
    fx = load_golden('kat_history')
    assert fx['history_log_likelihood'] == 4 * np.log(0.5)
    T = tree_from_edges(fx['edges'])
    P = np.array(fx['P'])
    for a, b in T.edges():
        T[a][b]['P'] = P
    allowed = dict((int(k), {v}) for k, v in fx['node_to_state'].items())
    lk = orc.mcy_dense_get_likelihood(T, 0, 3, node_to_allowed_states=allowed,
                                      root_distn=np.array(fx['root_distn']))
    assert np.log(lk) == pytest.approx(4 * np.log(0.5), rel=1e-15)


def test_jukes_cantor_closed_form():
    # reference _conditional_expectation.py:15-33
    fx = load_golden('jukes_cantor')
    for row in fx['rows']:
        n, t = row['n'], row['t']
        Q = np.full((n, n), 1.0 / (n - 1))
        np.fill_diagonal(Q, 0)
        Q -= np.diag(Q.sum(axis=1))
        for expm in (orc.custom_expm, orc.expm_pade):
            P = expm(Q, t)
            assert P[0, 0] == pytest.approx(row['p_same'], rel=1e-12)
            assert P[0, 1] == pytest.approx(row['p_diff'], rel=1e-12)


def test_random_sparse_trees_type_y_and_z():
    # set-up of reference tests/test_mc.py:52-102; expected values from the
    # reference's unaccelerated pset/set/pmap and _mcz pmap
    fx = load_golden('random_sparse')
    assert any(c['zero'] for c in fx['cases'])
    for c in fx['cases']:
        n = c['nstates']
        T = tree_from_edges(c['edges'], nodes=c['nodes'])
        root = c['root']
        for na, nb in nx.bfs_edges(T, root):
            T[na][nb]['P'] = np.array(c['P'][str(nb)])
        allowed = dict((int(k), set(v)) for k, v in c['allowed'].items())
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        mask = orc.define_state_mask(allowed, pre, n)
        m1 = orc.mcy_esd_get_node_to_pset(idx, ptr, esd, mask.copy())
        for i, v in enumerate(pre):
            assert set(np.flatnonzero(m1[i])) == set(c['pset'][str(v)])
        m2 = orc.esd_get_node_to_set(idx, ptr, esd, m1.copy())
        for i, v in enumerate(pre):
            assert set(np.flatnonzero(m2[i])) == set(c['set'][str(v)])
        pmap = orc.mcy_esd_get_node_to_pmap(idx, ptr, esd, m2)
        for i, v in enumerate(pre):
            np.testing.assert_allclose(pmap[i], c['pmap'][str(v)],
                                       rtol=RTOL, atol=0)
        distn = np.array(c['root_distn'])
        if c['zero']:
            with pytest.raises(orc.StructuralZeroProb):
                orc.mc0_get_likelihood(pmap[0], root_distn=distn)
        else:
            lk = orc.mc0_get_likelihood(pmap[0], root_distn=distn)
            assert lk == pytest.approx(c['likelihood'], rel=RTOL)
        # downward pass + joint endpoint distributions (reference _mc0.py:255-308,
        # 382-462 / _mc0_dense.py:217-270,400-489)
        if 'distn' in c:
            dn = orc.mc0_esd_get_node_to_distn(idx, ptr, esd, distn, pmap)
            J = orc.mc0_esd_get_joint_endpoint_distn(idx, ptr, esd, pmap, dn)
            for i, v in enumerate(pre):
                np.testing.assert_allclose(dn[i], c['distn'][str(v)], rtol=1e-12,
                                           atol=1e-300)
                assert dn[i].sum() == pytest.approx(1.0, rel=1e-12)
                if str(v) in c['joint']:
                    np.testing.assert_allclose(J[i], np.array(c['joint'][str(v)]),
                                               rtol=1e-12, atol=1e-300)
        # type-z (reference _mcz.py:140-163)
        obs = np.array([c['obs_lik'][str(v)] for v in pre])
        pz = orc.mcy_esd_get_node_to_pmap(idx, ptr, esd, m2, obs_lik=obs)
        for i, v in enumerate(pre):
            np.testing.assert_allclose(pz[i], c['pmap_z'][str(v)],
                                       rtol=RTOL, atol=0)


@pytest.mark.parametrize('name', ['c1', 'c2', 'c3', 'c5'])
def test_config_fixtures(name):
    fx = load_golden('config_' + name)
    T, root, n, Q_default, distn, sites = config_from_golden(fx)
    # single-site reference-shaped entry point
    for allowed, want in zip(sites[:2], fx['likelihoods']):
        lk = orc.mjp_dense_get_likelihood(T, allowed, root, n,
                                          root_distn=distn,
                                          Q_default=Q_default)
        assert lk == pytest.approx(want, rel=1e-11)
    # batched form
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(
        T, root, n, Q_default=Q_default)
    for k, P in fx['P_scipy'].items():
        np.testing.assert_allclose(esd[pre.index(int(k))], np.array(P),
                                   rtol=1e-12, atol=1e-300)
    obs_nodes = [pre.index(v) for v in fx['leaves']]
    obs = np.zeros((len(sites), len(obs_nodes), n))
    for i, allowed in enumerate(sites):
        for k, v in enumerate(fx['leaves']):
            obs[i, k, sorted(allowed[v])] = 1.0
    ll, status = orc.batch_log_likelihoods(idx, ptr, esd, obs_nodes, obs,
                                           root_distn=distn)
    assert not status.any()
    np.testing.assert_allclose(ll, fx['log_likelihoods'], rtol=1e-11)
    if 'pmaps' in fx:
        mask = orc.define_state_mask(sites[0], pre, n)
        _, pmap = orc.esd_get_node_to_pmap(idx, ptr, esd, mask)
        for i, v in enumerate(pre):
            np.testing.assert_allclose(pmap[i], fx['pmaps'][0][str(v)],
                                       rtol=1e-11, atol=0)


def test_expm_pade_matches_scipy_fixture():
    # reference tests/test_expm.py:20-82 matrix families + model matrices
    fx = load_golden('expm')
    worst = 0.0
    for row in fx['rows']:
        Q = np.array(row['Q'])
        want = np.array(row['P'])
        got = orc.expm_pade(Q, row['t'])
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        worst = max(worst, err)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(orc.custom_expm(Q, row['t']), want,
                                   rtol=1e-13, atol=1e-300)
    assert worst < 1e-13


def test_single_node_tree_and_errors():
    # reference _mcy_dense.py:472-487, _mjp_dense.py:397-398
    T = nx.Graph()
    T.add_node(7)
    Q = np.array([[-1.0, 1.0], [2.0, -2.0]])
    assert orc.mjp_dense_get_likelihood(T, {7: {0}}, 7, 2, None, Q) == 1
    lk = orc.mjp_dense_get_likelihood(T, {7: {0, 1}}, 7, 2,
                                      np.array([0.25, 0.5]), Q)
    assert lk == pytest.approx(0.75)
    with pytest.raises(orc.StructuralZeroProb):
        orc.mjp_dense_get_likelihood(T, {7: set()}, 7, 2, None, Q)
    with pytest.raises(ValueError):
        orc.mjp_dense_get_likelihood(T, {7: {0}}, 8, 2, None, Q)


def test_expected_history_statistics():
    """_mjp_dense.get_expected_history_statistics (:410-539) restated in the oracle
    against the reference's own outputs (its sparse twin, _mjp.py:431-595, which
    tests/test_mjp.py:166-237 pins the dense one to) and against the Jukes-Cantor
    closed form that test uses as the known answer."""
    cases = expectation_cases()
    assert len(cases) >= 50
    for label, T, allowed, root, n, distn, Q, want in cases:
        dwell, init, trans = orc.mjp_dense_get_expected_history_statistics(
            T, allowed, root, n, root_distn=distn, Q_default=Q)
        np.testing.assert_allclose(dwell, want['dwell'], rtol=1e-10, atol=1e-14,
                                   err_msg=label)
        np.testing.assert_allclose(init, want['init'], rtol=1e-12, atol=1e-15,
                                   err_msg=label)
        off = ~np.eye(n, dtype=bool)
        np.testing.assert_allclose(trans[off], np.array(want['trans'])[off], rtol=1e-10,
                                   atol=1e-14, err_msg=label)
        if 'closed_form_dwell' in want:
            np.testing.assert_allclose(dwell, want['closed_form_dwell'], rtol=1e-10,
                                       atol=1e-14, err_msg=label)
        # dwell times add up to the tree length whatever the data
        assert dwell.sum() == pytest.approx(
            sum(d['weight'] for _, _, d in T.edges(data=True)), rel=1e-10)


def test_expm_taylor_matches_scipy_fixture():
    # the restated algorithm of the device's default expm kernel for n > 4 (round 2:
    # Taylor / Paterson-Stockmeyer) against the stored scipy.linalg.expm matrices
    fx = load_golden('expm')
    for r in fx['rows']:
        want = np.array(r['P'])
        got = orc.expm_taylor(np.array(r['Q']), r['t'])
        np.testing.assert_allclose(got, want, rtol=1e-10,
                                   atol=1e-14 * max(1.0, np.abs(want).max()),
                                   err_msg='%s t=%g' % (r['form'], r['t']))
    for nrm, want in ((0.0, (3, 0)), (1e-5, (3, 0)), (9e-3, (6, 0)), (0.08, (9, 0)),
                      (0.29, (12, 0)), (0.64, (15, 0)), (0.65, (15, 1)), (1.28, (15, 1)),
                      (1.29, (15, 2)), (100.0, (15, 8))):
        assert orc.taylor_order_and_squarings(nrm) == want


def test_blinking_builder_matches_the_reference_builder():
    # tests/golden/blinking.json: rate matrix, root distribution and allowed compound
    # states produced by the reference's own examples/code2x3/run.py:329-461
    # (do_blinking_process) for the (nprimary = 5, nparts = 2) model of config 5
    from raoteh_amd import synth
    fx = load_golden('blinking')
    p2p = dict((int(k), v) for k, v in fx['primary_to_part'].items())
    Qp, dp = np.array(fx['Q_primary']), np.array(fx['primary_distn'])
    for r in fx['rows']:
        Q, distn = synth.blinking_model(Qp, dp, p2p, r['rate_on'], r['rate_off'])
        assert np.array_equal(Q, np.array(r['Q']))
        assert np.array_equal(distn, np.array(r['root_distn']))
        assert Q.shape[0] == r['nstates'] == 20
        for leaf, c in zip(r['leaves'], r['leaf_primary']):
            assert sorted(synth.blinking_allowed_states(c, fx['nprimary'], p2p)) == \
                r['allowed'][str(leaf)]
        # the reference restricts unobserved nodes to the compound states whose primary
        # state is tolerated; config 5 leaves them unrestricted.  Same likelihood: the
        # other states have zero prior and no rate leads into them.
        inner = r['allowed']['0']
        dead = sorted(set(range(20)) - set(inner))
        assert np.all(distn[dead] == 0) and np.all(Q[np.ix_(inner, dead)] == 0)
        T = nx.Graph()
        for a, b in ((0, 1), (1, 3), (1, 4), (0, 2), (2, 5), (2, 6)):
            T.add_edge(a, b, weight=0.1)
        liks = []
        for restrict in (True, False):
            allowed = dict((v, set(inner) if restrict else set(range(20))) for v in T)
            for leaf in r['leaves']:
                allowed[leaf] = set(r['allowed'][str(leaf)])
            liks.append(orc.mjp_dense_get_likelihood(T, allowed, 0, 20, root_distn=distn,
                                                     Q_default=Q))
        assert liks[0] == pytest.approx(liks[1], rel=1e-14) and liks[0] > 0


def test_chunk_forest_against_the_reference_chunk_trees():
    """raoteh_amd._sampler.chunk_forest (vectorised over chains) against the reference's
    _graph_transform.get_chunk_tree_type_b on histories with degree-two event nodes
    (tests/golden/chunk_trees.json, written by tools/gen_golden.py): the same partition
    of the history's edges into chunks and the same chunk-tree edges, up to the naming of
    the chunks (the reference numbers them in BFS order of the history, the batch in
    preorder of the base edges).  All cases go through ONE call as a batch of chains on
    a forest of different base trees is not what chunk_forest takes, so per base tree."""
    import json
    import os
    from raoteh_amd import _sampler
    path = os.path.join(os.path.dirname(__file__), 'golden', 'chunk_trees.json')
    cases = json.load(open(path))['cases']
    assert len(cases) >= 20
    for rec in cases:
        parent = np.array(rec['parent'], dtype=np.int64)
        counts = np.array(rec['counts'], dtype=np.int64)
        N = parent.shape[0]
        # two chains with the same history: the batch dimension must not mix them
        edge1 = np.repeat(np.arange(1, N), counts[1:]).astype(np.int64)
        chain = np.concatenate([np.zeros_like(edge1), np.ones_like(edge1)])
        edge = np.concatenate([edge1, edge1])
        offset, cparent, piece, node = _sampler.chunk_forest(parent, 2, chain, edge)
        nch = len(rec['chunk_nodes'])
        assert offset.tolist() == [0, nch, 2 * nch]
        ref_chunk = dict(((a, b), c) for a, b, c in rec['edge_to_chunk'])
        for c in range(2):
            lo = int(offset[c])
            to_mine = {}
            for row, (v, k, na, nb) in enumerate(rec['pieces']):
                mine = int(piece[c * len(rec['pieces']) + row]) - lo
                assert to_mine.setdefault(ref_chunk[(na, nb)], mine) == mine
            assert sorted(to_mine.values()) == list(range(nch))        # a bijection
            got = sorted(sorted((i, int(cparent[lo + i]))) for i in range(1, nch))
            want = sorted(sorted((to_mine[a], to_mine[b])) for a, b in rec['chunk_edges'])
            assert got == want
            # the root's chunk is chunk 0 in both numberings, and base nodes sit in the
            # chunk of the piece that ends at them
            assert to_mine[rec['chunk_nodes'][0]] == 0 and int(node[c, 0]) == 0
            for row, (v, k, na, nb) in enumerate(rec['pieces']):
                if nb == v:
                    assert int(node[c, v]) == int(piece[c * len(rec['pieces']) + row]) - lo


def test_spectral_restatement_matches_the_reference_qtop():
    """tests/golden/spectral.json: examples/p53/qtop.py's own decomposition and
    reconstruction of its own random reversible rate matrices (first state of zero
    stationary probability) and of the p53 codon matrix; next to scipy's expm of the same
    matrix -- what qtop.py:587-609 (test_spectral_v2_expm) compares, at its atol."""
    fx = load_golden('spectral')
    for c in fx['cases']:
        D, A, lam, B = (np.array(c[k]) for k in ('D', 'A', 'lam', 'B'))
        for k, t in enumerate(c['t']):
            got = orc.spectral_getp_v2(D, A, lam, B, t)
            np.testing.assert_allclose(got, np.array(c['P_spectral'][k]), rtol=1e-13, atol=1e-15)
            if k < len(c.get('P_expm', ())):
                np.testing.assert_allclose(got, np.array(c['P_expm'][k]), rtol=0,
                                           atol=1e-14 if c['n'] == 4 else 1e-12)
        if 'S' in c:
            # eigenvectors are fixed up to sign (and order within an eigenspace): compare
            # what the factors reconstruct
            A2, lam2, B2 = orc.spectral_decompose_v2(np.array(c['S']), D)
            np.testing.assert_allclose(np.sort(lam2), np.sort(lam), rtol=1e-10, atol=1e-12)
            for t in c['t']:
                np.testing.assert_allclose(orc.spectral_getp_v2(D, A2, lam2, B2, t),
                                           orc.spectral_getp_v2(D, A, lam, B, t),
                                           rtol=1e-10, atol=1e-13)


def test_switching_model_builder_matches_liwen():
    # examples/p53/liwen.py:599-627 run by tools/gen_golden.py with the reference's own
    # helpers: the compound rate matrix and prior of synth.switching_model are those
    fx, cases = switching_cases()
    assert fx['ncompound'] == 122
    for c in cases:
        want = c['want']
        np.testing.assert_allclose(c['compound_distn'], want['compound_distn'],
                                   rtol=1e-15, atol=0)
        if 'Q_compound_nonzero' in want:
            Q = np.zeros((122, 122))
            for a, b, v in want['Q_compound_nonzero']:
                Q[a, b] = v
            # io.mg94_from_code vs create_mg94: the same numbers to rounding
            np.testing.assert_allclose(c['Q_compound'], Q, rtol=1e-13, atol=0)
            assert ((c['Q_compound'] != 0) == (Q != 0)).all()


def test_switching_oracle_matches_reference():
    # 122 states: likelihood, StructuralZeroProb and the posterior probability that the
    # original root is in the reference process (liwen.py:367-415)
    fx, cases = switching_cases()
    seen_zero = False
    for c in cases:
        want = c['want']
        args = (c['T'], c['allowed'], c['root'], c['ncompound'])
        if want['log_likelihood'] is None:
            with pytest.raises(orc.StructuralZeroProb):
                orc.mjp_dense_get_likelihood(*args, root_distn=c['compound_distn'],
                                             Q_default=c['Q_compound'])
            seen_zero = True
            continue
        lk = orc.mjp_dense_get_likelihood(*args, root_distn=c['compound_distn'],
                                          Q_default=c['Q_compound'])
        assert np.log(lk) == pytest.approx(want['log_likelihood'], rel=1e-11)
        preorder, indices, indptr, esd = orc.get_expm_augmented_transitions(
            c['T'], c['root'], c['ncompound'], Q_default=c['Q_compound'])
        if 'P_scipy' in want:
            k = preorder.index(want['P_scipy_node'])
            np.testing.assert_allclose(esd[k][want['P_scipy_rows']], want['P_scipy'],
                                       rtol=1e-12, atol=1e-300)
        mask = orc.define_state_mask(c['allowed'], preorder, c['ncompound'])
        _, pmap = orc.esd_get_node_to_pmap(indices, indptr, esd, mask)
        np.testing.assert_allclose(pmap[0], want['root_pmap'], rtol=1e-10, atol=1e-300)
        distn = orc.mc0_esd_get_node_to_distn(indices, indptr, esd, c['compound_distn'], pmap)
        d0 = distn[preorder.index(c['original_root'])]
        np.testing.assert_allclose(d0, want['original_root_distn'], rtol=1e-9, atol=1e-18)
        assert d0[:61].sum() == pytest.approx(want['p_reference'], rel=1e-11)
    assert seen_zero


def test_lower_bound_transition_matrix_is_the_stated_integral():
    # liwen.py:59-77 states what an off-diagonal entry is: the integral over the time x of the
    # one change of exp(-ra x) rab exp(-rb (t - x)); the bound never exceeds expm(Q t)
    from scipy.integrate import quad
    rng = np.random.RandomState(3)
    for n in (2, 5):
        Q = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) < 0.7)
        np.fill_diagonal(Q, 0.0)
        Q[0, 1] = 0.5
        Q -= np.diag(Q.sum(axis=1))
        if n == 5:
            Q[1, 1] = Q[0, 0]                        # the ra == rb branch (:69-70)
        for t in (0.05, 0.7):
            P = orc.getp_lb(Q, t)
            for a in range(n):
                for b in range(n):
                    if a == b:
                        want = np.exp(t * Q[a, a])
                    else:
                        want = quad(lambda x: np.exp(Q[a, a] * x) * Q[a, b] *
                                    np.exp(Q[b, b] * (t - x)), 0.0, t, epsabs=0, epsrel=1e-13)[0]
                    assert P[a, b] == pytest.approx(want, rel=1e-11, abs=1e-300)
            if n == 2:
                assert (P <= orc.custom_expm(Q, t) + 1e-15).all()
