"""Pins oracle/oracle.c (the C restatement timed as the CPU baseline) to the
golden vectors generated from the reference.  CPU only."""
import networkx as nx
import numpy as np
import pytest

from conftest import load_golden, config_from_golden
from oracle import oracle_c, oracle_numpy as orc
from raoteh_amd._tree import TreeArrays


def test_c_expm_matches_scipy_fixture():
    fx = load_golden('expm')
    for row in fx['rows']:
        Q = np.array(row['Q'])
        want = np.array(row['P'])
        got, info = oracle_c.expm(Q, row['t'])
        np.testing.assert_allclose(got, want, rtol=1e-9,
                                   atol=1e-14 * max(1.0, np.abs(want).max()))
        assert info == orc.pade_order_and_squarings(
            np.abs(Q * row['t']).sum(axis=0).max())


@pytest.mark.parametrize('name', ['c1', 'c2', 'c3', 'c5'])
def test_c_batch_matches_reference_golden(name):
    fx = load_golden('config_' + name)
    T, root, n, Q_default, distn, sites = config_from_golden(fx)
    ta = TreeArrays(T, root)
    Q, node_q = ta.rate_matrices(n, Q_default)
    t = ta.branch_lengths()
    obs_nodes = [ta.node_to_index[v] for v in fx['leaves']]
    obs = np.zeros((len(sites), len(obs_nodes), n))
    for i, d in enumerate(sites):
        for k, v in enumerate(fx['leaves']):
            obs[i, k, sorted(d[v])] = 1.0
    ll, st = oracle_c.batch_loglik_faithful(ta.indices, ta.indptr, Q, node_q, t,
                                            obs_nodes, obs, distn)
    assert not st.any()
    np.testing.assert_allclose(ll, fx['log_likelihoods'], rtol=1e-11)
    _, _, _, esd = orc.get_expm_augmented_transitions(T, root, n, Q_default)
    ll2, _ = oracle_c.batch_loglik(ta.indices, ta.indptr, esd, obs_nodes, obs, distn)
    np.testing.assert_allclose(ll2, fx['log_likelihoods'], rtol=1e-11)


def test_c_zero_probability_site():
    fx = load_golden('random_sparse')
    c = [c for c in fx['cases'] if c['zero']][0]
    n = c['nstates']
    from conftest import tree_from_edges
    T = tree_from_edges(c['edges'], nodes=c['nodes'])
    for na, nb in nx.bfs_edges(T, c['root']):
        T[na][nb]['P'] = np.array(c['P'][str(nb)])
    pre, idx, ptr, esd = orc.get_esd_transitions(T, c['root'], n)
    obs = np.zeros((1, len(pre), n))
    for i, v in enumerate(pre):
        obs[0, i, c['allowed'][str(v)]] = 1.0
    ll, st = oracle_c.batch_loglik(idx, ptr, esd, list(range(len(pre))), obs,
                                   np.array(c['root_distn']))
    assert st[0] == 1 and np.isneginf(ll[0])
