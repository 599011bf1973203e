"""
CPU tests of the sparse-API marshalling (raoteh_amd/_sparse.py, _mc0.py) and of
the oracle on the same inputs, against tests/golden/sparse_api.json (values
from the reference's _linalg.sparse_expm_naive, _mcx, _mcy unaccelerated twins,
_mc0 and _mcz).  No GPU: the device passes are replaced by the oracle's here,
which is exactly what the GPU test replaces back.
"""
import networkx as nx
import numpy as np
import pytest

from conftest import load_golden
from _sparse_cases import build, int_keys
from oracle import oracle_numpy as orc


def test_densified_problem_reproduces_reference_sets_and_pmaps():
    from raoteh_amd._sparse import SparseProblem
    fx = load_golden('sparse_api')
    for c in fx['cases']:
        T, root, _, allowed, _, root_distn = build(c, with_P=True)
        prob = SparseProblem(T, root)
        assert prob.sorted_states == sorted(c['labels'])
        ta = prob.ta
        mask = prob.mask_from_allowed(allowed)
        orc.mcy_esd_get_node_to_pset(ta.indices, ta.indptr, prob.esd, mask)
        # the accelerated backward pass keeps states the parent edge cannot
        # produce; the reference's pure-Python pset (_mcy.py:424-431) drops them
        # at once -- both agree after the forward pass
        got = prob.mask_to_dict(mask)
        for v, want in int_keys(c['y_pset']).items():
            assert set(want) <= got[v]
        orc.esd_get_node_to_set(ta.indices, ta.indptr, prob.esd, mask)
        assert prob.mask_to_dict(mask) == dict(
            (v, set(s)) for v, s in int_keys(c['y_set']).items())
        if c['y_zero']:
            continue
        pmap = np.empty(mask.shape)
        orc.mcy_esd_get_node_to_pmap(ta.indices, ta.indptr, prob.esd, mask, pmap)
        got = prob.pmap_to_dict(mask, pmap)
        for v, want in int_keys(c['y_pmap']).items():
            assert set(got[v]) == set(int(k) for k in want)
            for k, x in want.items():
                assert got[v][int(k)] == pytest.approx(x, rel=1e-12)
        from raoteh_amd import _mc0
        assert _mc0.get_likelihood(got[root], root_distn=root_distn) == \
            pytest.approx(c['y_likelihood'], rel=1e-12)


def test_sparse_rate_matrix_marshalling():
    from raoteh_amd import _mjp
    fx = load_golden('sparse_api')
    for c in fx['cases'][:6]:
        T, root, Q_default, _, _, _ = build(c, with_P=False)
        states, Qd = _mjp._dense_rate_matrix(Q_default)
        assert states == sorted(c['labels'])
        np.testing.assert_allclose(Qd.sum(axis=1), 0, atol=1e-15)
        keep = _mjp._reachability(Q_default, states)
        assert keep.diagonal().all()
        for na, nb in nx.bfs_edges(T, root):
            if str(nb) in c['edge_Q']:
                continue
            # the structural zeros of the reference's P are the unreachable pairs
            edges = set((a, b) for a, b, _ in c['P'][str(nb)])
            mine = set((states[i], states[j]) for i in range(len(states))
                       for j in range(len(states)) if keep[i, j])
            assert edges == mine


def test_dict_root_reduction_errors():
    from raoteh_amd import _mc0, StructuralZeroProb
    with pytest.raises(StructuralZeroProb):
        _mc0.get_likelihood({1: 0.5}, root_distn={})
    with pytest.raises(StructuralZeroProb):
        _mc0.get_likelihood({}, root_distn={1: 1.0})
    with pytest.raises(StructuralZeroProb):
        _mc0.get_likelihood({1: 0.5}, root_distn={2: 1.0})
    with pytest.raises(ValueError):
        _mc0.get_likelihood(None)
    assert _mc0.get_likelihood({1: 0.5, 7: 0.25}) == 0.75
    assert _mc0.get_likelihood({1: 0.5, 7: 0.25}, root_distn={7: 2.0}) == 0.5
