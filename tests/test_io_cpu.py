"""
CPU tests of raoteh_amd/io.py on the reference's own p53 example data
(tests/golden/p53/, copied input data) and of the oracle on it.
"""
import os

import networkx as nx
import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import oracle_numpy as orc

P53 = os.path.join(GOLDEN, 'p53')


def p53_problem():
    from raoteh_amd import io
    al = io.read_phylip(os.path.join(P53, 'alignment.for.codeml.phylip'))
    T, root, leaf_name_pairs = io.read_newick(
        open(os.path.join(P53, 'p53S.const.tree')).read())
    code = io.read_genetic_code(os.path.join(P53, 'universal.code.txt'))
    fx = load_golden('p53_mg94')
    Q, distn = io.mg94_from_code(code, fx['kappa'], fx['omega'], fx['nt'])
    leaves, states = io.alignment_to_states(al, code, leaf_name_pairs)
    return T, root, leaves, states, Q, distn, code, fx


def test_readers_on_the_p53_example():
    T, root, leaves, states, Q, distn, code, fx = p53_problem()
    assert nx.is_tree(T) and T.number_of_nodes() == 49 and root == 48
    assert sorted(leaves) == list(range(25))           # leaves numbered first
    assert all(d['weight'] == 0.1 for _, _, d in T.edges(data=True))
    assert T.degree(root) == 2 and sum(1 for v in T if T.degree(v) == 1) == 25
    assert states.shape == (393, 25) and states.max() < 61
    assert [s for s, _, _ in code] == list(range(61))
    # first codon of every p53 sequence is ATG
    atg = [s for s, _, c in code if c == 'ATG'][0]
    assert (states[0] == atg).all()


def test_mg94_matches_the_reference_builder():
    _, _, _, _, Q, distn, _, fx = p53_problem()
    want = np.array(fx['Q_offdiagonal'])
    want -= np.diag(want.sum(axis=1))
    np.testing.assert_allclose(Q, want, rtol=0, atol=1e-15)
    np.testing.assert_allclose(distn, fx['distn'], rtol=0, atol=1e-16)
    assert -np.dot(distn, np.diag(Q)) == pytest.approx(1.0, abs=1e-15)
    np.testing.assert_allclose(distn @ Q, 0, atol=1e-16)   # stationary


def test_newick_reader_details():
    from raoteh_amd import io
    T, root, pairs = io.read_newick('((a:1.5,b:2)x:0.25,c:3,(d,e:1e-1):4);')
    assert pairs == [(0, 'a'), (1, 'b'), (2, 'c'), (3, 'd'), (4, 'e')]
    assert root == 7 and T.degree(root) == 3
    assert T[5][0]['weight'] == 1.5 and T[5][1]['weight'] == 2.0
    assert T[7][5]['weight'] == 0.25 and T[7][2]['weight'] == 3.0
    assert T[6][3]['weight'] == 1.0 and T[6][4]['weight'] == 0.1     # missing length -> 1
    with pytest.raises(ValueError):
        io.read_newick('((a,b),c)')


def test_pattern_compression_round_trip():
    from raoteh_amd import io
    _, _, _, states, _, _, _, _ = p53_problem()
    unique, inverse, counts = io.compress_patterns(states)
    assert unique.shape[0] == 381 and counts.sum() == 393
    np.testing.assert_array_equal(unique[inverse], states)
    rng = np.random.RandomState(3)
    dense = rng.randint(0, 3, size=(50, 4, 5)).astype(np.float64)
    u, inv, c = io.compress_patterns(dense)
    np.testing.assert_array_equal(u[inv], dense)
    assert len(u) == len(set(map(bytes, dense)))


def test_oracle_p53_total_log_likelihood_is_stable():
    """No expected value exists in the reference; this pins the
    oracle's own number so that a change of either side shows up, and checks that
    duplicate columns get identical values."""
    T, root, leaves, states, Q, distn, _, _ = p53_problem()
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, root, 61, Q_default=Q)
    dense = np.zeros((393, 25, 61))
    ii, kk = np.indices(states.shape)
    dense[ii, kk, states] = 1.0
    ll, st = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in leaves],
                                       dense, distn)
    assert (st == 0).all()
    assert ll.sum() == pytest.approx(-11202.4288003113, rel=1e-12)
    from raoteh_amd import io
    unique, inverse, counts = io.compress_patterns(states)
    for k in np.flatnonzero(counts > 1)[:5]:
        rows = np.flatnonzero(inverse == k)
        assert np.ptp(ll[rows]) == 0.0


def test_rate_matrix_text_format_round_trip(tmp_path):
    """craoteh/README.rst:1-10: N on the first line, `source sink rate` triples, missing
    entries zero, diagonal from the row sums."""
    import io as _io
    from raoteh_amd import io as rio, synth
    text = '3\n0\t1\t0.5\n1 2 2.0\n\n2\t0\t1e-3  # a comment\n'
    Q = rio.read_rate_matrix(_io.StringIO(text))
    np.testing.assert_array_equal(Q, [[-0.5, 0.5, 0.0], [0.0, -2.0, 2.0], [1e-3, 0.0, -1e-3]])
    Qm, _ = synth.mg94()
    path = str(tmp_path / 'mg94.rates')
    rio.write_rate_matrix(Qm, path)
    back = rio.read_rate_matrix(path)
    off = ~np.eye(len(Qm), dtype=bool)
    np.testing.assert_array_equal(back[off], Qm[off])
    np.testing.assert_allclose(np.diag(back), np.diag(Qm), rtol=1e-13)
    for bad in ('', 'x\n', '2\n0 2 1.0\n', '2\n0 0 1.0\n', '2\n0 1 -1\n', '2\n0 1 1\n0 1 2\n',
                '2\n0 1\n', '2\n0 1 nan\n'):
        with pytest.raises(ValueError):
            rio.read_rate_matrix(_io.StringIO(bad))
