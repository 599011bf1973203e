"""Shared test helpers.  GPU tests are marked ``@pytest.mark.gpu``."""
import json
import os
import sys

import networkx as nx
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')


def load_golden(name):
    with open(os.path.join(GOLDEN, name + '.json')) as f:
        return json.load(f)


def tree_from_edges(edges, nodes=None):
    """Rebuild the nx.Graph in the fixture's edge insertion order (adjacency
    order decides the reference's preorder, so it must be preserved)."""
    T = nx.Graph()
    if nodes is not None:
        T.add_nodes_from(nodes)
    for a, b, w in edges:
        T.add_edge(int(a), int(b), weight=float(w))
    return T


def config_from_golden(fx):
    """Fixture -> (T, root, nstates, Q_default, root_distn, list of per-site
    node_to_allowed_states dicts)."""
    T = tree_from_edges(fx['edges'])
    root = fx['root']
    n = fx['nstates']
    Q_default = None
    if 'Q_default' in fx:
        Q_default = np.array(fx['Q_default'])
    else:
        for na, nb in nx.bfs_edges(T, root):
            T[na][nb]['Q'] = np.array(fx['Q_edges'][str(nb)])
    sites = []
    for row in fx['leaf_states']:
        d = dict((v, set(range(n))) for v in T)
        for leaf, s in zip(fx['leaves'], row):
            if fx['obs_kind'] == 'state':
                d[leaf] = {int(s)}
            else:
                d[leaf] = set(fx['leaf_allowed'][int(s)])
        sites.append(d)
    return T, root, n, Q_default, np.array(fx['root_distn']), sites


def expectation_cases():
    """tests/golden/expectations.json -> list of (label, T, allowed, root, nstates,
    root_distn or None, Q_default, want) with want = dict(dwell, init, trans
    [, closed_form_dwell]) as produced by the reference (tools/gen_golden.py)."""
    fx = load_golden('expectations')
    out = []
    jc = fx['jukes_cantor']
    n = jc['nstates']
    Q = np.array(jc['Q'])
    for r in jc['rows']:
        T = tree_from_edges(jc['edges'])
        allowed = dict((v, set(range(n))) for v in T)
        allowed[0] = {r['a']}
        allowed[4] = {r['b']}
        out.append(('jc a=%d b=%d root=%d' % (r['a'], r['b'], r['root']), T, allowed,
                    r['root'], n, None, Q, r))
    for k, c in enumerate(fx['cases']):
        n = c['nstates']
        mats = [np.array(m) for m in c['Q']]
        T = nx.Graph()
        for a, b, w, q in c['edges']:
            T.add_edge(int(a), int(b), weight=float(w))
            if q:
                T[int(a)][int(b)]['Q'] = mats[q]
        allowed = dict((int(v), set(ss)) for v, ss in c['allowed'].items())
        out.append(('case %d' % k, T, allowed, c['root'], n, np.array(c['root_distn']),
                    mats[0], c))
    return out


def switching_cases():
    """tests/golden/switching.json (the 122-state model of examples/p53/liwen.py:599-621)
    rebuilt from the p53 data files with this repository's builders.  -> (fixture, list of
    dict(T, root, original_root, nstates, ncompound, Q_default, primary_distn, Q_compound,
    compound_distn, allowed, want)) where want is the reference's record of the site."""
    from raoteh_amd import io as rio, synth
    fx = load_golden('switching')
    here = os.path.join(GOLDEN, 'p53')
    code = rio.read_genetic_code(os.path.join(here, 'universal.code.txt'))
    codon_to_state = dict((c, s) for s, _, c in code)
    n = fx['nstates']
    Q_default, primary = rio.mg94_from_code(code, fx['kappa'], fx['omega'], fx['nt'])
    with open(os.path.join(here, 'p53S.const.tree')) as f:
        T, original_root, leaf_name_pairs = rio.read_newick(f.read())
    name_to_leaf = dict((name, leaf) for leaf, name in leaf_name_pairs)
    root = name_to_leaf['Has']
    assert (root, original_root) == (fx['root'], fx['original_root'])
    cases = []
    for rec in fx['sites']:
        Qc, dc = synth.switching_model(Q_default, primary, rec['benign_states'], fx['rho'])
        allowed = dict((v, set(range(2 * n))) for v in T)
        for name, codon in zip(fx['names'], rec['column']):
            allowed[name_to_leaf[name]] = set(
                synth.switching_allowed_states(codon_to_state[codon], n))
        cases.append(dict(T=T, root=root, original_root=original_root, nstates=n,
                          ncompound=2 * n, Q_default=Q_default, primary_distn=primary,
                          Q_compound=Qc, compound_distn=dc, allowed=allowed, want=rec))
    return fx, cases


@pytest.fixture(scope='session')
def golden():
    return load_golden
