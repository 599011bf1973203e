"""
Deterministic synthetic inputs for the five BASELINE.json configurations
(SURVEY.md section 8d).  Host-side numpy only; nothing here touches the GPU.

C1  4-state,  8-leaf balanced tree, 1 site           (plumbing)
C2  4-state HKY85, 64-leaf, 100 000 sites            (HBM-bound headline)
C3  61-state MG94 codon model, 64-leaf, 10 000 sites (f64 MFMA)
C4  as C3 with 1 000 000 sites sharded over 8 GPUs
C5  20-state blinking (tolerance) compound process, 32-leaf, 50 000 sites,
    a different rate matrix on every edge

The model builders are written from the model definitions (HKY85; Muse-Gaut 94
as parameterised in the reference's examples/p53/create_mg94.py:23-142; the
blinking process of examples/code2x3/run.py:341-434), not from that code.
"""
from __future__ import annotations

import itertools

import networkx as nx
import numpy as np

__all__ = [
    'balanced_tree', 'random_tree', 'jukes_cantor', 'hky85', 'mg94',
    'blinking_model', 'blinking_allowed_states', 'simulate_states',
    'one_hot', 'make_config', 'CONFIG_NAMES',
]

CONFIG_NAMES = ('c1', 'c2', 'c3', 'c4', 'c5')


# ---------------------------------------------------------------------------
# trees
# ---------------------------------------------------------------------------

def balanced_tree(nleaves, seed=0):
    """Perfectly balanced binary tree, heap numbering (root 0, children 2i+1
    and 2i+2), branch lengths 0.05 + 0.10*U(0,1) from RandomState(seed)."""
    if nleaves < 1 or nleaves & (nleaves - 1):
        raise ValueError('nleaves must be a power of two')
    nnodes = 2 * nleaves - 1
    rng = np.random.RandomState(seed)
    T = nx.Graph()
    T.add_node(0)
    for child in range(1, nnodes):
        parent = (child - 1) // 2
        T.add_edge(parent, child, weight=0.05 + 0.10 * rng.uniform())
    leaves = list(range(nleaves - 1, nnodes))
    return T, 0, leaves


def random_tree(nnodes, seed=0, max_children=3):
    """Random rooted tree with arbitrary (non-contiguous) integer node ids and
    multifurcations -- for edge-case tests (SURVEY.md section 7 hard parts)."""
    rng = np.random.RandomState(seed)
    ids = rng.permutation(np.arange(10, 10 + 3 * nnodes))[:nnodes].tolist()
    T = nx.Graph()
    T.add_node(ids[0])
    nchildren = {ids[0]: 0}
    for k in range(1, nnodes):
        while True:
            parent = ids[rng.randint(k)]
            if nchildren[parent] < max_children:
                break
        T.add_edge(parent, ids[k], weight=0.02 + 0.3 * rng.uniform())
        nchildren[parent] += 1
        nchildren[ids[k]] = 0
    leaves = [v for v in ids if nchildren[v] == 0]
    return T, ids[0], leaves


# ---------------------------------------------------------------------------
# rate matrices
# ---------------------------------------------------------------------------

def _finish_rate_matrix(R, distn, expected_rate=1.0):
    Q = np.array(R, dtype=float)
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(axis=1))
    if expected_rate is not None:
        rate = -float(np.dot(distn, np.diag(Q)))
        Q *= expected_rate / rate
    return Q


def jukes_cantor(n=4):
    """Jukes-Cantor: off-diagonal 1/(n-1) (reference
    _conditional_expectation.py:15-23), stationary distribution uniform."""
    Q = np.full((n, n), 1.0 / (n - 1))
    distn = np.full(n, 1.0 / n)
    return _finish_rate_matrix(Q, distn, expected_rate=None), distn


def hky85(kappa=2.0, pi=(0.1, 0.2, 0.3, 0.4)):
    """HKY85 over (A, C, G, T): rate i->j = pi_j, times kappa for transitions
    (A<->G, C<->T); scaled to expected rate 1."""
    pi = np.asarray(pi, dtype=float)
    pi = pi / pi.sum()
    transitions = {(0, 2), (2, 0), (1, 3), (3, 1)}
    R = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            if i != j:
                R[i, j] = pi[j] * (kappa if (i, j) in transitions else 1.0)
    return _finish_rate_matrix(R, pi), pi


_NT = 'ACGT'
# Standard genetic code, codons enumerated with nucleotides in TCAG order
# (the usual textbook table); '*' = stop.
_TCAG_AA = ('FFLLSSSSYY**CC*W' 'LLLLPPPPHHQQRRRR'
            'IIIMTTTTNNKKSSRR' 'VVVVAAAADDEEGGGG')


def genetic_code():
    """Sense codons of the universal code as (codon, amino acid) sorted by
    codon string: 61 states."""
    table = {}
    for idx, (a, b, c) in enumerate(itertools.product('TCAG', repeat=3)):
        table[a + b + c] = _TCAG_AA[idx]
    return [(cod, aa) for cod, aa in sorted(table.items()) if aa != '*']


def mg94(kappa=3.17632, omega=0.21925,
         nt_freqs=(0.25039, 0.30126, 0.25952, 0.18883)):
    """Muse-Gaut 94 codon model: single-nucleotide changes only; rate =
    pi[target nucleotide] * (kappa if transition) * (omega if the amino acid
    changes); stationary distribution proportional to the product of the three
    nucleotide frequencies; scaled to expected rate 1.  nt_freqs are
    (A, C, G, T); defaults are the PAML estimates at reference
    examples/p53/p53.py:22-27."""
    code = genetic_code()
    n = len(code)
    ntp = dict(zip(_NT, np.asarray(nt_freqs, dtype=float)))
    is_ts = {('A', 'G'), ('G', 'A'), ('C', 'T'), ('T', 'C')}
    R = np.zeros((n, n))
    for a, (ca, ra) in enumerate(code):
        for b, (cb, rb) in enumerate(code):
            diff = [(x, y) for x, y in zip(ca, cb) if x != y]
            if len(diff) != 1:
                continue
            x, y = diff[0]
            rate = ntp[y]
            if (x, y) in is_ts:
                rate *= kappa
            if ra != rb:
                rate *= omega
            R[a, b] = rate
    w = np.array([ntp[c[0]] * ntp[c[1]] * ntp[c[2]] for c, _ in code])
    distn = w / w.sum()
    return _finish_rate_matrix(R, distn), distn


def blinking_model(Q_primary, primary_distn, primary_to_part,
                   rate_on, rate_off):
    """Compound "blinking" (tolerance) process of reference
    examples/code2x3/run.py:341-434: state = (tolerance tuple in {0,1}^nparts,
    primary state), index = block * nprimary + primary, with blocks enumerated
    by itertools.product((0,1), repeat=nparts).  Primary moves c->d happen at
    Q_primary[c,d] only when both classes are tolerated; tolerance class p
    flips on at rate_on / off at rate_off provided the current primary state
    stays tolerated.  Returns (Q f64[n,n], distn f64[n]) with
    n = nprimary * 2**nparts; untolerated compound states have zero prior."""
    Q_primary = np.asarray(Q_primary, dtype=float)
    nprimary = Q_primary.shape[0]
    part = [primary_to_part[c] for c in range(nprimary)]
    nparts = len(set(part))
    tuples = list(itertools.product((0, 1), repeat=nparts))
    block_of = dict((t, i) for i, t in enumerate(tuples))
    n = nprimary * len(tuples)
    Q = np.zeros((n, n))
    for bi, tol in enumerate(tuples):
        a = bi * nprimary
        for c in range(nprimary):
            for d in range(nprimary):
                if c != d and tol[part[c]] and tol[part[d]]:
                    Q[a + c, a + d] = Q_primary[c, d]
        for p in range(nparts):
            adj = tuple(v if q != p else 1 - v for q, v in enumerate(tol))
            rate = rate_on if adj[p] else rate_off
            for c in range(nprimary):
                if tol[part[c]] and adj[part[c]]:
                    Q[a + c, block_of[adj] * nprimary + c] = rate
    Q -= np.diag(Q.sum(axis=1))
    tol_distn = np.array([rate_off, rate_on], dtype=float)
    tol_distn /= tol_distn.sum()
    distn = np.zeros(n)
    for bi, tol in enumerate(tuples):
        n_tol = sum(tol)
        n_untol = nparts - n_tol
        for c in range(nprimary):
            if tol[part[c]]:
                distn[bi * nprimary + c] = (primary_distn[c] *
                                            tol_distn[0] ** n_untol *
                                            tol_distn[1] ** (n_tol - 1))
    return Q, distn


def blinking_allowed_states(primary_state, nprimary, primary_to_part):
    """Allowed compound states at a leaf whose primary state is observed:
    the tolerance class of that state must be on, the others are free
    (reference examples/code2x3/run.py:436-461)."""
    part = [primary_to_part[c] for c in range(nprimary)]
    nparts = len(set(part))
    out = []
    for bi, tol in enumerate(itertools.product((0, 1), repeat=nparts)):
        if tol[part[primary_state]]:
            out.append(bi * nprimary + primary_state)
    return out


def switching_model(Q_default, primary_distn, benign_states, rho):
    """Compound "switching" process of reference examples/p53/liwen.py:599-621 over
    2 n states: states 0..n-1 are the reference process (only moves INTO benign
    states are allowed: edge sa -> sb is kept when sb is benign, :612-615), states
    n..2n-1 the default process (all of Q_default, :606-608), and every reference
    state s switches to its default twin n + s at rate rho (:619-620); never back.
    The diagonal is minus the row sum (_density.rate_matrix_to_numpy_array, :46-48).
    The prior puts the primary distribution, renormalised, on the benign reference
    states (:623-627).  Returns (Q f64[2n,2n], distn f64[2n])."""
    Q_default = np.asarray(Q_default, dtype=float)
    n = Q_default.shape[0]
    benign = np.zeros(n, dtype=bool)
    benign[list(benign_states)] = True
    off = Q_default - np.diag(np.diag(Q_default))
    Q = np.zeros((2 * n, 2 * n))
    Q[n:, n:] = off
    Q[:n, :n] = off * benign[None, :]
    Q[np.arange(n), n + np.arange(n)] = rho
    Q -= np.diag(Q.sum(axis=1))
    # the reference normalises a dict with Python's sum() in ascending state order
    # (_util.py:104-109): a plain left-to-right sum, spelled out so that it does not
    # depend on the interpreter's sum() (compensated from Python 3.12 on)
    p = np.asarray(primary_distn, dtype=float)
    total = 0.0
    for s in range(n):
        if benign[s]:
            total += float(p[s])
    distn = np.zeros(2 * n)
    distn[:n] = np.where(benign, p, 0.0) / total
    return Q, distn


def switching_allowed_states(codon_state, nstates):
    """Allowed compound states of a leaf whose codon is observed: the codon in either
    process (examples/p53/liwen.py:682)."""
    return [int(codon_state), int(nstates) + int(codon_state)]


# ---------------------------------------------------------------------------
# data
# ---------------------------------------------------------------------------

def _expm(Q, t):
    """exp(Q t) for SIMULATING data on the host (numpy only; the product's expm is
    the device kernel): scaling and squaring around a degree-18 Taylor polynomial,
    ||A / 2^s||_1 <= 0.5 (truncation error < 1e-21 there)."""
    A = np.asarray(Q, dtype=float) * float(t)
    nrm = np.abs(A).sum(axis=0).max() if A.size else 0.0
    s = max(0, int(np.ceil(np.log2(nrm / 0.5)))) if nrm > 0.5 else 0
    A = A / (2.0 ** s)
    X = np.eye(A.shape[0])
    term = np.eye(A.shape[0])
    for k in range(1, 19):
        term = term.dot(A) / k
        X = X + term
    for _ in range(s):
        X = X.dot(X)
    return X


def simulate_states(T, root, node_to_P, root_distn, nsites, seed):
    """Sample states at every node for ``nsites`` independent sites by walking
    down the tree.  node_to_P maps child node -> transition matrix of its
    parent edge.  Returns {node: int array[nsites]}."""
    rng = np.random.RandomState(seed)
    cdf0 = np.cumsum(root_distn)
    cdf0[-1] = 1.0
    out = {root: np.searchsorted(cdf0, rng.uniform(size=nsites),
                                 side='right').astype(np.int64)}
    np.minimum(out[root], len(cdf0) - 1, out=out[root])
    for na, nb in nx.bfs_edges(T, root):
        P = np.asarray(node_to_P[nb], dtype=float)
        cdf = np.cumsum(np.maximum(P, 0.0), axis=1)
        cdf /= cdf[:, -1:]
        u = rng.uniform(size=nsites)
        # child state = #{j: u >= cdf[parent, j]}: one searchsorted per parent state
        # (a [nsites, n] gather of cdf rows would be 0.5 GB per edge at 10^6 codon sites)
        parent = out[na]
        s = np.empty(nsites, dtype=np.int64)
        for a in np.unique(parent):
            sel = np.flatnonzero(parent == a)
            s[sel] = np.searchsorted(cdf[a], u[sel], side='right')
        out[nb] = np.minimum(s, P.shape[0] - 1)
    return out


def one_hot(states, nstates):
    """int[...] -> f64[..., nstates] one-hot leaf likelihood vectors."""
    states = np.asarray(states)
    out = np.zeros(states.shape + (nstates,), dtype=np.float64)
    np.put_along_axis(out, states[..., None], 1.0, axis=-1)
    return out


C4_NSITES = 1000000
C4_CHUNK = 15625          # 64 chunks; chunk c is simulated from RandomState([3, c])


def c4_leaf_states(lo, hi, T, root, leaves, Q, distn):
    """Sites [lo, hi) of THE config-4 batch (1 000 000 codon sites, seed 3).  The
    batch is defined chunk by chunk (C4_CHUNK sites from RandomState([3, chunk])),
    so a rank simulates only the chunks its shard_range overlaps and every rank
    count sees the same million sites."""
    if not (0 <= lo <= hi <= C4_NSITES):
        raise ValueError('site range outside the config-4 batch')
    node_to_P = dict((nb, _expm(Q, T[na][nb]['weight']))
                     for na, nb in nx.bfs_edges(T, root))
    parts = []
    for c in range(lo // C4_CHUNK, -(-hi // C4_CHUNK)):
        states = simulate_states(T, root, node_to_P, distn, C4_CHUNK,
                                 np.array([3, c], dtype=np.uint32))
        chunk = np.stack([states[v] for v in leaves], axis=1)
        a = max(lo, c * C4_CHUNK) - c * C4_CHUNK
        b = min(hi, (c + 1) * C4_CHUNK) - c * C4_CHUNK
        parts.append(chunk[a:b])
    if not parts:
        return np.zeros((0, len(leaves)), dtype=np.int64)
    return np.concatenate(parts, axis=0)


def make_config(name, nsites=None, site_range=None):
    """Build one of the BASELINE.json configurations.

    ``site_range=(lo, hi)`` (config 4 only): that slice of the one million-site
    batch (what rank r of a sharded run uploads: dist.shard_range); ``nsites`` for
    c4 is shorthand for ``site_range=(0, nsites)``.

    Returns a dict with keys: name, T (nx.Graph with 'weight' and, for C5, 'Q'
    per edge), root, leaves, nstates, Q_default (or None), root_distn,
    leaf_states int64[nsites, nleaves] (simulated), obs_kind ('state' for
    C1-C4, 'mask' for C5), and for C5 leaf_allowed (list per leaf state)."""
    name = name.lower()
    if name == 'c1':
        T, root, leaves = balanced_tree(8, seed=0)
        Q, distn = hky85()
        nsites = 1 if nsites is None else nsites
        seed = 11
    elif name == 'c2':
        T, root, leaves = balanced_tree(64, seed=0)
        Q, distn = hky85()
        nsites = 100000 if nsites is None else nsites
        seed = 1
    elif name == 'c4':
        T, root, leaves = balanced_tree(64, seed=0)
        Q, distn = mg94()
        if site_range is None:
            site_range = (0, C4_NSITES if nsites is None else nsites)
        leaf_states = c4_leaf_states(site_range[0], site_range[1], T, root, leaves,
                                     Q, distn)
        return dict(name=name, T=T, root=root, leaves=leaves, nstates=len(distn),
                    Q_default=Q, root_distn=distn, leaf_states=leaf_states,
                    obs_kind='state', site_range=tuple(site_range))
    elif name == 'c3':
        T, root, leaves = balanced_tree(64, seed=0)
        Q, distn = mg94()
        nsites = 10000 if nsites is None else nsites
        seed = 2
    elif name == 'c5':
        return _make_c5(50000 if nsites is None else nsites)
    elif name == 'c6':
        return _make_c6(10000 if nsites is None else nsites)
    else:
        raise ValueError('unknown config %r' % (name,))
    node_to_P = dict((nb, _expm(Q, T[na][nb]['weight']))
                     for na, nb in nx.bfs_edges(T, root))
    states = simulate_states(T, root, node_to_P, distn, nsites, seed)
    leaf_states = np.stack([states[v] for v in leaves], axis=1)
    return dict(name=name, T=T, root=root, leaves=leaves, nstates=len(distn),
                Q_default=Q, root_distn=distn, leaf_states=leaf_states,
                obs_kind='state')


def _make_c5(nsites):
    nprimary, nparts = 5, 2
    primary_to_part = {0: 0, 1: 0, 2: 0, 3: 1, 4: 1}
    T, root, leaves = balanced_tree(32, seed=0)
    rng = np.random.RandomState(4)
    # primary process: random reversible 5-state matrix, expected rate 1
    w = 0.5 + rng.exponential(size=nprimary)
    primary_distn = w / w.sum()
    S = 0.5 + rng.exponential(size=(nprimary, nprimary))
    S = (S + S.T) / 2
    Q_primary = _finish_rate_matrix(S * primary_distn[None, :], primary_distn)
    root_rates = None
    for na, nb in nx.bfs_edges(T, root):
        rate_on = 0.5 + rng.exponential()
        rate_off = 0.5 + rng.exponential()
        Q, distn = blinking_model(Q_primary, primary_distn, primary_to_part,
                                  rate_on, rate_off)
        T[na][nb]['Q'] = Q
        if root_rates is None:
            root_rates = (rate_on, rate_off)
            root_distn = distn
    node_to_P = dict((nb, _expm(T[na][nb]['Q'], T[na][nb]['weight']))
                     for na, nb in nx.bfs_edges(T, root))
    states = simulate_states(T, root, node_to_P, root_distn, nsites, 4)
    compound = np.stack([states[v] for v in leaves], axis=1)
    leaf_primary = compound % nprimary
    n = nprimary * 2 ** nparts
    allowed = [blinking_allowed_states(c, nprimary, primary_to_part)
               for c in range(nprimary)]
    return dict(name='c5', T=T, root=root, leaves=leaves, nstates=n,
                Q_default=None, root_distn=root_distn,
                leaf_states=leaf_primary, obs_kind='mask',
                leaf_allowed=allowed, nprimary=nprimary,
                primary_to_part=primary_to_part)


def _make_c6(nsites):
    """Not a BASELINE.json configuration: the state space of examples/p53/liwen.py:599-621
    (MG94 x {reference, default} = 122 compound states, switching_model above) at the shape
    of config 3 -- 64-leaf balanced tree, leaf sets {c, 61 + c} (liwen.py:682) for codons
    simulated from the default process, a seeded half of the amino acids benign."""
    T, root, leaves = balanced_tree(64, seed=0)
    Q, distn = mg94()
    code = genetic_code()
    rng = np.random.RandomState(6)
    residues = sorted(set(aa for _, aa in code))
    benign_res = set(r for r in residues if rng.uniform() < 0.5)
    benign = [s for s, (_, aa) in enumerate(code) if aa in benign_res]
    Qc, dc = switching_model(Q, distn, benign, 0.61610)
    node_to_P = dict((nb, _expm(Q, T[na][nb]['weight']))
                     for na, nb in nx.bfs_edges(T, root))
    states = simulate_states(T, root, node_to_P, distn, nsites, 6)
    leaf_states = np.stack([states[v] for v in leaves], axis=1)
    n = len(distn)
    return dict(name='c6', T=T, root=root, leaves=leaves, nstates=2 * n, Q_default=Qc,
                root_distn=dc, leaf_states=leaf_states, obs_kind='mask',
                leaf_allowed=[switching_allowed_states(c, n) for c in range(n)])


def leaf_likelihoods(cfg, dtype=np.float64):
    """Dense per-site leaf likelihood vectors f64[nsites, nleaves, n] for a
    config: one-hot for observed states (C1-C4), 0/1 allowed-set masks (C5)."""
    n = cfg['nstates']
    if cfg['obs_kind'] == 'state':
        return one_hot(cfg['leaf_states'], n).astype(dtype, copy=False)
    table = np.zeros((len(cfg['leaf_allowed']), n), dtype=dtype)
    for c, states in enumerate(cfg['leaf_allowed']):
        table[c, states] = 1.0
    return table[cfg['leaf_states']]


def site_node_to_allowed_states(cfg, site):
    """The reference-style observation dict for one site: every node mapped to
    a set of allowed states (``_mcy_dense.py:43-54`` needs every node)."""
    n = cfg['nstates']
    d = dict((v, set(range(n))) for v in cfg['T'])
    for k, leaf in enumerate(cfg['leaves']):
        s = int(cfg['leaf_states'][site, k])
        if cfg['obs_kind'] == 'state':
            d[leaf] = {s}
        else:
            d[leaf] = set(cfg['leaf_allowed'][s])
    return d
