"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

A CPU (numpy/scipy/networkx) restatement of the tree-CTMC likelihood hot path of
argriffing/raoteh.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; ``raoteh_amd``
never does (the product fails loudly when its HIP library is missing).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against ``tests/golden/*.json``, which ``tools/gen_golden.py`` produced by
importing the reference's own pure-Python modules (``_mcx``, ``_mc0``,
``_mcy``/``_mcz`` un-accelerated twins) from ``/root/reference`` in the build
container, and against the literal known answers of the reference's tests
(``tests/test_mc.py:150`` ``4*log(0.5)``, ``tests/test_mjp.py:91-164``
re-rooting invariance, ``tests/test_mjp.py:52-89`` sum-to-one,
``_conditional_expectation.py:25-33`` Jukes-Cantor p_ij(t)).

Each function cites the reference file:line (relative to /root/reference) whose
behaviour it restates.  The third-party arithmetic on the reference path is

* ``scipy.linalg.expm`` (called at ``raoteh/sampler/_mjp_dense.py:24-25``) --
  the oracle calls the same scipy function (scipy is unpinned by the
  reference; the fixtures record the version used), and
* ``pyfelscore`` (absent, unpinned: ``README.md:8-9``) -- restated from its
  pure-Python twins in the reference, cited per function below.
"""
from __future__ import annotations

import warnings

import networkx as nx
import numpy as np
import scipy.linalg


class ZeroProbError(Exception):
    """raoteh/sampler/_util.py:14-15"""


class StructuralZeroProb(ZeroProbError):
    """raoteh/sampler/_util.py:17-18"""


class NumericalZeroProb(ZeroProbError):
    """raoteh/sampler/_util.py:20-21"""


# ---------------------------------------------------------------------------
# expm
# ---------------------------------------------------------------------------

def custom_expm(Q, weight):
    """raoteh/sampler/_mjp_dense.py:24-25 -- ``scipy.linalg.expm(Q * weight)``."""
    return scipy.linalg.expm(np.asarray(Q, dtype=float) * weight)


# Pade coefficients and thresholds of N. J. Higham, "The scaling and squaring
# method for the matrix exponential revisited", SIAM J. Matrix Anal. Appl. 26(4)
# 2005, Table 2.3 and eq. (2.5)/(10.33 in Functions of Matrices).  This is the
# published algorithm the HIP kernel implements; scipy implements the later
# Al-Mohy & Higham 2009 refinement, so the two agree to rounding, not bitwise.
PADE_THETA = {
    3: 1.495585217958292e-2,
    5: 2.539398330063230e-1,
    7: 9.504178996162932e-1,
    9: 2.097847961257068e0,
    13: 5.371920351148152e0,
}
PADE_B = {
    3: (120., 60., 12., 1.),
    5: (30240., 15120., 3360., 420., 30., 1.),
    7: (17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.),
    9: (17643225600., 8821612800., 2075673600., 302702400., 30270240.,
        2162160., 110880., 3960., 90., 1.),
    13: (64764752532480000., 32382376266240000., 7771770303897600.,
         1187353796428800., 129060195264000., 10559470521600.,
         670442572800., 33522128640., 1323241920., 40840800., 960960.,
         16380., 182., 1.),
}


def pade_order_and_squarings(norm1):
    """Degree m and number of squarings s for a matrix of 1-norm ``norm1``
    (Higham 2005, Algorithm 2.3)."""
    for m in (3, 5, 7, 9):
        if norm1 <= PADE_THETA[m]:
            return m, 0
    s = 0
    if norm1 > PADE_THETA[13]:
        s = max(0, int(np.ceil(np.log2(norm1 / PADE_THETA[13]))))
    return 13, s


def expm_pade(Q, t):
    """Higham-2005 scaling-and-squaring expm(Q*t), the algorithm of the HIP
    kernel, restated in numpy so kernel-vs-scipy differences can be separated
    into "algorithm" and "implementation"."""
    A = np.asarray(Q, dtype=float) * float(t)
    n = A.shape[0]
    I = np.eye(n)
    norm1 = np.abs(A).sum(axis=0).max() if n else 0.0
    m, s = pade_order_and_squarings(norm1)
    if s:
        A = A * (2.0 ** -s)
    b = PADE_B[m]
    A2 = A @ A
    if m == 13:
        A4 = A2 @ A2
        A6 = A4 @ A2
        W = A6 @ (b[13] * A6 + b[11] * A4 + b[9] * A2)
        U = A @ (W + b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I)
        Z = A6 @ (b[12] * A6 + b[10] * A4 + b[8] * A2)
        V = Z + b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I
    else:
        powers = [I, A2]
        for _ in range(2, m // 2 + 1):
            powers.append(powers[-1] @ A2)
        W = sum(b[2 * k + 1] * powers[k] for k in range(m // 2 + 1))
        U = A @ W
        V = sum(b[2 * k] * powers[k] for k in range(m // 2 + 1))
    X = np.linalg.solve(V - U, V + U)
    for _ in range(s):
        X = X @ X
    return X


# The algorithm of the device's DEFAULT expm kernel for n > 4 (round 2): scaling and
# squaring around a Taylor polynomial evaluated by Paterson-Stockmeyer (products only,
# no solve).  theta_m = largest 1-norm for which the backward error of the degree-m
# Taylor polynomial stays below 2^-53 (Higham 2005 sec. 2 applied to the Taylor series;
# the values of Al-Mohy & Higham 2011, table 3.1).  Like expm_pade this restates OUR
# algorithm, so kernel-vs-scipy differences separate into algorithm and implementation;
# the reference's own call is scipy.linalg.expm (custom_expm above, _mjp_dense.py:24-25).
TAYLOR_THETA = {3: 1.3863479e-5, 6: 9.0656564e-3, 9: 8.9577602e-2, 12: 2.9961589e-1,
                15: 6.4108352e-1}


def taylor_order_and_squarings(norm1):
    """Degree m and number of squarings s of the Taylor kernel for 1-norm ``norm1``."""
    for m in (3, 6, 9, 12):
        if norm1 <= TAYLOR_THETA[m]:
            return m, 0
    s = 0
    if norm1 > TAYLOR_THETA[15]:
        s = max(0, int(np.ceil(np.log2(norm1 / TAYLOR_THETA[15]))))
    return 15, s


def expm_taylor(Q, t):
    """exp(Q t) by the algorithm of csrc/expm.hip expm_taylor_kernel: A^2 = A A,
    A^3 = A A^2, then Horner in A^3 over the blocks
    B_j = I / (3j)! + A / (3j+1)! + A^2 / (3j+2)!  (the top block also carries
    A^3 / m!), degree m = 3 q in {3, 6, 9, 12, 15}, then s squarings."""
    import math
    A = np.asarray(Q, dtype=float) * float(t)
    n = A.shape[0]
    I = np.eye(n)
    norm1 = np.abs(A).sum(axis=0).max() if n else 0.0
    m, s = taylor_order_and_squarings(norm1)
    if s:
        A = A * (2.0 ** -s)
    c = [1.0 / math.factorial(i) for i in range(m + 1)]
    A2 = A @ A
    A3 = A @ A2
    q = m // 3

    def block(j):
        return c[3 * j] * I + c[3 * j + 1] * A + c[3 * j + 2] * A2
    X = block(q - 1) + c[m] * A3
    for j in range(q - 2, -1, -1):
        X = A3 @ X + block(j)
    for _ in range(s):
        X = X @ X
    return X


def device_expm_restated(Q, t):
    """What the device computes by default: the Taylor scheme at every size (the Pade
    kernels remain behind RAOTEH_EXPM=pade)."""
    return expm_taylor(Q, t)


def device_expm_order_and_squarings(n, norm1):
    return taylor_order_and_squarings(norm1)


# ---------------------------------------------------------------------------
# the optional spectral path of one reversible rate matrix (examples/p53/qtop.py)
# ---------------------------------------------------------------------------

def spectral_decompose_v2(S, D):
    """qtop.py:128-150 (decompose_spectral + decompose_spectral_v2): Q = S diag(D) ->
    (A, lam, B) with A = diag(pseudo_reciprocal(sqrt D)) U, B = U^T diag(sqrt D),
    eigh(diag(sqrt D) S diag(sqrt D)) = (lam, U)."""
    S = np.asarray(S, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64)
    r = np.sqrt(D)
    lam, U = scipy.linalg.eigh(r[:, None] * S * r[None, :])
    with np.errstate(divide='ignore'):
        rinv = np.where(r == 0, r, np.reciprocal(r))
    return rinv[:, None] * U, lam, U.T * r[None, :]


def spectral_getp_v2(D, A, lam, B, t):
    """qtop.py:76-88 (getp_spectral_v2 over reconstruct_spectral_v2, :283-288)."""
    P = np.dot(np.asarray(A) * np.exp(t * np.asarray(lam))[None, :], np.asarray(B))
    off = np.asarray(D) == 0
    P[off, off] = 1
    return P


# ---------------------------------------------------------------------------
# tree marshalling (reference: _mcy_dense.py:246-255, _density.py:104-180)
# ---------------------------------------------------------------------------

def tree_to_arrays(T, root):
    """Preorder node list and children-CSR in preorder index space.

    raoteh/sampler/_mcy_dense.py:246-259: ``T_bfs`` = digraph of
    ``nx.bfs_edges(T, root)``; ``preorder_nodes = list(nx.dfs_preorder_nodes(T,
    root))``; ``_density.digraph_to_bool_csr`` (``_density.py:104-140``).
    """
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    T_bfs = nx.DiGraph()
    T_bfs.add_node(root)
    for na, nb in nx.bfs_edges(T, root):
        T_bfs.add_edge(na, nb)
    preorder_nodes = list(nx.dfs_preorder_nodes(T, root))
    node_to_index = dict((n, i) for i, n in enumerate(preorder_nodes))
    indices = []
    indptr = [0]
    for na in preorder_nodes:
        for nb in T_bfs[na]:
            indices.append(node_to_index[nb])
        indptr.append(len(indices))
    return (preorder_nodes,
            np.array(indices, dtype=np.int64),
            np.array(indptr, dtype=np.int64))


def get_expm_augmented_transitions(T, root, nstates, Q_default=None):
    """raoteh/sampler/_mjp_dense.py:328-359 followed by
    ``_density.get_esd_transitions`` (``_density.py:143-180``): per BFS edge
    ``P = expm(Q*weight)`` with ``Q = edge.get('Q', Q_default)``, stacked into
    f64[N,n,n] at the CHILD's preorder index; the root slot stays zero."""
    preorder_nodes, indices, indptr = tree_to_arrays(T, root)
    node_to_index = dict((n, i) for i, n in enumerate(preorder_nodes))
    esd = np.zeros((len(preorder_nodes), nstates, nstates), dtype=float)
    for na, nb in nx.bfs_edges(T, root):
        edge = T[na][nb]
        Q = edge.get('Q', Q_default)
        check_square_dense(Q)
        esd[node_to_index[nb]] = custom_expm(Q, edge['weight'])
    return preorder_nodes, indices, indptr, esd


def get_esd_transitions(T, root, nstates, P_default=None):
    """raoteh/sampler/_density.py:143-180 for a tree whose edges carry ``P``."""
    preorder_nodes, indices, indptr = tree_to_arrays(T, root)
    node_to_index = dict((n, i) for i, n in enumerate(preorder_nodes))
    esd = np.zeros((len(preorder_nodes), nstates, nstates), dtype=float)
    for na, nb in nx.bfs_edges(T, root):
        P = T[na][nb].get('P', P_default)
        check_square_dense(P)
        esd[node_to_index[nb]] = P
    return preorder_nodes, indices, indptr, esd


def check_square_dense(M):
    """raoteh/sampler/_density.py:79-101."""
    if M is None:
        raise ValueError('the matrix is None')
    try:
        shape = M.shape
    except AttributeError:
        raise ValueError('expected an ndarray')
    if len(shape) != 2:
        raise ValueError('expected len(M.shape) == 2')
    if shape[0] != shape[1]:
        raise ValueError('expected the array to be square')


def define_state_mask(node_to_allowed_states, preorder_nodes, nstates):
    """raoteh/sampler/_mcy_dense.py:43-54 (KeyError for a missing node is the
    reference's behaviour when the dict is given)."""
    nnodes = len(preorder_nodes)
    mask = np.ones((nnodes, nstates), dtype=np.int64)
    if node_to_allowed_states is not None:
        all_states = set(range(nstates))
        for i, na in enumerate(preorder_nodes):
            for sa in all_states - set(node_to_allowed_states[na]):
                mask[i, sa] = 0
    return mask


def define_state_mask_x(node_to_state, preorder_nodes, nstates):
    """raoteh/sampler/_mcx_dense.py:48-86 (type-x observations)."""
    nnodes = len(preorder_nodes)
    mask = np.ones((nnodes, nstates), dtype=np.int64)
    if node_to_state is not None:
        for i, na in enumerate(preorder_nodes):
            if na in node_to_state:
                mask[i] = 0
                mask[i, node_to_state[na]] = 1
    return mask


# ---------------------------------------------------------------------------
# the three pyfelscore passes (restated from their pure-Python twins)
# ---------------------------------------------------------------------------

def mcy_esd_get_node_to_pset(indices, indptr, esd, state_mask):
    """Backward (leaves->root) boolean pass, in place on ``state_mask``.

    pyfelscore.mcy_esd_get_node_to_pset as called at
    ``_mcy_dense.py:270``; twins ``_mcx.py:79-138`` / ``_mcy.py:409-467``:
    state ``s`` of a node stays allowed only if, for EVERY child, some allowed
    child state ``s'`` has ``P_child[s, s'] > 0``.  (The twins also intersect
    with the source states of the parent edge's P; for a dense P that is every
    state, which is what the dense call site passes.)"""
    nnodes = len(indptr) - 1
    for v in range(nnodes - 1, -1, -1):
        for c in indices[indptr[v]:indptr[v + 1]]:
            reach = (esd[c] > 0) @ (state_mask[c] != 0)
            state_mask[v] *= (reach > 0)
    return state_mask


def esd_get_node_to_set(indices, indptr, esd, state_mask):
    """Forward (root->leaves) boolean pass, in place.

    pyfelscore.esd_get_node_to_set as called at ``_mcy_dense.py:277``; twin
    ``_mc0.py:121-138``: a child state stays allowed only if it is reachable
    (``P[sa, sb] > 0``) from some allowed parent state."""
    nnodes = len(indptr) - 1
    for v in range(nnodes):
        for c in indices[indptr[v]:indptr[v + 1]]:
            reach = (state_mask[v] != 0) @ (esd[c] > 0)
            state_mask[c] *= (reach > 0)
    return state_mask


def mcy_esd_get_node_to_pmap(indices, indptr, esd, state_mask, out=None,
                             obs_lik=None):
    """Felsenstein upward pass.

    pyfelscore.mcy_esd_get_node_to_pmap as called at ``_mcy_dense.py:286``;
    twins ``_mcx.py:188-210``, ``_mcy.py:657-679``; type-z variant
    ``_mcz.py:140-163`` multiplies in ``obs_lik[node, state]``:

        L[v,s] = mask[v,s] * obs_lik[v,s] * prod_c sum_s' P_c[s,s'] L[c,s']

    No rescaling anywhere (the reference has none)."""
    nnodes = len(indptr) - 1
    nstates = esd.shape[1]
    if out is None:
        out = np.empty((nnodes, nstates), dtype=float)
    for v in range(nnodes - 1, -1, -1):
        acc = np.ones(nstates, dtype=float)
        for c in indices[indptr[v]:indptr[v + 1]]:
            acc = acc * (esd[c] @ out[c])
        if obs_lik is not None:
            acc = acc * obs_lik[v]
        out[v] = np.where(state_mask[v] != 0, acc, 0.0)
    return out


def mc0_get_likelihood(root_pmap, root_distn=None):
    """raoteh/sampler/_mc0_dense.py:147-212 (root reduction + exceptions)."""
    root_pmap = np.asarray(root_pmap, dtype=float)
    if root_distn is not None:
        root_distn = np.asarray(root_distn, dtype=float)
        if root_pmap.shape != root_distn.shape:
            raise ValueError('root shape mismatch: %s %s' % (
                root_pmap.shape, root_distn.shape))
        prior_feasible = set(s for s, p in enumerate(root_distn) if p)
        if not prior_feasible:
            raise StructuralZeroProb(
                'no root state has nonzero prior likelihood')
    if root_pmap.min() < 0:
        warnings.warn('root_pmap should have non-negative entries '
                      'but found minimum entry %s' % root_pmap.min())
        root_pmap = np.maximum(root_pmap, 0)
    if not root_pmap.sum():
        raise StructuralZeroProb(
            'all root states give a subtree likelihood of zero')
    feasible = set(s for s, p in enumerate(root_pmap) if p)
    if root_distn is not None:
        feasible &= prior_feasible
    if not feasible:
        raise StructuralZeroProb(
            'all root states have either zero prior likelihood '
            'or give a subtree likelihood of zero')
    if root_distn is not None:
        return float(root_distn.dot(root_pmap))
    return float(root_pmap.sum())


def mc0_esd_get_node_to_distn(indices, indptr, esd, root_distn, pmap):
    """Downward pass: posterior marginal state distribution per node.

    pyfelscore.mc0_esd_get_node_to_distn as called at ``_mc0_dense.py:381`` /
    ``_mcy_dense.py:195``; pure-Python twin ``_mc0_dense.py:446-486`` (sparse
    twin ``_mc0.py:382-462``).  Raises NumericalZeroProb where the reference's
    ``get_normalized_ndarray_distn`` would (``_util.py:164-165``)."""
    nnodes = len(indptr) - 1
    n = esd.shape[1]
    out = np.zeros((nnodes, n), dtype=float)
    w = pmap[0] * (1.0 if root_distn is None else np.asarray(root_distn, dtype=float))
    if not w.sum():
        raise NumericalZeroProb('the denominator is zero')
    out[0] = w / w.sum()
    for v in range(nnodes):
        for c in indices[indptr[v]:indptr[v + 1]]:
            d = np.zeros(n)
            for sa in range(n):
                pa = out[v, sa]
                if pa:
                    sb_w = esd[c][sa] * pmap[c]
                    tot = sb_w.sum()
                    if not tot:
                        raise NumericalZeroProb('the denominator is zero')
                    d += pa * (sb_w / tot)
            out[c] = d
    return out


def mc0_esd_get_joint_endpoint_distn(indices, indptr, esd, pmap, distn):
    """Joint (parent state, child state) posterior per edge, keyed by the child
    index.  pyfelscore.mc0_esd_get_joint_endpoint_distn (``_mcy_dense.py:205``);
    twin ``_mc0_dense.py:246-267`` (sparse twin ``_mc0.py:255-308``)."""
    nnodes = len(indptr) - 1
    n = esd.shape[1]
    J = np.zeros((nnodes, n, n), dtype=float)
    for v in range(nnodes):
        for c in indices[indptr[v]:indptr[v + 1]]:
            for sa in range(n):
                pa = distn[v, sa]
                if pa:
                    sb_w = esd[c][sa] * pmap[c]
                    tot = sb_w.sum()
                    if not tot:
                        raise NumericalZeroProb('the denominator is zero')
                    J[c, sa] = pa * (sb_w / tot)
    return J


# ---------------------------------------------------------------------------
# orchestration (reference: _mcy_dense.py:233-299,302-354,433-493;
#                _mjp_dense.py:362-407)
# ---------------------------------------------------------------------------

def esd_get_node_to_pmap(indices, indptr, esd, state_mask, obs_lik=None):
    """raoteh/sampler/_mcy_dense.py:261-291: the three passes in order.
    Returns (final state_mask, pmap f64[N,n])."""
    state_mask = np.array(state_mask, dtype=np.int64, copy=True)
    mcy_esd_get_node_to_pset(indices, indptr, esd, state_mask)
    esd_get_node_to_set(indices, indptr, esd, state_mask)
    pmap = mcy_esd_get_node_to_pmap(indices, indptr, esd, state_mask,
                                    obs_lik=obs_lik)
    return state_mask, pmap


def mcy_dense_get_node_to_pmap(T, root, nstates, node_to_allowed_states=None,
                               P_default=None):
    """raoteh/sampler/_mcy_dense.py:302-354 (dict of node -> f64[n])."""
    if len(T) == 1 and P_default is not None:
        if root not in T:
            raise ValueError('unrecognized root')
        allowed = set(range(nstates))
        if node_to_allowed_states is not None:
            allowed &= set(node_to_allowed_states[root])
        return {root: np.array(
            [1.0 if s in allowed else 0.0 for s in range(nstates)])}
    preorder_nodes, indices, indptr, esd = get_esd_transitions(
        T, root, nstates, P_default=P_default)
    mask = define_state_mask(node_to_allowed_states, preorder_nodes, nstates)
    _, pmap = esd_get_node_to_pmap(indices, indptr, esd, mask)
    return dict((na, pmap[i]) for i, na in enumerate(preorder_nodes))


def mcy_dense_get_likelihood(T, root, nstates, node_to_allowed_states=None,
                             root_distn=None, P_default=None):
    """raoteh/sampler/_mcy_dense.py:433-493."""
    if len(T) == 1:
        if root not in T:
            raise ValueError('unrecognized root')
        allowed_states = node_to_allowed_states[root]
        if not allowed_states:
            raise StructuralZeroProb('the tree has only a single node, '
                                     'and no state is allowed for the root')
        if root_distn is None:
            return 1
        pos = set(s for s in allowed_states if root_distn[s])
        if not pos:
            raise StructuralZeroProb(
                'the tree has only a single node, and every state with '
                'positive prior probability at the root is disallowed by a '
                'node state constraint')
        return sum(root_distn[s] for s in pos)
    node_to_pmap = mcy_dense_get_node_to_pmap(
        T, root, nstates, node_to_allowed_states=node_to_allowed_states,
        P_default=P_default)
    return mc0_get_likelihood(node_to_pmap[root], root_distn=root_distn)


def mjp_dense_get_likelihood(T, node_to_allowed_states, root, nstates,
                             root_distn=None, Q_default=None):
    """raoteh/sampler/_mjp_dense.py:362-407 -- the north-star entry point."""
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    T_aug = nx.Graph()
    if len(T) == 1:
        T_aug.add_node(root)
    for na, nb in nx.bfs_edges(T, root):
        edge = T[na][nb]
        Q = edge.get('Q', Q_default)
        check_square_dense(Q)
        T_aug.add_edge(na, nb, weight=edge['weight'],
                       P=custom_expm(Q, edge['weight']))
    return mcy_dense_get_likelihood(
        T_aug, root, nstates, node_to_allowed_states=node_to_allowed_states,
        root_distn=root_distn, P_default=None)


def mjp_dense_get_expected_history_statistics(T, node_to_allowed_states, root, nstates,
                                              root_distn=None, Q_default=None):
    """Expected dwell time per state, posterior root distribution and expected
    transition counts of one site: raoteh/sampler/_mjp_dense.py:410-539 (sparse twin
    _mjp.py:431-595).  Per edge, with J the joint endpoint posterior and P the
    transition matrix, every state c contributes sum_{a,b: J[a,b] != 0}
    J[a,b] * L(tQ, t E_cc)[a,b] / P[a,b] to its dwell time and every pair (c, d)
    with Q[c,d] != 0 (the dense reference includes c == d) the same sum with
    E_cd, times Q[c,d], to its transition count; L is scipy's expm_frechet, one
    call per (c, d) as in the reference.  Returns (dwell f64[n], root posterior
    f64[n], transitions f64[n,n]) -- the reference's dict / nx.DiGraph hold the
    same numbers."""
    if root not in T:
        raise ValueError('the specified root is not in the tree')
    n = nstates
    preorder_nodes, indices, indptr, esd = get_expm_augmented_transitions(
        T, root, n, Q_default=Q_default)
    mask = define_state_mask(node_to_allowed_states, preorder_nodes, n)
    _, pmap = esd_get_node_to_pmap(indices, indptr, esd, mask)
    distn = mc0_esd_get_node_to_distn(indices, indptr, esd, root_distn, pmap)
    J = mc0_esd_get_joint_endpoint_distn(indices, indptr, esd, pmap, distn)
    index = dict((v, i) for i, v in enumerate(preorder_nodes))
    dwell = np.zeros(n)
    trans = np.zeros((n, n))
    for na, nb in nx.bfs_edges(T, root):
        edge = T[na][nb]
        Q = np.asarray(edge.get('Q', Q_default), dtype=float)
        check_square_dense(Q)
        t = edge['weight']
        Pe, Je = esd[index[nb]], J[index[nb]]
        live = Je != 0
        ratio = np.zeros((n, n))
        ratio[live] = Je[live] / Pe[live]
        for c in range(n):
            for d in range(n):
                if c != d and not Q[c, d]:
                    continue
                C = np.zeros((n, n))
                C[c, d] = 1.0
                interact = scipy.linalg.expm_frechet(t * Q, t * C, compute_expm=False)
                total = float(np.sum(ratio[live] * interact[live]))
                if c == d:
                    dwell[c] += total
                if Q[c, d]:
                    trans[c, d] += Q[c, d] * total
    return dwell, distn[0], trans


def mjp_dense_expected_history_statistics_entries(T, node_to_allowed_states, root, nstates,
                                                  pairs, root_distn=None, Q_default=None):
    """The same numbers as mjp_dense_get_expected_history_statistics, for the listed
    (c, d) entries only: {(c, c): expected dwell time in c, (c, d): expected number of
    c -> d transitions}.  One expm_frechet call per listed direction and edge, exactly
    the reference's arithmetic (_mjp_dense.py:483-533) -- a 61-state codon model has
    ~590 directions per edge, too many to enumerate in a test; a sample of them is not."""
    n = nstates
    preorder_nodes, indices, indptr, esd = get_expm_augmented_transitions(
        T, root, n, Q_default=Q_default)
    mask = define_state_mask(node_to_allowed_states, preorder_nodes, n)
    _, pmap = esd_get_node_to_pmap(indices, indptr, esd, mask)
    distn = mc0_esd_get_node_to_distn(indices, indptr, esd, root_distn, pmap)
    J = mc0_esd_get_joint_endpoint_distn(indices, indptr, esd, pmap, distn)
    index = dict((v, i) for i, v in enumerate(preorder_nodes))
    out = dict(((int(c), int(d)), 0.0) for c, d in pairs)
    for na, nb in nx.bfs_edges(T, root):
        edge = T[na][nb]
        Q = np.asarray(edge.get('Q', Q_default), dtype=float)
        t = edge['weight']
        Pe, Je = esd[index[nb]], J[index[nb]]
        live = Je != 0
        ratio = np.zeros((n, n))
        ratio[live] = Je[live] / Pe[live]
        for c, d in out:
            if c != d and not Q[c, d]:
                continue
            C = np.zeros((n, n))
            C[c, d] = 1.0
            interact = scipy.linalg.expm_frechet(t * Q, t * C, compute_expm=False)
            total = float(np.sum(ratio[live] * interact[live]))
            out[c, d] += total if c == d else Q[c, d] * total
    return out, distn[0]


# ---------------------------------------------------------------------------
# batched forms (vectorised over sites) used by the parity tests and the
# "amortised" CPU baseline
# ---------------------------------------------------------------------------

def batch_upward(indices, indptr, esd, obs_nodes, obs_lik, root_distn=None):
    """Upward pass for many sites at once.

    obs_nodes : int[K] preorder indices of the nodes that carry per-site data
    obs_lik   : f64[nsites, K, n] likelihood (or 0/1 mask) per observed node
    Returns (lik f64[nsites], root_pmap f64[nsites, n]).  Same arithmetic as
    ``mcy_esd_get_node_to_pmap`` with the mask folded into ``obs_lik``."""
    nnodes = len(indptr) - 1
    nsites = obs_lik.shape[0]
    nstates = esd.shape[1]
    slot = dict((int(v), k) for k, v in enumerate(obs_nodes))
    msgs = [None] * nnodes
    for v in range(nnodes - 1, -1, -1):
        acc = None
        for c in indices[indptr[v]:indptr[v + 1]]:
            t = msgs[c] @ esd[c].T
            msgs[c] = None
            acc = t if acc is None else acc * t
        if acc is None:
            acc = np.ones((nsites, nstates), dtype=float)
        if v in slot:
            acc = acc * obs_lik[:, slot[v], :]
        msgs[v] = acc
    root_pmap = msgs[0]
    if root_distn is None:
        lik = root_pmap.sum(axis=1)
    else:
        lik = root_pmap @ np.asarray(root_distn, dtype=float)
    return lik, root_pmap


def batch_log_likelihoods(indices, indptr, esd, obs_nodes, obs_lik,
                          root_distn=None):
    """log-likelihood per site + status (0 ok, 1 zero probability: the
    reference raises ``StructuralZeroProb`` for these sites,
    ``_mc0_dense.py:190-203``)."""
    lik, _ = batch_upward(indices, indptr, esd, obs_nodes, obs_lik, root_distn)
    status = (~(lik > 0)).astype(np.int32)
    with np.errstate(divide='ignore', invalid='ignore'):
        ll = np.where(lik > 0, np.log(np.where(lik > 0, lik, 1.0)), -np.inf)
    return ll, status


def reference_faithful_site_loglik(T, root, nstates, node_to_allowed_states,
                                   root_distn=None, Q_default=None):
    """One site exactly the way the reference does it: E expm calls + nx
    marshalling + three passes + root reduce (``_mjp_dense.py:362-407``).
    This is the denominator of the north-star speed-up."""
    lik = mjp_dense_get_likelihood(T, node_to_allowed_states, root, nstates,
                                   root_distn=root_distn, Q_default=Q_default)
    return np.log(lik)


# ---------------------------------------------------------------------------
# pyfelscore.get_lb_transition_matrix (examples/p53/liwen.py:45)
# ---------------------------------------------------------------------------

def getp_lb(Q, t):
    """examples/p53/liwen.py:47-82 (``getp_lb``, the pure-Python twin of
    ``pyfelscore.get_lb_transition_matrix``): entry (a, a) = exp(t Q[a, a]), the probability
    of no change; entry (a, b) = the probability of exactly one change, of type a -> b:
    the integral over its time x of exp(-ra x) rab exp(-rb (t - x)).

    Parity status of THIS function: restated from the text of liwen.py, which cannot be
    imported here (it needs ``dendropy``, absent) -- "parity unpinned" against the reference's
    own output; tests/test_oracle_golden.py pins it to the integral the reference's comment
    states (numerical quadrature) and to expm(Q t) as an upper bound."""
    Q = np.asarray(Q, dtype=float)
    n = Q.shape[0]
    P = np.zeros_like(Q)
    for sa in range(n):
        for sb in range(n):
            if sa == sb:
                p = np.exp(t * Q[sa, sb])
            else:
                rab = Q[sa, sb]
                if rab:
                    ra, rb = -Q[sa, sa], -Q[sb, sb]
                    if ra == rb:
                        p = rab * t * np.exp(-rb * t)
                    else:
                        p = rab * ((np.exp(-ra * t) - np.exp(-rb * t)) / (rb - ra))
                else:
                    p = 0.0
            P[sa, sb] = p
    return P
