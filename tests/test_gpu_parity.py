"""
GPU parity tests (run on an MI355X through gpurun): every check goes through the
C ABI of libraoteh_hip.so and compares with the oracle (oracle/oracle_numpy.py,
pinned to the reference by tests/test_oracle_golden.py) or directly with the
golden vectors generated from the reference.

Tolerance: the north star asks for 1e-10 relative on log-likelihood; the tests
use RTOL_LL = 1e-10 for log-likelihoods and 1e-11..1e-12 where the arithmetic
is a plain re-ordering of the same f64 sums.  Masks / statuses are bit-exact.
"""
import itertools
import os
import sys
import warnings

import networkx as nx
import numpy as np
import pytest

from conftest import (expectation_cases, load_golden, tree_from_edges, config_from_golden,
                      switching_cases)
from oracle import oracle_numpy as orc

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-10


@pytest.fixture(scope='module')
def ra():
    import raoteh_amd
    from raoteh_amd import (_mjp_dense, _mcy_dense, _mcx_dense, _mcz_dense as _mcz, device,
                            _lib, synth, pyfelscore_compat)
    class NS(object):
        pass
    ns = NS()
    ns.pkg = raoteh_amd
    ns.mjp, ns.mcy, ns.mcx, ns.mcz = _mjp_dense, _mcy_dense, _mcx_dense, _mcz
    ns.device, ns.lib, ns.synth, ns.pyf = device, _lib, synth, pyfelscore_compat
    ns.ctx = device.get_context()
    # the suite checks which kernel a batch runs right after creating it: compile inside
    # rt_sites_create (the background path has its own test below)
    _lib.check(_lib.lib().rt_set_option(b'jit_async', 0))
    return ns


# ---------------------------------------------------------------------------
# expm
# ---------------------------------------------------------------------------

def test_expm_matches_scipy_fixture(ra):
    # reference tests/test_expm.py:20-82 families + model matrices, expected
    # values = scipy.linalg.expm as called at _mjp_dense.py:24-25
    fx = load_golden('expm')
    by_n = {}
    for row in fx['rows']:
        by_n.setdefault(len(row['Q']), []).append(row)
    for n, rows in by_n.items():
        Q = np.array([r['Q'] for r in rows])
        t = np.array([r['t'] for r in rows])
        P, info = ra.ctx.expm(Q, t, return_info=True)
        for k, r in enumerate(rows):
            want = np.array(r['P'])
            # scipy's own documented accuracy is relative to the matrix norm
            np.testing.assert_allclose(P[k], want, rtol=1e-10,
                                       atol=1e-14 * max(1.0, np.abs(want).max()),
                                       err_msg='%s t=%g' % (r['form'], r['t']))
            assert np.abs(P[k].sum(axis=1) - 1).max() < 1e-12
            m, s = orc.device_expm_order_and_squarings(
                n, np.abs(np.array(r['Q']) * r['t']).sum(axis=0).max())
            assert tuple(info[k]) == (m, s)


def test_pade_expm_kernels_match_scipy_fixture(ra, monkeypatch):
    """RAOTEH_EXPM=pade: the [m/m] Pade scaling-and-squaring kernels (what north_star names;
    the default is the Taylor / Paterson-Stockmeyer scheme) against the same
    stored scipy matrices, plus a codon-model likelihood through rt_model_set_rates."""
    monkeypatch.setenv('RAOTEH_EXPM', 'pade')
    fx = load_golden('expm')
    by_n = {}
    for row in fx['rows']:
        by_n.setdefault(len(row['Q']), []).append(row)
    for n, rows in by_n.items():
        if n > 64:
            continue                      # the Pade kernel keeps its matrices in LDS: n <= 64
        Q = np.array([r['Q'] for r in rows])
        t = np.array([r['t'] for r in rows])
        P, info = ra.ctx.expm(Q, t, return_info=True)
        for k, r in enumerate(rows):
            want = np.array(r['P'])
            np.testing.assert_allclose(P[k], want, rtol=1e-10,
                                       atol=1e-14 * max(1.0, np.abs(want).max()),
                                       err_msg='pade %s t=%g' % (r['form'], r['t']))
            m, s = orc.pade_order_and_squarings(np.abs(np.array(r['Q']) * r['t']).sum(axis=0).max())
            assert tuple(info[k]) == (m, s)
    fx = load_golden('config_c3')
    T, root, n, Q_default, distn, sites = config_from_golden(fx)
    for site, want in zip(sites, fx['log_likelihoods']):
        lk = ra.mjp.get_likelihood(T, site, root, n, root_distn=distn, Q_default=Q_default)
        assert np.log(lk) == pytest.approx(want, rel=RTOL_LL)


@pytest.mark.parametrize('n', [49, 61, 64])
def test_two_workgroup_expm_is_bit_identical(ra, n, monkeypatch):
    """csrc/expm.hip, SPLIT: few 49..64-state matrices (the edges of one tree) take two
    workgroups each, the Horner steps in column halves.  Same arithmetic per entry: the
    transition matrices, the A fragments the pruning kernels read (through the
    log-likelihoods) and the order / squarings are those of the one-workgroup kernel, also
    where squarings force both workgroups through the whole chain."""
    rng = np.random.RandomState(n)
    T, root, leaves = ra.synth.balanced_tree(16)
    Q = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) < 0.4)
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(axis=1))
    Q /= np.abs(np.diag(Q)).mean()
    # branch lengths over every degree and into the squarings
    ts = [1e-6, 1e-3, 0.01, 0.05, 0.1, 0.2, 0.6, 1.5, 7.0]
    for k, (a, b) in enumerate(T.edges()):
        T[a][b]['weight'] = ts[k % len(ts)]
    dense = rng.uniform(0.1, 1.0, size=(40, len(leaves), n))
    out = {}
    for split in ('0', '1'):
        monkeypatch.setenv('RAOTEH_EXPM_SPLIT', split)
        model = ra.device.TreeModel(T, root, n)
        model.set_rates(Q_default=Q)
        name = ra.ctx.kernel_time(0)[2]
        assert name == ('expm_taylor_ps_mfma_split2' if split == '1' else 'expm_taylor_ps_mfma'), name
        batch = model.upload_sites(leaves, dense, kind='dense')
        ll, _ = model.log_likelihoods(batch)
        model.step(batch)
        ll2, _ = model.fetch_log_likelihoods(batch)
        np.testing.assert_array_equal(ll, ll2)
        out[split] = (model.get_transitions(), model.expm_info(), ll)
    np.testing.assert_array_equal(out['0'][1], out['1'][1])
    assert set(out['0'][1][1:, 0]) >= {3, 6, 9, 12, 15} and out['0'][1][:, 1].max() >= 2
    np.testing.assert_array_equal(out['0'][0], out['1'][0])
    np.testing.assert_array_equal(out['0'][2], out['1'][2])
    import scipy.linalg
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, root, n, Q_default=Q)
    np.testing.assert_allclose(out['1'][0][1:], esd[1:], rtol=1e-9, atol=1e-14)


def test_spectral_reconstruction_matches_the_reference_qtop(ra):
    """csrc/spectral.hip through rt_expm_spectral and rt_model_set_rates_spectral against
    examples/p53/qtop.py's own getp_spectral_v2 outputs (tests/golden/spectral.json), and
    the likelihood of a tree whose edges come from the spectral path against the one whose
    edges come from expm."""
    from raoteh_amd import _spectral
    fx = load_golden('spectral')
    for c in fx['cases']:
        D, A, lam, B = (np.array(c[k]) for k in ('D', 'A', 'lam', 'B'))
        n = c['n']
        want = np.array(c['P_spectral'])
        # entries are sums of n products of magnitude <= max|A| max|B|
        atol = 1e-15 * n * np.abs(A).max() * np.abs(B).max()
        got = _spectral.getp_spectral_v2(D, A, lam, B, np.array(c['t']))
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=atol)
        np.testing.assert_allclose(_spectral.getp_spectral_v2(D, A, lam, B, c['t'][0]), want[0],
                                   rtol=1e-12, atol=atol)
        assert (got[:, D == 0, D == 0] == 1).all()
        # the resident path: a random tree, P of every edge in the layouts the pruning
        # kernels read; likelihoods against the same tree with the fixture's matrices set
        # directly
        rng = np.random.RandomState(n)
        T, root, leaves = ra.synth.random_tree(12, seed=n)
        for a, b in T.edges():
            T[a][b]['weight'] = float(rng.choice(c['t']))
        model = ra.device.TreeModel(T, root, n)
        model.set_rates_spectral(A, lam, B, D=D)
        esd = model.get_transitions()
        ts = model.tree.branch_lengths()
        for v in range(1, model.tree.nnodes):
            k = c['t'].index(ts[v])
            np.testing.assert_allclose(esd[v], want[k], rtol=1e-12, atol=atol)
        assert not esd[0].any()
        w = np.where(D > 0, D, 0.0)
        model.set_root_distn(w)
        states = rng.randint(1, n, size=(40, len(leaves))).astype(np.uint8)
        batch = model.upload_sites(leaves, states, kind='state')
        ll, st = model.log_likelihoods(batch)
        model.step(batch)                      # rebuilds the edges from the decomposition
        ll_step, _ = model.fetch_log_likelihoods(batch)
        np.testing.assert_array_equal(ll, ll_step)
        twin = ra.device.TreeModel(T, root, n)
        twin.set_transitions(esd)
        twin.set_root_distn(w)
        ll2, st2 = twin.log_likelihoods(twin.upload_sites(leaves, states, kind='state'))
        np.testing.assert_array_equal(ll, ll2)
        # a tree-specialised kernel on the same batch (for n > 4 one whose leaves are gathered
        # columns: the model's leaf-column table must follow the spectral rebuild too)
        ra.lib.check(ra.lib.lib().rt_set_option(b'jit', 1))
        try:
            bj = model.upload_sites(leaves, states, kind='state')
            model.step(bj)
            llj, _ = model.fetch_log_likelihoods(bj)
            assert 'jit' in bj.kernel_name
        finally:
            ra.lib.check(ra.lib.lib().rt_set_option(b'jit', -1))
        np.testing.assert_array_equal(llj, ll)
        if 'P_expm' in c and n > 4:
            # ... and against expm of the same rate matrix (Q = S diag(D) = A diag(lam) B)
            Q = (A * lam[None, :]) @ B
            Q[D == 0] = 0.0
            full = ra.device.TreeModel(T, root, n)
            full.set_rates(Q=Q, t=ts)
            full.set_root_distn(w)
            ll3, _ = full.log_likelihoods(full.upload_sites(leaves, states, kind='state'))
            ok = st == 0
            # the spectral form holds entries of P to an ABSOLUTE 1e-14..1e-13 (qtop.py:606
            # tests atol = 1e-14 at n = 4; cancellation among n terms of size |A||B|), the
            # expm kernels hold small entries relatively: on these random matrices (stationary
            # weights down to 1e-4) and on uniformly random leaf states (for the codon matrix:
            # three-substitution entries of P(0.004) ~ 1e-9) the log-likelihoods agree to ~1e-7;
            # on data simulated from the model to 1e-11 (bench.py, spectral_step_c3)
            np.testing.assert_allclose(ll[ok], ll3[ok], rtol=1e-5)
    with pytest.raises(Exception):
        ra.ctx.expm_spectral(np.eye(70), np.zeros(70), np.eye(70), [0.1])


def test_expm_against_own_algorithm_restated(ra):
    rng = np.random.RandomState(7)
    for n in (1, 2, 5, 17, 32, 47, 62):
        Q = rng.exponential(size=(6, n, n))
        for q in Q:
            np.fill_diagonal(q, 0)
            q -= np.diag(q.sum(axis=1))
        t = np.array([1e-3, 0.02, 0.3, 1.0, 3.0, 11.0])
        P = ra.ctx.expm(Q, t)
        for k in range(6):
            want = orc.device_expm_restated(Q[k], t[k])
            np.testing.assert_allclose(P[k], want, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(P[k], orc.custom_expm(Q[k], t[k]),
                                       rtol=1e-8, atol=1e-13)


def test_expm_orders_above_64_through_global_scratch(ra):
    # 64 < n <= 128: the four matrices of the Taylor kernel live in an L2-resident slice of
    # global scratch per workgroup (the Frechet blocks of the 61-state codon model have
    # order 122); same algorithm, checked against scipy and the restated algorithm
    rng = np.random.RandomState(17)
    for n in (65, 80, 97, 122, 128):
        Q = rng.exponential(size=(3, n, n)) * (rng.uniform(size=(3, n, n)) < 0.2)
        for q in Q:
            np.fill_diagonal(q, 0)
            q -= np.diag(q.sum(axis=1))
            q /= np.abs(np.diag(q)).max()
        t = np.array([0.003, 0.4, 7.0])
        P, info = ra.ctx.expm(Q, t, return_info=True)
        for k in range(3):
            want = orc.custom_expm(Q[k], t[k])
            np.testing.assert_allclose(P[k], want, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(P[k], orc.expm_taylor(Q[k], t[k]), rtol=1e-9, atol=1e-13)
            assert np.abs(P[k].sum(axis=1) - 1).max() < 1e-12
            assert tuple(info[k]) == orc.taylor_order_and_squarings(
                np.abs(Q[k] * t[k]).sum(axis=0).max())
    # a block-triangular matrix of order 122 gives the Frechet derivative in its corner
    n = 61
    Qc, _ = ra.synth.mg94()
    E = rng.uniform(size=(n, n))
    B = np.zeros((2 * n, 2 * n))
    B[:n, :n] = B[n:, n:] = 0.1 * Qc
    B[:n, n:] = E
    got = ra.ctx.expm(B, [1.0])[0]
    import scipy.linalg
    want_P, want_L = scipy.linalg.expm_frechet(0.1 * Qc, E)
    np.testing.assert_allclose(got[:n, :n], want_P, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(got[:n, n:], want_L, rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize('n', [65, 80, 81, 100, 112, 122, 128])
def test_lds_resident_expm_above_64_is_bit_identical_to_the_global_form(ra, n, monkeypatch):
    """expm_wide.hip (one matrix in LDS in B-fragment order, A operands and accumulators in
    registers; two workgroups per matrix when the matrices are few) against the global-scratch
    kernel it replaces: the same arithmetic per entry, so the same bits -- one workgroup per
    matrix and two, without and with squarings, several rate matrices (the root slot of a
    model's launch: the tree tests at n > 64)."""
    rng = np.random.RandomState(900 + n)
    nq = 3
    Q = rng.exponential(size=(nq, n, n)) * (rng.uniform(size=(nq, n, n)) < 0.3)
    for q in Q:
        np.fill_diagonal(q, 0)
        q -= np.diag(q.sum(axis=1))
        q /= np.abs(np.diag(q)).max()
    count = 11
    t = np.concatenate([[1e-6, 0.004, 0.04, 0.12, 0.3], rng.uniform(0.3, 9.0, size=count - 5)])
    qidx = rng.randint(0, nq, size=count)
    out = {}
    for wide, split in ((0, 0), (1, 0), (1, 1)):
        monkeypatch.setenv('RAOTEH_EXPM_WIDE', str(wide))
        monkeypatch.setenv('RAOTEH_EXPM_SPLIT', str(split))
        out[wide, split] = ra.ctx.expm(Q, t, q_index=qidx, return_info=True)
        ra.ctx.set_timing(True)
        ra.ctx.expm(Q, t, q_index=qidx)
        name = ra.ctx.kernel_time(0)[2]
        ra.ctx.set_timing(False)
        assert name == ('expm_taylor_ps_mfma_global' if not wide else
                        'expm_taylor_ps_mfma_wide_split2' if split else 'expm_taylor_ps_mfma_wide')
    monkeypatch.delenv('RAOTEH_EXPM_WIDE')
    monkeypatch.delenv('RAOTEH_EXPM_SPLIT')
    P0, info0 = out[0, 0]
    assert info0[:, 1].max() >= 2 and info0[:5, 1].min() == 0 and len(set(info0[:, 0])) >= 3
    for key in ((1, 0), (1, 1)):
        np.testing.assert_array_equal(out[key][1], info0)
        np.testing.assert_array_equal(out[key][0], P0)
    k = 6
    np.testing.assert_allclose(P0[k], orc.custom_expm(Q[qidx[k]], t[k]), rtol=1e-9, atol=1e-13)


def test_expm_shared_q_and_errors(ra):
    Q, _ = ra.synth.hky85()
    t = np.linspace(0.01, 2.0, 50)
    P = ra.ctx.expm(Q, t)                     # one Q, many t
    for k in (0, 17, 49):
        np.testing.assert_allclose(P[k], orc.custom_expm(Q, t[k]), rtol=1e-11,
                                   atol=1e-15)
    np.testing.assert_allclose(ra.mjp.custom_expm(Q, 0.3),
                               orc.custom_expm(Q, 0.3), rtol=1e-11, atol=1e-15)
    with pytest.raises(ValueError):
        ra.mjp.custom_expm(np.zeros((3, 4)), 1.0)
    with pytest.raises(ValueError):
        ra.mjp.custom_expm(None, 1.0)
    with pytest.raises(Exception):
        ra.ctx.expm(np.zeros((129, 129)), [1.0])   # > RT_MAX_EXPM_STATES
    out = np.empty((3, 3))
    Q3 = np.array([[-1., 1, 0], [2, -5, 3], [0, 0, 0]])
    ra.pyf.get_tolerance_rate_matrix(0.7, Q3, out)
    np.testing.assert_allclose(out, orc.custom_expm(Q3, 0.7), rtol=1e-11,
                               atol=1e-16)


# ---------------------------------------------------------------------------
# the three reference-format passes
# ---------------------------------------------------------------------------

def test_passes_match_reference_golden(ra):
    fx = load_golden('random_sparse')
    for c in fx['cases']:
        n = c['nstates']
        T = tree_from_edges(c['edges'], nodes=c['nodes'])
        root = c['root']
        for na, nb in nx.bfs_edges(T, root):
            T[na][nb]['P'] = np.array(c['P'][str(nb)])
        allowed = dict((int(k), set(v)) for k, v in c['allowed'].items())
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        mask = orc.define_state_mask(allowed, pre, n)
        ra.pyf.mcy_esd_get_node_to_pset(idx, ptr, esd, mask)
        for i, v in enumerate(pre):
            assert set(np.flatnonzero(mask[i])) == set(c['pset'][str(v)])
        ra.pyf.esd_get_node_to_set(idx, ptr, esd, mask)
        for i, v in enumerate(pre):
            assert set(np.flatnonzero(mask[i])) == set(c['set'][str(v)])
        pmap = np.empty((len(pre), n))
        ra.pyf.mcy_esd_get_node_to_pmap(idx, ptr, esd, mask, pmap)
        for i, v in enumerate(pre):
            np.testing.assert_allclose(pmap[i], c['pmap'][str(v)], rtol=1e-13,
                                       atol=0)
        # module-level mirrors
        pm = ra.mcy.get_node_to_pmap(T, root, n, node_to_allowed_states=allowed)
        for v in pre:
            np.testing.assert_allclose(pm[v], c['pmap'][str(v)], rtol=1e-13)
        distn = np.array(c['root_distn'])
        if c['zero']:
            with pytest.raises(ra.pkg.StructuralZeroProb):
                ra.mcy.get_likelihood(T, root, n, node_to_allowed_states=allowed,
                                      root_distn=distn)
        else:
            lk = ra.mcy.get_likelihood(T, root, n,
                                       node_to_allowed_states=allowed,
                                       root_distn=distn)
            assert lk == pytest.approx(c['likelihood'], rel=1e-13)
        # downward pass + joint endpoint distributions (values from the reference's
        # _mc0.get_node_to_distn / get_joint_endpoint_distn)
        if 'distn' in c:
            from raoteh_amd import _mc0_dense
            dn = _mc0_dense.get_node_to_distn(T, root, pm, n, root_distn=distn)
            TJ = _mc0_dense.get_joint_endpoint_distn(T, root, pm, dn, n)
            for v in pre:
                np.testing.assert_allclose(dn[v], c['distn'][str(v)], rtol=1e-12,
                                           atol=1e-300)
            for na, nb in nx.bfs_edges(T, root):
                np.testing.assert_allclose(TJ[na][nb]['J'],
                                           np.array(c['joint'][str(nb)]),
                                           rtol=1e-12, atol=1e-300)
            mask2, pm2, dn2, ej = ra.mcy.kitchen_sink(
                T, root, n, node_to_allowed_states=allowed, root_distn=distn)
            for v in pre:
                np.testing.assert_allclose(dn2[v], c['distn'][str(v)], rtol=1e-12,
                                           atol=1e-300)
            for (na, nb), Jm in ej.items():
                np.testing.assert_allclose(Jm, np.array(c['joint'][str(nb)]),
                                           rtol=1e-12, atol=1e-300)
        # type z (_mcz.py:140-163)
        obs = dict((int(k), dict(enumerate(v))) for k, v in c['obs_lik'].items())
        nset = dict((int(k), set(v)) for k, v in c['set'].items())
        pz = ra.mcz.get_node_to_pmap(T, root, n, node_to_state_to_likelihood=obs,
                                     node_to_set=nset)
        for v in pre:
            np.testing.assert_allclose(pz[v], c['pmap_z'][str(v)], rtol=1e-13)


def test_downward_pass_batched_and_zero_denominator(ra):
    rng = np.random.RandomState(8)
    T, root, leaves = ra.synth.random_tree(17, seed=9)
    n = 5
    for na, nb in nx.bfs_edges(T, root):
        P = rng.exponential(size=(n, n))
        T[na][nb]['P'] = P / P.sum(axis=1, keepdims=True)
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    nsites = 21
    masks = (rng.uniform(size=(nsites, len(pre), n)) > 0.2).astype(np.int64)
    masks[:, :, 0] = 1
    pmap = np.empty(masks.shape)
    ra.ctx.node_to_pmap(idx, ptr, esd, masks, pmap)
    w = rng.uniform(0.1, 1, size=n)
    dn, st = ra.ctx.node_to_distn(idx, ptr, esd, w, pmap)
    J = ra.ctx.joint_endpoint_distn(idx, ptr, esd, pmap, dn)
    assert not st.any()
    for s in range(nsites):
        want = orc.mc0_esd_get_node_to_distn(idx, ptr, esd, w, pmap[s])
        np.testing.assert_allclose(dn[s], want, rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(
            J[s], orc.mc0_esd_get_joint_endpoint_distn(idx, ptr, esd, pmap[s], want),
            rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(dn[s].sum(axis=1), 1.0, rtol=1e-12)
        # the joint of an edge marginalises to the two node distributions
        for v in range(len(pre)):
            for c in idx[ptr[v]:ptr[v + 1]]:
                np.testing.assert_allclose(J[s, c].sum(axis=1), dn[s, v], rtol=1e-11,
                                           atol=1e-15)
                np.testing.assert_allclose(J[s, c].sum(axis=0), dn[s, c], rtol=1e-11,
                                           atol=1e-15)
    # a site whose root pmap is all zero: status 2 <-> NumericalZeroProb
    pmap[3, 0, :] = 0.0
    dn, st = ra.ctx.node_to_distn(idx, ptr, esd, w, pmap)
    assert st[3] == 2 and not st[[0, 1, 2, 4]].any()
    with pytest.raises(orc.NumericalZeroProb):
        orc.mc0_esd_get_node_to_distn(idx, ptr, esd, w, pmap[3])


def test_passes_batched_over_sites(ra):
    rng = np.random.RandomState(3)
    T, root, leaves = ra.synth.random_tree(23, seed=5)
    n = 6
    for na, nb in nx.bfs_edges(T, root):
        P = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) > 0.3)
        P[np.arange(n), np.arange(n)] += 0.1
        T[na][nb]['P'] = P / P.sum(axis=1, keepdims=True)
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    nsites = 37
    masks = (rng.uniform(size=(nsites, len(pre), n)) > 0.25).astype(np.int64)
    want = masks.copy()
    wantp = np.empty(masks.shape)
    for s in range(nsites):
        orc.mcy_esd_get_node_to_pset(idx, ptr, esd, want[s])
        orc.esd_get_node_to_set(idx, ptr, esd, want[s])
        orc.mcy_esd_get_node_to_pmap(idx, ptr, esd, want[s], out=wantp[s])
    got = masks.copy()
    ra.ctx.node_to_pset(idx, ptr, esd, got)
    ra.ctx.node_to_set(idx, ptr, esd, got)
    np.testing.assert_array_equal(got, want)
    gotp = np.empty(masks.shape)
    ra.ctx.node_to_pmap(idx, ptr, esd, got, gotp)
    np.testing.assert_allclose(gotp, wantp, rtol=1e-13, atol=0)


# ---------------------------------------------------------------------------
# reference-shaped single-site API
# ---------------------------------------------------------------------------

def test_rerooting_known_answer(ra):
    fx = load_golden('test_mjp_rerooting')        # tests/test_mjp.py:91-164
    T = tree_from_edges(fx['edges'])
    Q = np.array(fx['Q'])
    distn = np.array(fx['root_distn'])
    n = fx['nstates']
    allowed = dict((v, set(range(n))) for v in T)
    for k, s in fx['node_to_state'].items():
        allowed[int(k)] = {s}
    for r in fx['rootings']:
        lk = ra.mjp.get_likelihood(T, allowed, r['root'], n, root_distn=distn,
                                   Q_default=Q)
        assert lk == pytest.approx(r['likelihood'], rel=1e-11)
        assert lk == pytest.approx(0.002296828148732273, rel=1e-11)
    # type-x mirror on the augmented tree
    T_aug = ra.mjp.get_expm_augmented_tree(T, 0, Q_default=Q)
    nts = dict((int(k), s) for k, s in fx['node_to_state'].items())
    lk = ra.mcx.get_likelihood(T_aug, 0, n, node_to_state=nts, root_distn=distn)
    assert lk == pytest.approx(0.002296828148732273, rel=1e-11)


def test_sum_to_one(ra):
    fx = load_golden('sum_to_one')                # tests/test_mjp.py:52-89
    T = tree_from_edges(fx['edges'])
    Q = np.array(fx['Q'])
    distn = np.array(fx['root_distn'])
    total = 0.0
    for assignment, want in zip(fx['assignments'], fx['likelihoods']):
        allowed = dict((v, {s}) for v, s in enumerate(assignment))
        lk = ra.mjp.get_likelihood(T, allowed, 0, 3, root_distn=distn,
                                   Q_default=Q)
        assert lk == pytest.approx(want, rel=1e-10)
        total += lk
    assert total == pytest.approx(1.0, rel=1e-11)
    # the same 81 assignments as one batch through the fast path
    obs_nodes = [0, 1, 2, 3]
    states = np.array(fx['assignments'], dtype=np.uint8)
    ll, st = ra.mjp.get_log_likelihoods(T, 0, 3, obs_nodes, states, kind='state',
                                        root_distn=distn, Q_default=Q)
    assert not st.any()
    np.testing.assert_allclose(ll, np.log(fx['likelihoods']), rtol=RTOL_LL)
    assert np.exp(ll).sum() == pytest.approx(1.0, rel=1e-11)


@pytest.mark.parametrize('nprimary,parts', [(6, (0, 0, 1, 1, 2, 2)), (5, (0, 0, 0, 1, 1)),
                                            (6, (0, 1, 2, 3, 3, 3))])
def test_compound_tolerance_model_leaf_marginals_sum_to_one(ra, nprimary, parts):
    """The reference's tests/test_tmjp.py:201-283 (slow there: one get_likelihood call per
    leaf pattern): on its 6-node tree the probabilities of ALL nprimary^4 primary-state leaf
    patterns under the compound tolerance process add up to one.  Here the patterns are one
    batch of allowed-set masks (48 / 20 / 96 compound states: split-M, one-wave and two-word
    kernels), every pattern also against the oracle."""
    import itertools
    rng = np.random.RandomState(40 + nprimary + len(set(parts)))
    Qp = rng.exponential(size=(nprimary, nprimary))
    np.fill_diagonal(Qp, 0.0)
    Qp -= np.diag(Qp.sum(axis=1))
    pd = rng.dirichlet(np.ones(nprimary))
    primary_to_part = dict(enumerate(parts))
    Qc, dc = ra.synth.blinking_model(Qp, pd, primary_to_part, 0.5, 1.5)    # rate_on, rate_off (:206-207)
    n = Qc.shape[0]
    assert n == nprimary * 2 ** len(set(parts)) and dc.sum() == pytest.approx(1.0, rel=1e-12)
    T = nx.Graph()
    for a, b in ((0, 1), (0, 2), (0, 3), (3, 4), (3, 5)):                  # :226-231
        T.add_edge(a, b, weight=rng.exponential(scale=0.1))
    leaves = [1, 2, 4, 5]
    patterns = np.array(list(itertools.product(range(nprimary), repeat=4)))
    words = (n + 63) // 64
    per_state = np.zeros((nprimary, words), dtype=np.uint64)
    for c in range(nprimary):
        for k in ra.synth.blinking_allowed_states(c, nprimary, primary_to_part):
            per_state[c, k // 64] |= np.uint64(1) << np.uint64(k % 64)
    masks = per_state[patterns]                                            # [nsites][4][words]
    ll, st = ra.mjp.get_log_likelihoods(T, 0, n, leaves, masks, kind='mask', root_distn=dc,
                                        Q_default=Qc)
    assert not st.any()
    assert np.exp(ll).sum() == pytest.approx(1.0, rel=1e-11)
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, 0, n, Q_default=Qc)
    dense = np.zeros((len(patterns), 4, n))
    for c in range(nprimary):
        ii, kk = np.nonzero(patterns == c)
        for a in ra.synth.blinking_allowed_states(c, nprimary, primary_to_part):
            dense[ii, kk, a] = 1.0
    want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in leaves], dense, dc)
    assert not wst.any()
    np.testing.assert_allclose(ll, want, rtol=RTOL_LL)
    # the reference's single-site call on a few patterns
    for k in (0, len(patterns) // 3, len(patterns) - 1):
        allowed = dict((v, set(range(n))) for v in T)
        for leaf, c in zip(leaves, patterns[k]):
            allowed[leaf] = set(ra.synth.blinking_allowed_states(c, nprimary, primary_to_part))
        lk = ra.mjp.get_likelihood(T, allowed, 0, n, root_distn=dc, Q_default=Qc)
        assert np.log(lk) == pytest.approx(ll[k], rel=1e-10)


def test_kat_four_log_half(ra):
    fx = load_golden('kat_history')               # tests/test_mc.py:131-150
    T = tree_from_edges(fx['edges'])
    P = np.array(fx['P'])
    allowed = dict((int(k), {v}) for k, v in fx['node_to_state'].items())
    lk = ra.mcy.get_likelihood(T, 0, 3, node_to_allowed_states=allowed,
                               root_distn=np.array(fx['root_distn']),
                               P_default=P)
    assert np.log(lk) == pytest.approx(4 * np.log(0.5), rel=1e-15)


def test_single_node_tree_and_errors(ra):
    T = nx.Graph()
    T.add_node(7)
    Q = np.array([[-1.0, 1.0], [2.0, -2.0]])
    assert ra.mjp.get_likelihood(T, {7: {0}}, 7, 2, None, Q) == 1
    lk = ra.mjp.get_likelihood(T, {7: {0, 1}}, 7, 2, np.array([0.25, 0.5]), Q)
    assert lk == pytest.approx(0.75)
    with pytest.raises(ra.pkg.StructuralZeroProb):
        ra.mjp.get_likelihood(T, {7: set()}, 7, 2, None, Q)
    with pytest.raises(ValueError):
        ra.mjp.get_likelihood(T, {7: {0}}, 8, 2, None, Q)
    T2 = nx.Graph()
    T2.add_edge(0, 1, weight=0.1)
    with pytest.raises(ValueError):               # no rate matrix at all
        ra.mjp.get_likelihood(T2, {0: {0}, 1: {0}}, 0, 2, None, None)
    with pytest.raises(KeyError):                 # _mcy_dense.py:52
        ra.mjp.get_likelihood(T2, {0: {0}}, 0, 2, None, Q)
    with pytest.raises(ValueError):               # root shape mismatch
        ra.mjp.get_likelihood(T2, {0: {0}, 1: {0}}, 0, 2, np.ones(3), Q)
    # batched API on a single-node tree
    ll, st = ra.mjp.get_log_likelihoods(T, 7, 2, [7], np.array([[0], [1], [255]],
                                        dtype=np.uint8), kind='state',
                                        root_distn=np.array([0.25, 0.5]),
                                        Q_default=Q)
    np.testing.assert_allclose(ll, np.log([0.25, 0.5, 0.75]), rtol=1e-15)


# ---------------------------------------------------------------------------
# batched hot path vs golden configs (values computed by the reference)
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('name', ['c1', 'c2', 'c3', 'c5'])
def test_config_fixtures_batched(ra, name):
    fx = load_golden('config_' + name)
    T, root, n, Q_default, distn, sites = config_from_golden(fx)
    leaves = fx['leaves']
    want = np.array(fx['log_likelihoods'])
    masks = ra.mjp.allowed_states_to_masks(sites, leaves)
    ll, st = ra.mjp.get_log_likelihoods(T, root, n, leaves, masks, kind='mask',
                                        root_distn=distn, Q_default=Q_default)
    assert not st.any()
    np.testing.assert_allclose(ll, want, rtol=RTOL_LL)
    dense = np.zeros((len(sites), len(leaves), n))
    for i, d in enumerate(sites):
        for k, v in enumerate(leaves):
            dense[i, k, sorted(d[v])] = 1.0
    ll2, _ = ra.mjp.get_log_likelihoods(T, root, n, leaves, dense, kind='dense',
                                        root_distn=distn, Q_default=Q_default)
    np.testing.assert_array_equal(ll, ll2)
    if fx['obs_kind'] == 'state':
        states = np.array(fx['leaf_states'], dtype=np.uint8)
        ll3, _ = ra.mjp.get_log_likelihoods(T, root, n, leaves, states,
                                            kind='state', root_distn=distn,
                                            Q_default=Q_default)
        np.testing.assert_array_equal(ll, ll3)
    # single-site reference-shaped call on the first site
    lk = ra.mjp.get_likelihood(T, sites[0], root, n, root_distn=distn,
                               Q_default=Q_default)
    assert np.log(lk) == pytest.approx(want[0], rel=RTOL_LL)
    # device expm vs the scipy matrices stored in the fixture
    model = ra.device.TreeModel(T, root, n)
    model.set_rates(Q_default=Q_default)
    esd = model.get_transitions()
    assert not esd[0].any()
    for k, P in fx['P_scipy'].items():
        i = model.tree.node_to_index[int(k)]
        np.testing.assert_allclose(esd[i], np.array(P), rtol=1e-10, atol=1e-15)


# ---------------------------------------------------------------------------
# fast kernels vs the oracle on seeded random inputs
# ---------------------------------------------------------------------------

def _random_case(ra, rng, n, nnodes, nsites, internal_obs=True, sparse=False):
    T, root, leaves = ra.synth.random_tree(nnodes, seed=int(rng.randint(1 << 30)),
                                           max_children=4)
    ta_nodes = list(T)
    P = {}
    for na, nb in nx.bfs_edges(T, root):
        M = rng.exponential(size=(n, n))
        if sparse:
            M *= rng.uniform(size=(n, n)) > 0.4
            M[np.arange(n), np.arange(n)] += 0.05
        T[na][nb]['P'] = M / M.sum(axis=1, keepdims=True)
    obs_nodes = list(leaves)
    if internal_obs:
        inner = [v for v in ta_nodes if v not in leaves]
        obs_nodes += inner[::2]
    w = rng.uniform(0.0, 1.0, size=n)
    return T, root, obs_nodes, w


@pytest.mark.parametrize('n', [1, 2, 3, 4, 5, 8, 13, 16, 20, 33, 48, 61, 64,
                               65, 77, 96, 100, 113, 122, 128])
def test_fast_kernels_random_trees(ra, n):
    rng = np.random.RandomState(100 + n)
    for nnodes, nsites in ((2, 1), (7, 65), (30, 257), (41, 1000)):
        T, root, obs_nodes, w = _random_case(ra, rng, n, nnodes, nsites,
                                             sparse=(nnodes == 30))
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        oidx = [pre.index(v) for v in obs_nodes]
        model = ra.device.TreeModel(T, root, n)
        model.set_transitions(esd)
        model.set_root_distn(w)
        # type z: arbitrary likelihoods; a few sites forced to zero probability
        dense = rng.uniform(0.05, 1.0, size=(nsites, len(obs_nodes), n))
        dense[rng.uniform(size=dense.shape) < 0.2] = 0.0
        if nsites > 3:
            dense[3, 0, :] = 0.0
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense, w)
        batch = model.upload_sites(obs_nodes, dense, kind='dense')
        ll, st = model.log_likelihoods(batch)
        np.testing.assert_array_equal(st & 1, wst)
        ok = wst == 0
        np.testing.assert_allclose(ll[ok], want[ok], rtol=RTOL_LL)
        assert np.all(np.isneginf(ll[~ok]))
        tot = model.fetch_totals(batch)
        assert tot[1] == (~ok).sum() and tot[2] == nsites
        assert tot[0] == pytest.approx(want[ok].sum(), rel=1e-11, abs=1e-9)
        # type y: bit masks
        if n <= 64:
            bits = rng.randint(1, 1 << min(n, 30), size=(nsites, len(obs_nodes)))
            masks = bits.astype(np.uint64)
            dm = ((masks[..., None] >> np.arange(n, dtype=np.uint64)) & 1
                  ).astype(np.float64)
        else:
            # ceil(n / 64) words per node: bit s % 64 of word s // 64
            dm = (rng.uniform(size=(nsites, len(obs_nodes), n)) < 0.3).astype(np.float64)
            dm[..., rng.randint(n)] = 1.0
            masks = np.zeros((nsites, len(obs_nodes), (n + 63) // 64), dtype=np.uint64)
            for sidx in range(n):
                masks[..., sidx // 64] |= (dm[..., sidx].astype(np.uint64)
                                           << np.uint64(sidx % 64))
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dm, w)
        ll, st = model.log_likelihoods(
            model.upload_sites(obs_nodes, masks, kind='mask'))
        np.testing.assert_array_equal(st & 1, wst)
        np.testing.assert_allclose(ll[wst == 0], want[wst == 0], rtol=RTOL_LL)
        # type x: states with some unobserved
        states = rng.randint(0, n, size=(nsites, len(obs_nodes))).astype(np.uint8)
        states[rng.uniform(size=states.shape) < 0.15] = 255
        dx = np.ones((nsites, len(obs_nodes), n))
        obs_mask = states != 255
        dx[obs_mask] = 0.0
        ii, kk = np.nonzero(obs_mask)
        dx[ii, kk, states[ii, kk]] = 1.0
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dx, w)
        ll, st = model.log_likelihoods(
            model.upload_sites(obs_nodes, states, kind='state'))
        np.testing.assert_array_equal(st & 1, wst)
        np.testing.assert_allclose(ll[wst == 0], want[wst == 0], rtol=RTOL_LL)


@pytest.mark.parametrize('n', [4, 20, 61, 122])
def test_generic_kernel_agrees(ra, n):
    rng = np.random.RandomState(n)
    T, root, obs_nodes, w = _random_case(ra, rng, n, 25, 300)
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    oidx = [pre.index(v) for v in obs_nodes]
    dense = rng.uniform(0.05, 1.0, size=(300, len(obs_nodes), n))
    want, _ = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense, w)
    model = ra.device.TreeModel(T, root, n)
    model.set_transitions(esd)
    model.set_root_distn(w)
    fast, _ = model.log_likelihoods(model.upload_sites(obs_nodes, dense))
    ra.lib.check(ra.lib.lib().rt_set_option(b'force_generic', 1))
    try:
        gen, _ = model.log_likelihoods(model.upload_sites(obs_nodes, dense))
        assert ra.ctx.kernel_time(1)[2].startswith('prune_generic')
    finally:
        ra.lib.check(ra.lib.lib().rt_set_option(b'force_generic', 0))
    np.testing.assert_allclose(gen, want, rtol=RTOL_LL)
    np.testing.assert_allclose(fast, want, rtol=RTOL_LL)


def test_sparse_api_matches_reference_golden(ra):
    """The reference's sparse surface (nx.DiGraph matrices over arbitrary state
    labels, dict results): _mjp (:349-428), _mcy (:323-741), _mcx (:36-256),
    _mcz (:30-209) against values produced by the reference's own pure-Python
    functions (tests/golden/sparse_api.json)."""
    from raoteh_amd import _mjp, _mcy, _mcx, _mcz as mcz_sparse
    from _sparse_cases import build, int_keys
    fx = load_golden('sparse_api')

    def check_pmap(got, want):
        want = int_keys(want)
        assert set(got) == set(want)
        for v, m in want.items():
            assert set(got[v]) == set(int(k) for k in m), v
            for k, x in m.items():
                assert got[v][int(k)] == pytest.approx(x, rel=1e-11, abs=1e-300)

    for c in fx['cases']:
        # continuous time: per-edge expm of the sparse rate matrices
        T, root, Q_default, allowed, node_to_state, root_distn = build(c, False)
        T_aug = _mjp.get_expm_augmented_tree(T, root, Q_default=Q_default)
        for na, nb in nx.bfs_edges(T, root):
            want = dict(((a, b), w) for a, b, w in c['P'][str(nb)])
            P = T_aug[na][nb]['P']
            assert set(P.edges()) == set(want)
            for (a, b), w in want.items():
                assert P[a][b]['weight'] == pytest.approx(w, rel=1e-10, abs=1e-15)
        if c['y_zero']:
            with pytest.raises(ra.pkg.StructuralZeroProb):
                _mjp.get_likelihood(T, allowed, root, root_distn=root_distn,
                                    Q_default=Q_default)
        else:
            lk = _mjp.get_likelihood(T, allowed, root, root_distn=root_distn,
                                     Q_default=Q_default)
            assert lk == pytest.approx(c['y_likelihood'], rel=1e-10)
        # discrete time on the reference's own P digraphs
        T, root, _, allowed, node_to_state, root_distn = build(c, True)
        nset = _mcy.get_node_to_set(T, root, node_to_allowed_states=allowed)
        assert nset == dict((v, set(s)) for v, s in int_keys(c['y_set']).items())
        pset = _mcy.get_node_to_pset(T, root, node_to_allowed_states=allowed)
        for v, s in int_keys(c['y_pset']).items():
            assert set(s) <= pset[v] and nset[v] <= pset[v]
        if not c['y_zero']:
            check_pmap(_mcy.get_node_to_pmap(T, root, node_to_allowed_states=allowed),
                       c['y_pmap'])
            check_pmap(_mcy.get_node_to_pmap(T, root, node_to_allowed_states=allowed,
                                             node_to_set=nset), c['y_pmap'])
            assert _mcy.get_likelihood(
                T, root, node_to_allowed_states=allowed, root_distn=root_distn
            ) == pytest.approx(c['y_likelihood'], rel=1e-11)
            obs = dict((v, int_keys(m)) for v, m in int_keys(c['z_obs']).items())
            check_pmap(mcz_sparse.get_node_to_pmap(
                T, root, node_to_state_to_likelihood=obs), c['z_pmap'])
            check_pmap(mcz_sparse.get_node_to_pmap(
                T, root, node_to_state_to_likelihood=obs, node_to_set=nset),
                c['z_pmap'])
            assert mcz_sparse.get_node_to_set(
                T, root, node_to_state_to_likelihood=obs) == nset
        else:
            with pytest.raises(ra.pkg.StructuralZeroProb):
                _mcy.get_likelihood(T, root, node_to_allowed_states=allowed,
                                    root_distn=root_distn)
        if c['x_zero']:
            with pytest.raises(ra.pkg.StructuralZeroProb):
                _mcx.get_likelihood(T, root, node_to_state=node_to_state,
                                    root_distn=root_distn)
        else:
            check_pmap(_mcx.get_node_to_pmap(T, root, node_to_state=node_to_state),
                       c['x_pmap'])
            assert _mcx.get_likelihood(
                T, root, node_to_state=node_to_state, root_distn=root_distn
            ) == pytest.approx(c['x_likelihood'], rel=1e-11)
    # single-node trees and argument errors (_mcy.py:721-735, _mjp.py:397-398)
    S = nx.Graph()
    S.add_node(9)
    assert _mcy.get_likelihood(S, 9, node_to_allowed_states={9: {1, 2}}) == 1
    assert _mcy.get_likelihood(S, 9, node_to_allowed_states={9: {1, 2}},
                               root_distn={2: 0.25, 5: 0.5}) == 0.25
    with pytest.raises(ra.pkg.StructuralZeroProb):
        _mcy.get_likelihood(S, 9, node_to_allowed_states={9: set()})
    with pytest.raises(ValueError):
        _mjp.get_likelihood(nx.Graph([(0, 1, dict(weight=1.0))]), {}, 5)


@pytest.mark.parametrize('n', [1, 2, 3, 4, 5, 13, 16, 20, 32, 33, 48, 61, 64, 65, 90, 122, 128])
def test_tree_specialised_kernel_is_bit_identical(ra, n):
    """jit.hip: the hiprtc-compiled straight-line kernel for one tree performs
    the interpreter kernel's arithmetic in the interpreter's order."""
    rng = np.random.RandomState(900 + n)
    set_option = ra.lib.lib().rt_set_option
    for nnodes, nsites in ((2, 70), (9, 129), (34, 4000)):
        T, root, obs_nodes, w = _random_case(ra, rng, n, nnodes, nsites,
                                             sparse=(nnodes == 9))
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        oidx = [pre.index(v) for v in obs_nodes]
        dense = rng.uniform(0.05, 1.0, size=(nsites, len(obs_nodes), n))
        dense[rng.uniform(size=dense.shape) < 0.1] = 0.0
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense, w)
        model = ra.device.TreeModel(T, root, n)
        model.set_transitions(esd)
        model.set_root_distn(w)
        out = {}
        # (jit, sites per wave): 64 = the interpreter's blocks; fewer sites per
        # wave change the HBM layout and the partial sums, not a site's value
        # n > 4 (MFMA family): 1..4 site tiles per wave instead
        # (n > 32: tiles per workgroup of NT waves, at most 3)
        # n > 32, 'h': the two root programs as separate workgroups + the combine kernel
        # ('h5': five half-tiles per workgroup)
        variants = (((0, 0), (1, 64), (1, 49), (1, 7)) if n <= 4 else
                    ((0, 0), (1, 64), (1, 2), (1, 3), (1, 4)) if n <= 32 else
                    # ('hu': the halves with the combine step folded into the pruning kernel,
                    # RAOTEH_JIT_FOLD=1)
                    ((0, 0), (1, 64), (1, 2), (1, 3), (1, 'h'), (1, 'h5'), (1, 'hu')) if n <= 64 else
                    # 64 < n <= 128: NT = 5..8 waves share one or two tiles
                    ((0, 0), (1, 64), (1, 2), (1, 'h'), (1, 'hu')))
        for jit, bs in variants:
            ra.lib.check(set_option(b'jit', jit))
            if n <= 4:
                ra.lib.check(set_option(b'jit_block_sites', bs))
            elif jit:
                os.environ['RAOTEH_JIT_TILES'] = str(1 if bs in (64, 'h', 'hu') else 5 if bs == 'h5' else bs)
                os.environ['RAOTEH_JIT_HALVES'] = '1' if bs in ('h', 'h5', 'hu') else '0'
                os.environ['RAOTEH_JIT_FOLD'] = '1' if bs == 'hu' else '0'
            try:
                batch = model.upload_sites(obs_nodes, dense, kind='dense')
                ll, st = model.log_likelihoods(batch)
                name = ra.ctx.kernel_time(1)[2]
                out[jit, bs] = (ll, st, model.fetch_totals(batch), name)
                twin = batch.clone()
                ll2, st2 = model.log_likelihoods(twin)
                np.testing.assert_array_equal(ll, ll2)
            finally:
                ra.lib.check(set_option(b'jit', -1))
                ra.lib.check(set_option(b'jit_block_sites', 0))
                os.environ.pop('RAOTEH_JIT_TILES', None)
                os.environ.pop('RAOTEH_JIT_HALVES', None)
                os.environ.pop('RAOTEH_JIT_FOLD', None)
        assert out[0, 0][3].startswith(('prune_lane', 'prune_mfma')), out[0, 0][3]
        for key in variants[1:]:
            assert out[key][3].startswith('prune_tree_jit'), out[key][3]
            # (a root with one child cannot be cut)
            assert ('halves' in out[key][3]) == (key[1] in ('h', 'h5', 'hu') and T.degree(root) > 1), \
                out[key][3]
            np.testing.assert_array_equal(out[0, 0][0], out[key][0])
            np.testing.assert_array_equal(out[0, 0][1], out[key][1])
            assert out[key][2][1] == out[0, 0][2][1] and out[key][2][2] == nsites
            assert out[key][2][0] == pytest.approx(out[0, 0][2][0], rel=1e-13)
        # (the batch sum adds the per-wave partial sums in a fixed order that depends on where
        # a kernel leaves them: the same places up to 64 states, eight waves apart above)
        if n <= 64:
            np.testing.assert_array_equal(out[0, 0][2], out[1, 64][2])
        out[1] = out[1, 64]
        np.testing.assert_array_equal(out[1][1] & 1, wst)
        np.testing.assert_allclose(out[1][0][wst == 0], want[wst == 0], rtol=RTOL_LL)


def test_switching_model_122_states(ra):
    """The 122-state switching model of examples/p53/liwen.py:599-621 on the p53 tree rooted
    at the leaf 'Has': _mcy_dense.get_likelihood as liwen.py:150-158 calls it, the node
    marginals of :138-148, StructuralZeroProb, and the batched path -- against the
    reference's own numbers (tests/golden/switching.json, unaccelerated _mcy / _mc0 twins
    on scipy's expm) and the oracle."""
    from raoteh_amd import _mc0_dense
    from raoteh_amd._util import StructuralZeroProb
    fx, cases = switching_cases()
    n2 = fx['ncompound']
    seen_zero = False
    for c in cases:
        want = c['want']
        T_aug = ra.mjp.get_expm_augmented_tree(c['T'], c['root'], Q_default=c['Q_compound'])
        if want['log_likelihood'] is None:
            with pytest.raises(StructuralZeroProb):
                ra.mcy.get_likelihood(T_aug, c['root'], n2, node_to_allowed_states=c['allowed'],
                                      root_distn=c['compound_distn'], P_default=None)
            seen_zero = True
            continue
        lk = ra.mcy.get_likelihood(T_aug, c['root'], n2, node_to_allowed_states=c['allowed'],
                                   root_distn=c['compound_distn'], P_default=None)
        assert np.log(lk) == pytest.approx(want['log_likelihood'], rel=RTOL_LL)
        lk2 = ra.mjp.get_likelihood(c['T'], c['allowed'], c['root'], n2,
                                    root_distn=c['compound_distn'], Q_default=c['Q_compound'])
        assert lk2 == lk
        node_to_pmap = ra.mcy.get_node_to_pmap(T_aug, c['root'], n2,
                                               node_to_allowed_states=c['allowed'])
        np.testing.assert_allclose(node_to_pmap[c['root']], want['root_pmap'], rtol=1e-10,
                                   atol=1e-300)
        node_to_distn = _mc0_dense.get_node_to_distn_esd(
            T_aug, c['root'], node_to_pmap, n2, root_distn=c['compound_distn'])
        d0 = node_to_distn[c['original_root']]
        np.testing.assert_allclose(d0, want['original_root_distn'], rtol=1e-9, atol=1e-18)
        assert d0[:fx['nstates']].sum() == pytest.approx(want['p_reference'], rel=RTOL_LL)
        if 'P_scipy' in want:
            nb = want['P_scipy_node']
            P = T_aug[dict(nx.bfs_predecessors(T_aug, c['root']))[nb]][nb]['P']
            np.testing.assert_allclose(P[want['P_scipy_rows']], want['P_scipy'], rtol=1e-11,
                                       atol=1e-18)
    assert seen_zero
    # batched: the sites that share a rate matrix cannot be one batch here (a matrix per
    # column), so each column is replicated with its leaf sets perturbed: masks of two
    # words per node against the oracle, interpreter and generic kernels
    c = cases[0]
    leaves = [v for v in c['T'] if c['T'].degree(v) == 1]
    rng = np.random.RandomState(5)
    nsites = 300
    dm = np.zeros((nsites, len(leaves), n2))
    for i in range(nsites):
        for k, v in enumerate(leaves):
            cod = rng.randint(fx['nstates']) if rng.uniform() < 0.2 else min(c['allowed'][v])
            dm[i, k, [cod, fx['nstates'] + cod]] = 1.0
    masks = np.zeros((nsites, len(leaves), 2), dtype=np.uint64)
    for sidx in range(n2):
        masks[..., sidx // 64] |= dm[..., sidx].astype(np.uint64) << np.uint64(sidx % 64)
    sites = [dict((v, set(np.nonzero(dm[i, k])[0])) for k, v in enumerate(leaves))
             for i in range(4)]
    np.testing.assert_array_equal(ra.mjp.allowed_states_to_masks(sites, leaves, n2), masks[:4])
    model = ra.device.TreeModel(c['T'], c['root'], n2)
    model.set_rates(Q_default=c['Q_compound'])
    model.set_root_distn(c['compound_distn'])
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(c['T'], c['root'], n2,
                                                            Q_default=c['Q_compound'])
    np.testing.assert_allclose(model.get_transitions()[1:], esd[1:], rtol=1e-10, atol=1e-16)
    oidx = [pre.index(v) for v in leaves]
    wl, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dm, c['compound_distn'])
    assert 0 < wst.sum() < nsites
    for kind, data in (('mask', masks), ('dense', dm)):
        batch = model.upload_sites(leaves, data, kind=kind)
        ll, st = model.log_likelihoods(batch)
        assert batch.kernel_name.startswith('prune_mfma<8,31'), batch.kernel_name
        np.testing.assert_array_equal(st & 1, wst)
        np.testing.assert_allclose(ll[wst == 0], wl[wst == 0], rtol=RTOL_LL)
        assert np.all(np.isneginf(ll[wst != 0]))


@pytest.mark.parametrize('n', [65, 100, 128])
def test_reference_format_passes_above_64_states(ra, n):
    """pset / set / pmap / distn / joint on the reference's arrays at 64 < n <= 128 (one
    128-lane workgroup per site) against the oracle's restatement of the pyfelscore twins."""
    rng = np.random.RandomState(n)
    T, root, obs_nodes, w = _random_case(ra, rng, n, 19, 3, sparse=True)
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    nsites = 3
    mask = (rng.uniform(size=(nsites, len(pre), n)) < 0.5).astype(np.int64)
    mask[:, :, rng.randint(n)] = 1
    want_mask, want_pmap, want_distn, want_joint = [], [], [], []
    for i in range(nsites):
        m = mask[i].copy()
        orc.mcy_esd_get_node_to_pset(idx, ptr, esd, m)
        orc.esd_get_node_to_set(idx, ptr, esd, m)
        pm = orc.mcy_esd_get_node_to_pmap(idx, ptr, esd, m)
        dn = orc.mc0_esd_get_node_to_distn(idx, ptr, esd, w, pm)
        want_mask.append(m)
        want_pmap.append(pm)
        want_distn.append(dn)
        want_joint.append(orc.mc0_esd_get_joint_endpoint_distn(idx, ptr, esd, pm, dn))
    got_mask = mask.copy()
    ra.ctx.node_to_pset(idx, ptr, esd, got_mask)
    ra.ctx.node_to_set(idx, ptr, esd, got_mask)
    np.testing.assert_array_equal(got_mask, np.array(want_mask))
    pmap = np.empty(mask.shape)
    ra.ctx.node_to_pmap(idx, ptr, esd, got_mask, pmap)
    np.testing.assert_allclose(pmap, np.array(want_pmap), rtol=1e-12, atol=1e-300)
    both = mask.copy()
    pmap2 = np.empty(mask.shape)
    ra.ctx.passes(idx, ptr, esd, both, pmap2)
    np.testing.assert_array_equal(both, got_mask)
    np.testing.assert_array_equal(pmap2, pmap)
    distn, status = ra.ctx.node_to_distn(idx, ptr, esd, w, pmap)
    assert not status.any()
    np.testing.assert_allclose(distn, np.array(want_distn), rtol=1e-11, atol=1e-300)
    J = ra.ctx.joint_endpoint_distn(idx, ptr, esd, pmap, distn)
    np.testing.assert_allclose(J, np.array(want_joint), rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize('n', [33, 61, 64, 90, 122])
def test_leaf_state_kernels_gather_columns_bit_identically(ra, n, monkeypatch):
    """Observed STATES at the leaves (type x, the reference's _mcx case) above 32 states: the
    tree-specialised kernel's leaf steps gather columns of P instead of multiplying (jit.hip,
    `sparse`) -- the same numbers bit for bit as the interpreter kernel on the dense expansion,
    for one and two tiles per workgroup, random multifurcating trees, compiled in the
    foreground and in the background; batches it does not cover (an unobserved leaf, an
    observed inner node) take the dense kernels."""
    rng = np.random.RandomState(4000 + n)
    set_option = ra.lib.lib().rt_set_option
    for nnodes, nsites in ((2, 40), (9, 130), (40, 700)):
        T, root, obs_nodes, w = _random_case(ra, rng, n, nnodes, nsites, internal_obs=False)
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        states = rng.randint(0, n, size=(nsites, len(obs_nodes))).astype(np.uint8)
        dense = np.zeros((nsites, len(obs_nodes), n))
        ii, kk = np.indices(states.shape)
        dense[ii, kk, states] = 1.0
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in obs_nodes],
                                              dense, w)
        model = ra.device.TreeModel(T, root, n)
        model.set_transitions(esd)
        model.set_root_distn(w)
        ra.lib.check(set_option(b'jit', 0))
        try:
            ref_ll, ref_st = model.log_likelihoods(model.upload_sites(obs_nodes, states, kind='state'))
        finally:
            ra.lib.check(set_option(b'jit', -1))
        np.testing.assert_array_equal(ref_st & 1, wst)
        np.testing.assert_allclose(ref_ll[wst == 0], want[wst == 0], rtol=RTOL_LL)
        for tiles in (1, 2):
            monkeypatch.setenv('RAOTEH_JIT_TILES', str(tiles))
            ra.lib.check(set_option(b'jit', 1))
            try:
                batch = model.upload_sites(obs_nodes, states, kind='state')
                ll, st = model.log_likelihoods(batch)
                # (above 64 states the generator with fewer tiles may be the one that fits)
                assert 'leaf-states' in batch.kernel_name and \
                    (('T%d' % tiles) in batch.kernel_name or n > 64), batch.kernel_name
                np.testing.assert_array_equal(ll, ref_ll)
                np.testing.assert_array_equal(st, ref_st)
                twin = batch.clone()
                ll2, _ = model.log_likelihoods(twin)
                assert 'leaf-states' in twin.kernel_name
                np.testing.assert_array_equal(ll2, ref_ll)
                tot = model.fetch_totals(batch)
                assert tot[2] == nsites and tot[0] == pytest.approx(ref_ll[wst == 0].sum(), rel=1e-12)
                # dense input of the same batch: the dense kernels, the same numbers
                dll, _ = model.log_likelihoods(model.upload_sites(obs_nodes, dense, kind='dense'))
                np.testing.assert_array_equal(dll, ref_ll)
            finally:
                ra.lib.check(set_option(b'jit', -1))
                monkeypatch.delenv('RAOTEH_JIT_TILES')
        # not covered: an unobserved leaf / an observed inner node -> the dense kernels
        ra.lib.check(set_option(b'jit', 1))
        try:
            gap = states.copy()
            gap[0, 0] = 255
            b1 = model.upload_sites(obs_nodes, gap, kind='state')
            model.log_likelihoods(b1)
            assert 'leaf-states' not in b1.kernel_name and b1.kernel_name.startswith('prune_tree_jit')
            inner = [v for v in T if T.degree(v) > 1 and v != root]
            if inner:
                more = obs_nodes + inner[:1]
                st2 = np.concatenate([states, rng.randint(0, n, size=(nsites, 1)).astype(np.uint8)], axis=1)
                b2 = model.upload_sites(more, st2, kind='state')
                model.log_likelihoods(b2)
                assert 'leaf-states' not in b2.kernel_name
            # switched off (what the benchmark's C4, defined on dense vectors, asks for): the
            # products at the leaves, the same numbers
            ra.ctx.set_option('leaf_state_kernels', 0)
            try:
                b3 = model.upload_sites(obs_nodes, states, kind='state')
            finally:
                ra.ctx.set_option('leaf_state_kernels', None)
            ll3, st3 = model.log_likelihoods(b3)
            assert 'leaf-states' not in b3.kernel_name and b3.kernel_name.startswith('prune_tree_jit')
            b4 = model.upload_sites(obs_nodes, states, kind='state')
            ll4, st4 = model.log_likelihoods(b4)
            assert 'leaf-states' in b4.kernel_name
            np.testing.assert_array_equal(ll3, ll4)
            np.testing.assert_array_equal(st3, st4)
        finally:
            ra.lib.check(set_option(b'jit', -1))
    # in the background: interpreter first, the column-gathering kernel after the switch
    ctx = ra.device.Context(0)
    ctx.set_option('jit_async', 1)
    nsites = max(70000 // n, 700)
    T, root, obs_nodes, w = _random_case(ra, rng, n, 21, nsites, internal_obs=False)
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    states = rng.randint(0, n, size=(nsites, len(obs_nodes))).astype(np.uint8)
    model = ra.device.TreeModel(T, root, n, ctx=ctx)
    model.set_transitions(esd)
    model.set_root_distn(w)
    batch = model.upload_sites(obs_nodes, states, kind='state')
    ll0, _ = model.log_likelihoods(batch)
    batch.wait_for_kernel()
    ll1, _ = model.log_likelihoods(batch)
    assert 'leaf-states' in batch.kernel_name, batch.kernel_name
    np.testing.assert_array_equal(ll0, ll1)
    ctx.close()


@pytest.mark.parametrize('n', [5, 12, 20, 24, 40, 61, 96, 122])
def test_leaf_sets_of_one_or_two_states_are_gathered_columns(ra, n):
    """Allowed sets of one or two states at every leaf (the compound models: a codon in either
    class of the switching model, liwen.py:682) uploaded as masks: the specialised kernels add two
    gathered columns of P instead of multiplying by a 0/1 vector -- bit-identical with the
    interpreter kernel on the same masks and on their dense expansion."""
    set_option = ra.lib.lib().rt_set_option
    rng = np.random.RandomState(2200 + n)
    nsites = 900
    T, root, obs_nodes, w = _random_case(ra, rng, n, 23, nsites, internal_obs=False)
    K = len(obs_nodes)
    a = rng.randint(0, n, size=(nsites, K))
    b = rng.randint(0, n, size=(nsites, K))
    b[rng.uniform(size=b.shape) < 0.25] = -1                    # single-state sets among them
    words = (n + 63) // 64
    masks = np.zeros((nsites, K, words), dtype=np.uint64)
    dense = np.zeros((nsites, K, n))
    ii, kk = np.indices(a.shape)
    for arr in (a, b):
        ok = arr >= 0
        i2, k2, s2 = ii[ok], kk[ok], arr[ok]
        np.bitwise_or.at(masks, (i2, k2, s2 // 64), np.uint64(1) << (s2 % 64).astype(np.uint64))
        dense[i2, k2, s2] = 1.0
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    model = ra.device.TreeModel(T, root, n)
    model.set_transitions(esd)
    model.set_root_distn(w)
    out = {}
    for jit in (0, 1):
        ra.lib.check(set_option(b'jit', jit))
        try:
            bm = model.upload_sites(obs_nodes, masks, kind='mask')
            out[jit] = model.log_likelihoods(bm) + (bm.kernel_name,)
            if jit:                               # a clone carries the leaf words too
                twin = bm.clone()
                np.testing.assert_array_equal(model.log_likelihoods(twin)[0], out[jit][0])
                assert 'leaf-states' in twin.kernel_name
        finally:
            ra.lib.check(set_option(b'jit', -1))
    # (above 32 states the interpreter kernel gathers at the leaves too; the dense upload below
    # runs the products)
    assert 'leaf-states' in out[1][2] and 'jit' in out[1][2] and out[0][2].startswith('prune_mfma'), \
        (out[0][2], out[1][2])
    np.testing.assert_array_equal(out[1][0], out[0][0])
    np.testing.assert_array_equal(out[1][1], out[0][1])
    ra.lib.check(set_option(b'jit', 0))
    try:
        bd = model.upload_sites(obs_nodes, dense, kind='dense')
        lld, std = model.log_likelihoods(bd)
    finally:
        ra.lib.check(set_option(b'jit', -1))
    np.testing.assert_array_equal(out[1][0], lld)
    # the same with one observed state per leaf (n <= 32: the one-wave 4x4x4 kernels read the
    # columns from the blocks they park in LDS; above: test_leaf_state_kernels_...)
    if n <= 32:
        st1 = a.astype(np.uint8)
        res = {}
        for jit in (0, 1):
            ra.lib.check(set_option(b'jit', jit))
            try:
                bs = model.upload_sites(obs_nodes, st1, kind='state')
                res[jit] = model.log_likelihoods(bs) + (bs.kernel_name,)
            finally:
                ra.lib.check(set_option(b'jit', -1))
        assert 'leaf-states' in res[1][2] and '4x4' in res[1][2], res[1][2]
        np.testing.assert_array_equal(res[1][0], res[0][0])
        np.testing.assert_array_equal(res[1][1], res[0][1])
    # a set of three states somewhere: the dense kernels
    m3 = masks.copy()
    m3[0, 0, 0] = np.uint64(7)
    ra.lib.check(set_option(b'jit', 1))
    try:
        b3 = model.upload_sites(obs_nodes, m3, kind='mask')
        model.log_likelihoods(b3)
        assert 'leaf-states' not in b3.kernel_name
    finally:
        ra.lib.check(set_option(b'jit', -1))


def test_deep_caterpillar_and_wide_star(ra):
    rng = np.random.RandomState(11)
    n = 4
    T = nx.Graph()                       # caterpillar: 300 levels deep
    for i in range(300):
        T.add_edge(2 * i, 2 * i + 2, weight=0.05)
        T.add_edge(2 * i, 2 * i + 1, weight=0.08)
    leaves = [2 * i + 1 for i in range(300)] + [600]
    Q, pi = ra.synth.hky85()
    S = nx.Graph()                       # star with 90 leaves
    for i in range(1, 91):
        S.add_edge(0, i, weight=0.01 * i)
    for tree, root, lv in ((T, 0, leaves), (S, 0, list(range(1, 91)))):
        states = rng.randint(0, n, size=(130, len(lv))).astype(np.uint8)
        ll, st = ra.mjp.get_log_likelihoods(tree, root, n, lv, states,
                                            kind='state', root_distn=pi,
                                            Q_default=Q)
        pre, idx, ptr, esd = orc.get_expm_augmented_transitions(
            tree, root, n, Q_default=Q)
        want, wst = orc.batch_log_likelihoods(
            idx, ptr, esd, [pre.index(v) for v in lv],
            ra.synth.one_hot(states, n), pi)
        np.testing.assert_array_equal(st, wst)
        np.testing.assert_allclose(ll, want, rtol=RTOL_LL)


def test_underflow_is_reported_not_hidden(ra):
    # the reference never rescales: a likelihood below the
    # f64 range comes out as zero probability there, and here
    n = 4
    T = nx.Graph()
    for i in range(1, 1400):
        T.add_edge(0, i, weight=1.0)
    Q, pi = ra.synth.jukes_cantor(4)
    states = np.zeros((2, 1399), dtype=np.uint8)
    states[1, ::2] = 1
    ll, st = ra.mjp.get_log_likelihoods(T, 0, n, list(range(1, 1400)), states,
                                        kind='state', root_distn=pi, Q_default=Q)
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, 0, n, Q_default=Q)
    want, wst = orc.batch_log_likelihoods(idx, ptr, esd, list(range(1, 1400)),
                                          ra.synth.one_hot(states, n), pi)
    np.testing.assert_array_equal(st, wst)
    assert wst.all() and np.all(np.isneginf(ll))


def _extended_log_likelihoods(tree, esd, leaf_idx, states, n, root_w):
    """Felsenstein pruning in np.longdouble (x87 extended: exponent range 2^-16445) on the
    device's own transition matrices -- what the f64 recursion would give without underflow."""
    ld = np.longdouble
    nsites = states.shape[0]
    L = [None] * tree.nnodes
    col = dict((v, k) for k, v in enumerate(leaf_idx))
    for v in range(tree.nnodes - 1, -1, -1):
        x = np.ones((nsites, n), dtype=ld)
        if v in col:
            s = states[:, col[v]]
            obs = s != 255
            x[obs] = 0
            x[np.nonzero(obs)[0], s[obs]] = 1
        for c in tree.indices[tree.indptr[v]:tree.indptr[v + 1]]:
            x = x * (L[c] @ esd[c].astype(ld).T)
            L[c] = None
        L[v] = x
    lik = L[0] @ np.asarray(root_w, dtype=ld)
    return np.log(lik).astype(np.float64)


@pytest.mark.parametrize('n,generic', [(2, False), (4, False), (13, False), (20, False),
                                       (61, False), (100, False), (4, True), (20, True)])
def test_opt_in_rescaling_recovers_likelihoods_below_the_f64_range(ra, n, generic):
    """'rescale' (rt_ctx_set_option): the interpreter kernels multiply a site's messages by an
    exact power of two whenever their largest entry falls below 2^-256 and carry the exponent
    per site.  The reference has no rescaling: a 1 024-leaf tree is zero
    probability there and, by default, here (test_underflow_is_reported_not_hidden); with the
    option the log-likelihoods are those of an extended-precision recursion, and a batch that
    never comes near the threshold gets the default kernels' numbers bit for bit."""
    rng = np.random.RandomState(600 + n)
    nleaves = 2048 if n <= 4 else 1024 if n <= 20 else 256
    T, root, leaves = ra.synth.balanced_tree(nleaves, seed=n)
    for a, b in T.edges():
        T[a][b]['weight'] *= 8.0            # long branches: little signal, tiny likelihoods
    Q = rng.uniform(0.1, 1.0, size=(n, n))
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(axis=1))
    pi = rng.dirichlet(np.ones(n))
    nsites = 150
    states = rng.randint(0, n, size=(nsites, nleaves)).astype(np.uint8)
    states[rng.uniform(size=states.shape) < 0.03] = 255
    small = states[:, :8].copy()            # an 8-leaf problem: nowhere near the threshold
    T8, root8, leaves8 = ra.synth.balanced_tree(8, seed=n)
    got = {}
    for rescale in (0, 1):
        ra.ctx.set_option('rescale', rescale)
        ra.ctx.set_option('force_generic', 1 if generic else None)
        try:
            model = ra.device.TreeModel(T, root, n)
            model.set_root_distn(pi)
            model.set_rates(Q_default=Q)
            batch = model.upload_sites(leaves, states, kind='state')
            ll, st = model.log_likelihoods(batch)
            tot = model.fetch_totals(batch)
            m8 = ra.device.TreeModel(T8, root8, n)
            m8.set_root_distn(pi)
            m8.set_rates(Q_default=Q)
            b8 = m8.upload_sites(leaves8, small, kind='state')
            got[rescale] = (ll, st, batch.kernel_name, tot, m8.log_likelihoods(b8)[0],
                            b8.kernel_name)
            if rescale:
                esd = model.get_transitions()
                want = _extended_log_likelihoods(
                    model.tree, esd, [model.tree.node_to_index[v] for v in leaves], states, n, pi)
            b8.close(); m8.close(); batch.close(); model.close()
        finally:
            ra.ctx.set_option('rescale', None)
            ra.ctx.set_option('force_generic', None)
    # default: the f64 recursion underflows, every site is reported as zero probability
    assert np.all(np.isneginf(got[0][0])) and np.all(got[0][1] & 1)
    ll, st, name, tot, ll8, name8 = got[1]
    assert 'rescale' in name or name.startswith('prune_generic'), name
    assert np.all(st == 0)
    assert want.max() < -745.0              # log of the smallest subnormal: all below the range
    np.testing.assert_allclose(ll, want, rtol=RTOL_LL)
    assert tot[1] == 0 and abs(tot[0] - want.sum()) <= 1e-10 * abs(want.sum())
    # exact powers of two: a batch that never rescales is untouched
    assert 'jit' not in name8
    np.testing.assert_array_equal(ll8, got[0][4])


def test_expected_history_statistics(ra):
    """_mjp_dense.get_expected_history_statistics (:410-539) and its sparse twin
    (_mjp.py:431-595) on the device -- one Frechet block exponential per edge by
    the adjoint identity -- against the reference's own outputs, the Jukes-Cantor
    closed form of tests/test_mjp.py:166-237, and the oracle's restatement (which
    calls scipy's expm_frechet once per direction, as the reference does)."""
    from raoteh_amd import _mjp, _mjp_dense
    worst = 0.0
    for label, T, allowed, root, n, distn, Q, want in expectation_cases():
        dwell, init, trans = _mjp_dense.get_expected_history_statistics(
            T, allowed, root, n, root_distn=distn, Q_default=Q)
        got_dwell = np.array([dwell[c] for c in range(n)])
        got_trans = np.zeros((n, n))
        for c, d, dat in trans.edges(data=True):
            got_trans[c, d] = dat['weight']
        odwell, oinit, otrans = orc.mjp_dense_get_expected_history_statistics(
            T, allowed, root, n, root_distn=distn, Q_default=Q)
        off = ~np.eye(n, dtype=bool)
        scale = np.abs(odwell).max()
        np.testing.assert_allclose(got_dwell, want['dwell'], rtol=1e-10, atol=1e-13 * scale,
                                   err_msg=label)
        np.testing.assert_allclose(init, want['init'], rtol=1e-12, atol=1e-15, err_msg=label)
        np.testing.assert_allclose(got_trans[off], np.array(want['trans'])[off], rtol=1e-10,
                                   atol=1e-13 * scale, err_msg=label)
        # the diagonal entries (dense reference only) against the oracle
        np.testing.assert_allclose(got_trans, otrans, rtol=1e-10, atol=1e-13 * scale,
                                   err_msg=label)
        np.testing.assert_allclose(got_dwell, odwell, rtol=1e-10, atol=1e-13 * scale,
                                   err_msg=label)
        if 'closed_form_dwell' in want:
            np.testing.assert_allclose(got_dwell, want['closed_form_dwell'], rtol=1e-10,
                                       atol=1e-13 * scale, err_msg=label)
        assert got_dwell.sum() == pytest.approx(
            sum(d['weight'] for _, _, d in T.edges(data=True)), rel=1e-10)
        worst = max(worst, float(np.max(np.abs(got_dwell - odwell) / scale)))
    assert worst < 1e-11

    # sparse API: rate matrices as loop-free digraphs over arbitrary labels
    labels = [7, 3, 40, 11, 25, 2]
    for label, T, allowed, root, n, distn, Q, want in expectation_cases()[48:]:
        lab = sorted(labels[:n])

        def to_digraph(D):
            G = nx.DiGraph()
            G.add_nodes_from(lab)
            for i in range(n):
                for j in range(n):
                    if i != j and D[i, j]:
                        G.add_edge(lab[i], lab[j], weight=float(D[i, j]))
            return G
        graphs = {}
        Ts = nx.Graph()
        for a, b, d in T.edges(data=True):
            Ts.add_edge(a, b, weight=d['weight'])
            if 'Q' in d:
                graphs.setdefault(id(d['Q']), to_digraph(d['Q']))
                Ts[a][b]['Q'] = graphs[id(d['Q'])]
        sdwell, sinit, strans = _mjp.get_expected_history_statistics(
            Ts, dict((v, set(lab[s] for s in ss)) for v, ss in allowed.items()), root,
            root_distn=dict((lab[i], float(p)) for i, p in enumerate(distn)),
            Q_default=to_digraph(Q))
        np.testing.assert_allclose([sdwell[lab[c]] for c in range(n)], want['dwell'],
                                   rtol=1e-10, err_msg=label)
        np.testing.assert_allclose([sinit.get(lab[c], 0.0) for c in range(n)], want['init'],
                                   rtol=1e-12, atol=1e-15, err_msg=label)
        wt = np.array(want['trans'])
        for c in range(n):
            for d in range(n):
                if c != d and wt[c, d]:
                    assert strans[lab[c]][lab[d]]['weight'] == pytest.approx(wt[c, d], rel=1e-10)
        assert strans.number_of_edges() == int(np.count_nonzero(wt))


def test_tolerance_process_closed_forms(ra):
    """The 3-state tolerance process entry points of pyfelscore (_linalg.py:41-69,
    107-118; _tmjp_dense.py:339): the contract their call sites state is equality with
    scipy.linalg.expm / expm_frechet (tests/test_expm.py:20-42 does that for the
    blocks), in all three regimes of the closed forms."""
    import scipy.linalg
    pyf = ra.pyf
    for a, w, r in ((0.7, 1.3, 0.4), (0.7, 0.0, 0.4), (0.9, 0.0, 0.9), (2.0, 5.0, 0.0)):
        Q = np.array([[-a, a, 0.0], [w, -(w + r), r], [0.0, 0.0, 0.0]])
        for t in (2.0 ** -5, 0.3, 1.0, 8.0):
            Pw = scipy.linalg.expm(t * Q)
            blk = pyf.get_mmpp_block(a, w, r, t) if w else pyf.get_mmpp_block_zero_off_rate(a, r, t)
            np.testing.assert_allclose(blk, Pw[:2, :2], rtol=1e-11, atol=1e-15)
            for ci in range(3):
                for di in range(3):
                    C = np.zeros((3, 3))
                    C[ci, di] = 1.0
                    L = scipy.linalg.expm_frechet(t * Q, t * C, compute_expm=False)
                    for ai in range(3):
                        for bi in range(3):
                            if w:
                                got = pyf.get_mmpp_frechet_all_positive(a, w, r, t, ai, bi, ci, di)
                            elif a != r:
                                got = pyf.get_mmpp_frechet_diagonalizable_w_zero(
                                    a, r, t, ai, bi, ci, di)
                            else:
                                got = pyf.get_mmpp_frechet_defective_w_zero(a, t, ai, bi, ci, di)
                            assert got == pytest.approx(L[ai, bi], rel=1e-9, abs=1e-14)
            # one edge of the tolerance expectations against the general formula
            rng = np.random.RandomState(3)
            J = rng.uniform(size=(3, 3)) * (Pw > 0)
            J /= J.sum()
            dwell, trans = np.zeros(2), np.zeros((2, 2))
            absorb = pyf.get_tolerance_expectations(t, Q, Pw, J, dwell, trans)
            live = J != 0
            ratio = np.zeros((3, 3))
            ratio[live] = J[live] / Pw[live]

            def contract(c, d):
                C = np.zeros((3, 3))
                C[c, d] = 1.0
                return float(np.sum(ratio * scipy.linalg.expm_frechet(
                    t * Q, t * C, compute_expm=False)))
            assert dwell[0] == pytest.approx(contract(0, 0), rel=1e-9, abs=1e-14)
            assert dwell[1] == pytest.approx(contract(1, 1), rel=1e-9, abs=1e-14)
            assert trans[0, 1] == pytest.approx(a * contract(0, 1), rel=1e-9, abs=1e-14)
            assert trans[1, 0] == pytest.approx(w * contract(1, 0), rel=1e-9, abs=1e-14)
            assert absorb == pytest.approx(r * contract(1, 2), rel=1e-9, abs=1e-14)
            assert trans[0, 0] == 0 and trans[1, 1] == 0


def test_expected_history_statistics_codon_model(ra):
    """61 states: the Frechet blocks have order 122 (the global-scratch Taylor kernel);
    block assembly and the contraction over the edges run on the device.  A 9-node tree,
    a sample of the reference's per-direction numbers from the oracle, every entry
    against the same one-derivative-per-edge formula evaluated with scipy on the host,
    and the invariant sum of dwell times = tree length."""
    import scipy.linalg
    from raoteh_amd import _mjp_dense
    Q, pi = ra.synth.mg94()
    n = 61
    rng = np.random.RandomState(61)
    T = nx.Graph()
    for a, b in ((0, 1), (0, 2), (1, 3), (1, 4), (2, 5), (2, 6), (6, 7), (6, 8)):
        T.add_edge(a, b, weight=float(rng.uniform(0.05, 0.4)))
    leaves = [3, 4, 5, 7, 8]
    allowed = dict((v, set(range(n))) for v in T)
    for v in leaves:
        allowed[v] = {int(rng.randint(n))}
    allowed[5] = set(int(x) for x in rng.choice(n, 3, replace=False))   # an ambiguous leaf
    dwell, init, trans = _mjp_dense.get_expected_history_statistics(
        T, allowed, 0, n, root_distn=pi, Q_default=Q)
    dwell = np.array([dwell[c] for c in range(n)])
    total = sum(d['weight'] for _, _, d in T.edges(data=True))
    assert dwell.sum() == pytest.approx(total, rel=1e-10)
    assert np.all(dwell >= 0) and init.sum() == pytest.approx(1.0, rel=1e-12)
    # (1) a sample of directions the reference's way (one expm_frechet per direction)
    off = np.argwhere((Q != 0) & ~np.eye(n, dtype=bool))
    pairs = [tuple(off[k]) for k in rng.choice(len(off), 10, replace=False)]
    pairs += [(int(c), int(c)) for c in rng.choice(n, 6, replace=False)]
    want, want_init = orc.mjp_dense_expected_history_statistics_entries(
        T, allowed, 0, n, pairs, root_distn=pi, Q_default=Q)
    np.testing.assert_allclose(init, want_init, rtol=1e-11, atol=1e-16)
    for (c, d), v in want.items():
        got = dwell[c] if c == d else trans[c][d]['weight']
        assert got == pytest.approx(v, rel=1e-9, abs=1e-15), (c, d)
    # (2) every entry: the batch form against scipy's expm of the same blocks
    states = np.full((3, len(leaves)), 255, dtype=np.int64)
    states[0] = [sorted(allowed[v])[0] for v in leaves]
    states[1] = rng.randint(n, size=len(leaves))
    states[2, :2] = rng.randint(n, size=2)
    w = np.array([2.0, 1.0, 3.0])
    bd, bi, bt = _mjp_dense.get_expected_history_statistics_batch(
        T, 0, n, root_distn=pi, Q_default=Q, weights=w, obs_nodes=leaves, data=states,
        kind='state')
    assert bd.sum() == pytest.approx(total * w.sum(), rel=1e-10)
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, 0, n, Q_default=Q)
    ref_d, ref_t = np.zeros(n), np.zeros((n, n))
    for k in range(3):
        al = dict((v, set(range(n))) for v in T)
        for v, s in zip(leaves, states[k]):
            if s != 255:
                al[v] = {int(s)}
        mask = orc.define_state_mask(al, pre, n)
        _, pmap = orc.esd_get_node_to_pmap(idx, ptr, esd, mask)
        distn = orc.mc0_esd_get_node_to_distn(idx, ptr, esd, pi, pmap)
        J = orc.mc0_esd_get_joint_endpoint_distn(idx, ptr, esd, pmap, distn)
        for i, v in enumerate(pre):
            if i == 0:
                continue
            pa = [u for u in T[v] if pre.index(u) < i][0]
            t = T[pa][v]['weight']
            live = J[i] != 0
            W = np.zeros((n, n))
            W[live] = J[i][live] / esd[i][live]
            B = np.zeros((2 * n, 2 * n))
            B[:n, :n] = B[n:, n:] = t * Q.T
            B[:n, n:] = W
            M = scipy.linalg.expm(B)[:n, n:]
            ref_d += w[k] * t * np.diag(M)
            ref_t += w[k] * np.where(Q != 0, t * Q * M, 0.0)
    np.testing.assert_allclose(bd, ref_d, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(bt, ref_t, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize('n', [2, 4, 5, 8, 9, 20, 31, 40, 61, 64])
def test_resident_expectation_step_matches_the_reference_shaped_path(ra, n):
    """rt_expect_step on an uploaded batch (nothing marshalled per call) against
    get_expected_history_statistics_batch -- the reference-shaped path that is pinned to the
    reference's own numbers by tests/golden/expectations.json -- for dense / state / mask
    batches, per-edge rate matrices, site weights, a batch that runs a tree-specialised
    pruning kernel, and repeated calls with new rates (an EM loop)."""
    from raoteh_amd import _mjp_dense
    rng = np.random.RandomState(300 + n)
    T, root, leaves = ra.synth.random_tree(23, seed=n, max_children=3)

    def random_rates():
        Q = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) < 0.5)
        np.fill_diagonal(Q, 0.0)
        Q[np.arange(n), (np.arange(n) + 1) % n] += 0.2
        Q -= np.diag(Q.sum(axis=1))
        return Q / np.abs(np.diag(Q)).mean()

    Q0 = random_rates()
    some_edge = list(nx.bfs_edges(T, root))[3]
    T[some_edge[0]][some_edge[1]]['Q'] = random_rates()        # one edge with its own matrix
    pi = rng.dirichlet(np.ones(n))
    nsites = 150
    states = rng.randint(n, size=(nsites, len(leaves)))
    states[rng.uniform(size=states.shape) < 0.2] = 255
    w = rng.randint(1, 4, size=nsites).astype(float)
    want = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, root_distn=pi, Q_default=Q0, weights=w, obs_nodes=leaves, data=states,
        kind='state')
    model = ra.device.TreeModel(T, root, n)
    model.set_rates(Q_default=Q0)
    model.set_root_distn(pi)
    total = sum(d['weight'] for _, _, d in T.edges(data=True))
    dense = np.ones((nsites, len(leaves), n))
    obs = states != 255
    dense[obs] = 0.0
    ii, kk = np.nonzero(obs)
    dense[ii, kk, states[ii, kk]] = 1.0
    masks = np.zeros((nsites, len(leaves)), dtype=np.uint64)
    for sidx in range(n):
        masks |= dense[..., sidx].astype(np.uint64) << np.uint64(sidx)
    for kind, data in (('state', states.astype(np.uint8)), ('dense', dense), ('mask', masks)):
        for jit in (0, 1):
            ra.lib.check(ra.lib.lib().rt_set_option(b'jit', jit))
            try:
                batch = model.upload_sites(leaves, data, kind=kind).set_weights(w)
            finally:
                ra.lib.check(ra.lib.lib().rt_set_option(b'jit', -1))
            ll, _ = model.log_likelihoods(batch)
            assert batch.kernel_name.startswith('prune_tree_jit' if jit else
                                                ('prune_lane' if n <= 4 else 'prune_mfma'))
            dwell, rootp, trans, status = model.expected_history_statistics(
                batch, return_status=True)
            assert not status.any()
            np.testing.assert_allclose(dwell, want[0], rtol=1e-10, atol=1e-14)
            np.testing.assert_allclose(rootp, want[1], rtol=1e-10, atol=1e-14)
            np.testing.assert_allclose(trans, want[2], rtol=1e-10, atol=1e-13)
            assert dwell.sum() == pytest.approx(total * w.sum(), rel=1e-10)
            assert rootp.sum() == pytest.approx(w.sum(), rel=1e-11)
            # the likelihood path of the same batch is untouched by the expectation step
            ll2, _ = model.log_likelihoods(batch)
            np.testing.assert_array_equal(ll, ll2)
    # new rates, same resident batch: what an EM iteration does
    Q1 = random_rates()
    model.set_rates(Q_default=Q1)
    got = model.expected_history_statistics(batch, recompute_transitions=False)
    want1 = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, root_distn=pi, Q_default=Q1, weights=w, obs_nodes=leaves, data=states,
        kind='state')
    for a, b in zip(got, want1):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-13)
    batch.set_weights(None)
    got = model.expected_history_statistics(batch)
    assert got[1].sum() == pytest.approx(nsites, rel=1e-11)
    # a site the data rule out: flagged, as the reference raises NumericalZeroProb
    bad = np.array([[0.0] * n] + [[1.0] * n] * (len(leaves) - 1))[None]
    b2 = model.upload_sites(leaves, bad, kind='dense')
    out = model.expected_history_statistics(b2, return_status=True)
    assert out[3][0] == 2
    # the generic fallback layout is not resident in a form the passes read
    ra.lib.check(ra.lib.lib().rt_set_option(b'force_generic', 1))
    try:
        bg = model.upload_sites(leaves, dense, kind='dense')
    finally:
        ra.lib.check(ra.lib.lib().rt_set_option(b'force_generic', 0))
    with pytest.raises(ra.lib.RaotehHipError):
        model.expected_history_statistics(bg)


def test_expected_history_statistics_batch(ra):
    """The batched form: the site sum of the reference's per-site statistics from
    ONE Frechet block exponential per edge, with site-pattern weights."""
    from raoteh_amd import _mjp_dense
    rng = np.random.RandomState(8)
    cfg = ra.synth.make_config('c2', nsites=12)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    sites = []
    for row in cfg['leaf_states']:
        d = dict((leaf, {int(s)}) for leaf, s in zip(cfg['leaves'], row))
        sites.append(d)
    sites[3][cfg['leaves'][5]] = {0, 2}              # an ambiguous leaf
    w = rng.randint(1, 4, size=len(sites)).astype(float)
    dwell, init, trans = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, sites, root_distn=cfg['root_distn'], Q_default=cfg['Q_default'],
        weights=w)
    want_d, want_i, want_t = np.zeros(n), np.zeros(n), np.zeros((n, n))
    full = [dict((v, set(range(n))) for v in T) for _ in sites]
    for f, s in zip(full, sites):
        f.update(s)
    for k, f in enumerate(full):
        od, oi, ot = orc.mjp_dense_get_expected_history_statistics(
            T, f, root, n, root_distn=cfg['root_distn'], Q_default=cfg['Q_default'])
        want_d += w[k] * od
        want_i += w[k] * oi
        want_t += w[k] * ot
    np.testing.assert_allclose(dwell, want_d, rtol=1e-10)
    np.testing.assert_allclose(init, want_i, rtol=1e-12)
    np.testing.assert_allclose(trans, want_t, rtol=1e-10, atol=1e-12)
    total = sum(d['weight'] for _, _, d in T.edges(data=True))
    assert dwell.sum() == pytest.approx(total * w.sum(), rel=1e-10)
    # the array form of the same batch (allowed-set bit masks per leaf)
    masks = (1 << cfg['leaf_states'].astype(np.int64))
    masks[3, 5] = 0b0101
    d2, i2, t2 = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, root_distn=cfg['root_distn'], Q_default=cfg['Q_default'], weights=w,
        obs_nodes=cfg['leaves'], data=masks, kind='mask')
    np.testing.assert_array_equal(d2, dwell)
    np.testing.assert_array_equal(i2, init)
    np.testing.assert_array_equal(t2, trans)
    # more sites than site chunks (the per-edge sums are taken by 64 chunks of sites):
    # the device sums against J / P assembled on the host from the joint endpoint
    # distributions of the reference-format pass (itself checked against the oracle)
    from raoteh_amd._tree import TreeArrays
    def random_config(n, nleaves, nb):
        rT, rroot, rleaves = ra.synth.balanced_tree(nleaves, seed=3)
        R = rng.exponential(size=(n, n))
        np.fill_diagonal(R, 0.0)
        return dict(T=rT, root=rroot, nstates=n, leaves=rleaves, obs_kind='state',
                    leaf_states=rng.randint(n, size=(nb, nleaves)),
                    root_distn=np.full(n, 1.0 / n), Q_default=R - np.diag(R.sum(axis=1)))
    for name, nb in (('c2', 333), ('c5', 70), ('c1', 90), (9, 150), (11, 67), (12, 66)):
        big = (ra.synth.make_config(name, nsites=nb) if isinstance(name, str)
               else random_config(name, 16, nb))
        bT, broot, bn = big['T'], big['root'], big['nstates']
        if big['obs_kind'] == 'state':
            bmask = 1 << big['leaf_states'].astype(np.int64)
            bmask[::7, 2] = 0b1010
        else:
            table = np.array([sum(1 << x for x in ss) for ss in big['leaf_allowed']],
                             dtype=np.int64)
            bmask = table[big['leaf_states']]
        bw = rng.uniform(0.5, 2.0, size=nb)
        T_aug = _mjp_dense.get_expm_augmented_tree(bT, broot, Q_default=big.get('Q_default'))
        ta = TreeArrays(T_aug, broot)
        esd = ta.esd_transitions(bn)
        m3 = np.ones((nb, ta.nnodes, bn), dtype=np.int64)
        cols = [ta.node_to_index[v] for v in big['leaves']]
        m3[:, cols, :] = (bmask[:, :, None] >> np.arange(bn)) & 1
        W, rp, st = ra.ctx.expectation_weights(ta.indices, ta.indptr, esd, big['root_distn'],
                                               m3.copy(), site_weights=bw)
        assert not st.any()
        W2, rp2, st2 = ra.ctx.expectation_weights_obs(
            ta.indices, ta.indptr, esd, big['root_distn'], cols, bmask, 'mask', site_weights=bw)
        if bn <= 8:          # both forms go through the same fused kernel
            np.testing.assert_array_equal(W2, W)
            np.testing.assert_array_equal(rp2, rp)
        else:                # compact observations: the matrix-pipe passes; full masks: per-pass
            np.testing.assert_allclose(W2, W, rtol=1e-11, atol=1e-14 * np.abs(W).max())
            np.testing.assert_allclose(rp2, rp, rtol=1e-12)
        pm = np.empty(m3.shape)
        ra.ctx.passes(ta.indices, ta.indptr, esd, m3.copy(), pm)
        dn, _ = ra.ctx.node_to_distn(ta.indices, ta.indptr, esd, big['root_distn'], pm)
        J = ra.ctx.joint_endpoint_distn(ta.indices, ta.indptr, esd, pm, dn)
        for i in range(1, ta.nnodes):
            ratio = np.where(J[:, i] != 0, J[:, i] / np.where(esd[i] != 0, esd[i], 1.0), 0.0)
            want = np.tensordot(bw, ratio, axes=(0, 0))
            np.testing.assert_allclose(W[i], want, rtol=1e-12, atol=1e-13 * np.abs(want).max(),
                                       err_msg='%s node %d' % (name, i))
        np.testing.assert_allclose(rp, np.tensordot(bw, dn[:, 0], axes=(0, 0)), rtol=1e-13)
        assert not W[0].any()
    # an infeasible site: the reference's normaliser raises NumericalZeroProb
    # (_util.py:164-165, reached from _mc0_dense.get_node_to_distn)
    bad = masks.copy()
    bad[7, 0] = 0
    with pytest.raises(ra.pkg.NumericalZeroProb):
        _mjp_dense.get_expected_history_statistics_batch(
            T, root, n, root_distn=cfg['root_distn'], Q_default=cfg['Q_default'],
            obs_nodes=cfg['leaves'], data=bad, kind='mask')
    with pytest.raises(ra.pkg.NumericalZeroProb):
        impossible = dict((v, set(range(n))) for v in T)
        impossible[cfg['leaves'][0]] = set()
        _mjp_dense.get_expected_history_statistics(
            T, impossible, root, n, root_distn=cfg['root_distn'], Q_default=cfg['Q_default'])
    states = cfg['leaf_states'].copy()
    d3, _, _ = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, root_distn=cfg['root_distn'], Q_default=cfg['Q_default'],
        obs_nodes=cfg['leaves'], data=states, kind='state')
    sites[3][cfg['leaves'][5]] = {int(states[3, 5])}
    d4, _, _ = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, sites, root_distn=cfg['root_distn'], Q_default=cfg['Q_default'])
    np.testing.assert_array_equal(d3, d4)


def _random_csr(rng, nnodes, shape):
    """children lists of a random tree with parent < child labels."""
    parent = np.zeros(nnodes, dtype=np.int64)
    for v in range(1, nnodes):
        if shape == 'caterpillar':
            parent[v] = v - 1 if v % 2 else max(v - 2, 0)     # spine of even labels, one leaf each
        elif shape == 'star':
            half = max(2, nnodes // 2)
            parent[v] = 0 if v < half else rng.randint(1, half)
        else:
            parent[v] = rng.randint(max(0, v - 6), v)
    kids = [[] for _ in range(nnodes)]
    for v in range(1, nnodes):
        kids[parent[v]].append(v)
    indices = np.array([c for k in kids for c in k], dtype=np.int64)
    indptr = np.cumsum([0] + [len(k) for k in kids]).astype(np.int64)
    return indices, indptr, parent


@pytest.mark.parametrize('shape', ['random', 'caterpillar', 'star'])
def test_expectation_weights_kernels_agree_on_ragged_trees(ra, shape, monkeypatch):
    """The three device forms of rt_mjp_esd_expectation_weights -- the LDS stack
    program (heaviest child first), the global-memory lane kernel and the per-pass
    kernels -- on multifurcating, caterpillar and star trees, sparse transition
    supports, restricted internal nodes and a site the data exclude; and the
    per-pass form is the one checked against the oracle above."""
    rng = np.random.RandomState({'random': 1, 'caterpillar': 2, 'star': 3}[shape])
    for n, nnodes, nb in ((2, 9, 70), (3, 40, 130), (4, 127, 200), (5, 33, 64), (7, 21, 65),
                          (8, 60, 129), (4, 1, 5), (4, 2, 64)):
        indices, indptr, parent = _random_csr(rng, nnodes, shape)
        esd = rng.uniform(0.05, 1.0, size=(nnodes, n, n))
        esd[rng.uniform(size=esd.shape) < 0.2] = 0.0
        esd[:, np.arange(n), np.arange(n)] += 0.5
        esd /= esd.sum(axis=2, keepdims=True)
        esd[0] = 0.0
        mask = np.ones((nb, nnodes, n), dtype=np.int64)
        leaves = np.setdiff1d(np.arange(nnodes), parent[1:]) if nnodes > 1 else np.array([0])
        st = rng.randint(n, size=(nb, len(leaves)))
        mask[:, leaves, :] = 0
        mask[np.arange(nb)[:, None], leaves[None, :], st] = 1
        mask[::5, leaves[0], :] |= rng.randint(2, size=n)          # ambiguous leaves
        if nnodes > 3:
            mask[::3, 1, : n // 2] = 0                                 # a restricted internal node
        if nb > 9:
            mask[9, leaves[-1], :] = 0                                 # an impossible site
        root_distn = rng.dirichlet(np.ones(n))
        w = rng.uniform(0.5, 2.0, size=nb)
        got = {}
        for name, env in (('lds', None), ('global', 'RAOTEH_EXPECT_GLOBAL'),
                          ('legacy', 'RAOTEH_EXPECT_LEGACY')):
            if env:
                monkeypatch.setenv(env, '1')
            got[name] = ra.ctx.expectation_weights(indices, indptr, esd, root_distn, mask.copy(),
                                                   site_weights=w)
            if env:
                monkeypatch.delenv(env)
        W, rp, status = got['legacy']
        assert status[9] == 2 if nb > 9 else not status.any()
        ok = status == 0
        assert ok.any()
        for name in ('lds', 'global'):
            W2, rp2, st2 = got[name]
            np.testing.assert_array_equal(st2, status, err_msg='%s n=%d' % (name, n))
            scale = np.abs(W).max() or 1.0
            np.testing.assert_allclose(W2, W, rtol=1e-11, atol=1e-13 * scale,
                                       err_msg='%s n=%d nnodes=%d' % (name, n, nnodes))
            np.testing.assert_allclose(rp2, rp, rtol=1e-12, atol=1e-300)
            assert not W2[0].any()


@pytest.mark.parametrize('nsites', [100000, 400001, 3000])
def test_relaunch_and_clone_give_the_same_totals(ra, nsites):
    """Totals are bitwise reproducible launch after launch, and a clone used the
    moment rt_sites_clone returns holds the whole batch (its device-to-device
    copy is ordered on the library's stream -- a null-stream copy was not, and a
    launch right behind it read a partly copied batch)."""
    cfg = ra.synth.make_config('c2', nsites=nsites)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    dense = ra.synth.leaf_likelihoods(cfg)
    dense[::97, 3, :] = 0.0              # some structurally impossible sites
    model = ra.device.TreeModel(T, root, n)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    set_option = ra.lib.lib().rt_set_option
    got = {}
    try:
        for jit in (0, 1):
            ra.lib.check(set_option(b'jit', jit))
            batch = model.upload_sites(cfg['leaves'], dense, kind='dense')
            ll, st = model.log_likelihoods(batch)
            assert ra.ctx.kernel_time(1)[2].startswith('prune_tree_jit' if jit else 'prune_lane')
            tots = [model.fetch_totals(batch)]
            for _ in range(6):
                model.step(batch)
                tots.append(model.fetch_totals(batch))
            for t in tots[1:]:
                np.testing.assert_array_equal(t, tots[0])
            twin = batch.clone()
            ll2, st2 = model.log_likelihoods(twin)
            np.testing.assert_array_equal(model.fetch_totals(twin), tots[0])
            np.testing.assert_array_equal(ll2, ll)
            got[jit] = (ll, st, tots[0])
    finally:
        ra.lib.check(set_option(b'jit', -1))
    np.testing.assert_array_equal(got[0][0], got[1][0])
    np.testing.assert_array_equal(got[0][1], got[1][1])
    ll, st, tot = got[1]
    nz = int(((st & 1) != 0).sum())
    assert nz == len(range(0, nsites, 97)) and tot[1] == nz and tot[2] == nsites
    assert tot[0] == pytest.approx(ll[(st & 1) == 0].sum(), rel=1e-12)
    assert got[0][2][0] == pytest.approx(tot[0], rel=1e-13) and got[0][2][1] == nz


# ---------------------------------------------------------------------------
# full BASELINE sizes: oracle on everything it can finish in seconds +
# size-independent properties
# ---------------------------------------------------------------------------

@pytest.mark.parametrize('name,nsites', [('c2', 100000), ('c3', 10000),
                                         ('c5', 50000)])
def test_full_size_configs(ra, name, nsites):
    cfg = ra.synth.make_config(name, nsites=nsites)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    leaves, distn = cfg['leaves'], cfg['root_distn']
    dense = ra.synth.leaf_likelihoods(cfg)
    model = ra.device.TreeModel(T, root, n)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(distn)
    batch = model.upload_sites(leaves, dense, kind='dense')
    ll, st = model.log_likelihoods(batch)
    tot = model.fetch_totals(batch)
    assert not st.any() and np.isfinite(ll).all()
    # oracle (scipy expm + numpy pruning) on the full batch
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(
        T, root, n, Q_default=cfg['Q_default'])
    oidx = [pre.index(v) for v in leaves]
    want, _ = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense, distn)
    np.testing.assert_allclose(ll, want, rtol=RTOL_LL)
    assert tot[0] == pytest.approx(want.sum(), rel=RTOL_LL)
    # properties: the batch sum is the sum of the per-site values; identical
    # columns give identical values; permuting sites permutes the output
    assert tot[0] == pytest.approx(ll.sum(), rel=1e-12)
    perm = np.random.RandomState(0).permutation(nsites)
    ll2, _ = model.log_likelihoods(model.upload_sites(leaves, dense[perm]))
    np.testing.assert_array_equal(ll2, ll[perm])
    # the golden sites (values computed by the reference) appended to a slice
    # of this batch come out the same as in their own small batch
    fx = load_golden('config_' + name)
    _, _, _, _, _, gsites = config_from_golden(fx)
    gd = np.zeros((len(gsites), len(leaves), n))
    for i, d in enumerate(gsites):
        for k, v in enumerate(fx['leaves']):
            gd[i, k, sorted(d[v])] = 1.0
    mixed = np.concatenate([dense[:1000], gd])
    ll4, _ = model.log_likelihoods(model.upload_sites(leaves, mixed))
    np.testing.assert_allclose(ll4[1000:], fx['log_likelihoods'], rtol=RTOL_LL)
    np.testing.assert_array_equal(ll4[:1000], ll[:1000])
    # re-running the per-edge expm from the resident rates changes nothing
    model.recompute_transitions()
    ll3, _ = model.log_likelihoods(batch)
    np.testing.assert_array_equal(ll3, ll)


def test_config4_rank0_shard_of_eight(ra):
    """Config 4 = the one million-site codon batch (seed 3) sharded over 8 GPUs
    (examples/p53/p53.py:88-100 is the sum being sharded): rank 0 of 8 owns sites
    dist.shard_range(1 000 000, 0, 8) = [0, 125 000).  The whole shard runs on the one
    GPU here: the oracle on a 2 000-site sample, and size-independent properties on all
    of it."""
    from raoteh_amd.dist import shard_range
    lo, hi = shard_range(ra.synth.C4_NSITES, 0, 8)
    assert (lo, hi) == (0, 125000)
    cfg = ra.synth.make_config('c4', site_range=(lo, hi))
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    leaves, distn = cfg['leaves'], cfg['root_distn']
    states = cfg['leaf_states']
    assert states.shape == (125000, 64) and n == 61
    # the batch is defined chunk by chunk: any sub-range gives the same sites
    sub = ra.synth.make_config('c4', site_range=(31000, 31500))['leaf_states']
    np.testing.assert_array_equal(sub, states[31000:31500])
    model = ra.device.TreeModel(T, root, n)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(distn)
    # uint8 states cross PCIe; the resident layout is the dense f64 one (3.9 GB)
    batch = model.upload_sites(leaves, states, kind='state')
    assert batch.device_bytes >= 125000 * 64 * 61 * 8
    ll, st = model.log_likelihoods(batch)
    tot = model.fetch_totals(batch)
    assert batch.kernel_name.startswith('prune_tree_jit_mfma<61')
    assert not st.any() and np.isfinite(ll).all() and tot[1] == 0 and tot[2] == 125000
    # the oracle on a bounded sample
    pick = np.sort(np.random.RandomState(8).choice(125000, 2000, replace=False))
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(
        T, root, n, Q_default=cfg['Q_default'])
    oidx = [pre.index(v) for v in leaves]
    want, wst = orc.batch_log_likelihoods(
        idx, ptr, esd, oidx, ra.synth.one_hot(states[pick], n), distn)
    assert not wst.any()
    np.testing.assert_allclose(ll[pick], want, rtol=RTOL_LL)
    # checksum of checksums: the device total is the sum of the per-site values
    assert tot[0] == pytest.approx(ll.sum(), rel=1e-12)
    # shards add: two half shards, each in its own batch (other tile positions, the
    # second through a dense upload), give the same per-site values bit for bit
    a = model.upload_sites(leaves, states[:62500], kind='state')
    lla, _ = model.log_likelihoods(a)
    ta = model.fetch_totals(a)
    a.close()
    b = model.upload_sites(leaves, ra.synth.one_hot(states[62500:70500], n), kind='dense')
    llb, _ = model.log_likelihoods(b)
    b.close()
    np.testing.assert_array_equal(lla, ll[:62500])
    np.testing.assert_array_equal(llb, ll[62500:70500])
    assert ta[0] + ll[62500:].sum() == pytest.approx(tot[0], rel=1e-12)
    # one step of the repeated-evaluation loop (expm + prune, reduce deferred) leaves
    # every number where it was
    model.step(batch)
    tot2 = model.fetch_totals(batch)
    ll2, _ = model.fetch_log_likelihoods(batch)
    np.testing.assert_array_equal(ll2, ll)
    assert tot2[0] == tot[0]


def test_reversible_model_rerooting_invariance_full_batch(ra):
    # reference tests/test_mjp.py:126-137 property at batch scale
    cfg = ra.synth.make_config('c2', nsites=5000)
    T, n, leaves = cfg['T'], cfg['nstates'], cfg['leaves']
    states = cfg['leaf_states'].astype(np.uint8)
    base = None
    for root in (0, 5, 126, 64):
        ll, st = ra.mjp.get_log_likelihoods(
            T, root, n, leaves, states, kind='state',
            root_distn=cfg['root_distn'], Q_default=cfg['Q_default'])
        assert not st.any()
        if base is None:
            base = ll
        else:
            np.testing.assert_allclose(ll, base, rtol=1e-10)


@pytest.mark.parametrize('n', [1, 2, 3, 4])
def test_single_launch_step_is_the_two_launch_step(ra, n):
    """n <= 4, tree-specialised lane kernel: rt_step runs ONE launch -- the kernel computes
    the exponential of every edge in its prologue (the text of the library's expm kernel,
    csrc/expm_small.inc), workgroup 0 leaves the matrices where the expm launch would have,
    an extra workgroup carries the previous step's batch sum.  Bit for bit the two-launch step:
    transition matrices, order / squarings, log-likelihoods, totals; one rate matrix for
    all edges and one per edge; the same batch in consecutive steps (two partial-sum buffers)."""
    rng = np.random.RandomState(40 + n)
    T, root, leaves = ra.synth.balanced_tree(16, seed=3)
    nedges = T.number_of_edges()
    set_option = ra.lib.lib().rt_set_option
    for per_edge in (False, True):
        nq = nedges if per_edge else 1
        Q = rng.exponential(size=(nq, n, n)) * rng.choice([1e-3, 0.3, 4.0], size=(nq, 1, 1))
        for q in Q:
            np.fill_diagonal(q, 0)
            q -= np.diag(q.sum(axis=1))
        model = ra.device.TreeModel(T, root, n)
        ta = model.tree
        node_q = np.zeros(ta.nnodes, dtype=np.int64)
        if per_edge:
            node_q[1:] = rng.permutation(nedges)
        t = np.concatenate([[0.0], rng.uniform(0.01, 2.0, size=ta.nnodes - 1)])
        w = rng.uniform(0.1, 1.0, size=n)
        dense = rng.uniform(0.05, 1.0, size=(3000, len(leaves), n))
        ra.lib.check(set_option(b'jit', 1))
        try:
            out = {}
            for fused in ('1', '0'):
                os.environ['RAOTEH_JIT_FUSE_EXPM'] = fused
                model.set_rates(Q=Q, node_q=node_q, t=t)
                model.set_root_distn(w)
                b1 = model.upload_sites(leaves, dense[:2000], kind='dense')
                b2 = model.upload_sites(leaves, dense[2000:], kind='dense')
                P0, info0 = model.get_transitions(), model.expm_info()
                model.prune(b1)
                model.prune(b2)
                ref = (model.fetch_log_likelihoods(b1), model.fetch_totals(b1), model.fetch_totals(b2))
                assert ',expm' not in ra.ctx.kernel_time(1)[2]
                # poison what a step must rebuild
                model.set_transitions(np.full_like(P0, 0.25))
                for b in (b1, b1, b2, b1):
                    model.step(b)
                assert (',expm' in ra.ctx.kernel_time(1)[2]) == (fused == '1'), ra.ctx.kernel_time(1)
                np.testing.assert_array_equal(model.get_transitions(), P0)
                np.testing.assert_array_equal(model.expm_info(), info0)
                got = (model.fetch_log_likelihoods(b1), model.fetch_totals(b1), model.fetch_totals(b2))
                for a, b in zip(ref, got):
                    np.testing.assert_array_equal(a[0] if isinstance(a, tuple) else a,
                                                  b[0] if isinstance(b, tuple) else b)
                # a step without recomputation, and a plain launch, still see the right table
                model.step(b2, recompute_transitions=False)
                model.prune(b1)
                np.testing.assert_array_equal(model.fetch_totals(b2), ref[2])
                np.testing.assert_array_equal(model.fetch_totals(b1), ref[1])
                out[fused] = (P0, ref[0][0])
                b1.close()
                b2.close()
            np.testing.assert_array_equal(out['1'][0], out['0'][0])
            np.testing.assert_array_equal(out['1'][1], out['0'][1])
        finally:
            ra.lib.check(set_option(b'jit', -1))
            os.environ.pop('RAOTEH_JIT_FUSE_EXPM', None)
        model.close()


def test_deferred_reduce_gives_the_same_totals(ra):
    # rt_step leaves the fixed-order reduction of the batch sum to the next expm launch
    # (one extra workgroup) or to whoever reads the totals first; the numbers are those
    # of the stand-alone reduce kernel bit for bit
    cfg = ra.synth.make_config('c2', nsites=30011)
    model = ra.device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'])
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    dense = ra.synth.leaf_likelihoods(cfg)
    b1 = model.upload_sites(cfg['leaves'], dense[:20000])
    b2 = model.upload_sites(cfg['leaves'], dense[20000:])
    model.prune(b1)                       # reduce launched right away
    model.prune(b2)
    t1, t2 = model.fetch_totals(b1), model.fetch_totals(b2)
    assert t1[2] == 20000 and t2[2] == 10011
    for order in ((b1, b2, b1, b1, b2), (b2, b2, b1)):
        for b in order:
            model.step(b)                 # reduce of the previous step rides on this expm
        np.testing.assert_array_equal(model.fetch_totals(b1), t1)
        np.testing.assert_array_equal(model.fetch_totals(b2), t2)
    model.step(b1)
    model.step(b2)
    ra.ctx.sync()                         # flushes the reduction still pending
    np.testing.assert_array_equal(model.fetch_totals(b2), t2)
    # a batch may go away while its reduction is pending
    model.step(b1)
    b3 = model.upload_sites(cfg['leaves'], dense[:777])
    model.step(b3)
    b3.close()
    model.step(b2)
    np.testing.assert_array_equal(model.fetch_totals(b1), t1)
    np.testing.assert_array_equal(model.fetch_totals(b2), t2)
    # the same with a model whose expm kernel is the workgroup-per-matrix one
    cfg5 = ra.synth.make_config('c5', nsites=3000)
    m5 = ra.device.TreeModel(cfg5['T'], cfg5['root'], cfg5['nstates'])
    m5.set_rates(Q_default=cfg5['Q_default'])
    m5.set_root_distn(cfg5['root_distn'])
    c1 = m5.upload_sites(cfg5['leaves'], ra.synth.leaf_likelihoods(cfg5))
    m5.prune(c1)
    want = m5.fetch_totals(c1)
    for _ in range(3):
        m5.step(c1)
    np.testing.assert_array_equal(m5.fetch_totals(c1), want)
    model.step(b1)                        # two models of one context interleave
    m5.step(c1)
    model.step(b2)
    np.testing.assert_array_equal(m5.fetch_totals(c1), want)
    np.testing.assert_array_equal(model.fetch_totals(b1), t1)


def test_options_are_per_context_and_kernels_are_verified(ra):
    cfg = ra.synth.make_config('c5', nsites=500)
    dense = ra.synth.leaf_likelihoods(cfg)
    other = ra.device.Context(0)
    names = {}
    for ctx, jit in ((ra.ctx, None), (other, 0), (ra.ctx, 1)):
        if jit is not None:
            ctx.set_option('jit', jit)
        model = ra.device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'], ctx=ctx)
        model.set_rates(Q_default=cfg['Q_default'])
        model.set_root_distn(cfg['root_distn'])
        b = model.upload_sites(cfg['leaves'], dense)
        ll, st = model.log_likelihoods(b)
        names[(ctx is other, jit)] = (b.kernel_name, ll)
    ra.ctx.set_option('jit', None)
    # 500 sites: automatic = interpreter; the other context was told "never", this one
    # "always" afterwards -- neither setting leaked into the other context
    assert names[(False, None)][0].startswith('prune_mfma_solo')
    assert names[(True, 0)][0].startswith('prune_mfma_solo')
    assert names[(False, 1)][0].startswith('prune_tree_jit_mfma')
    np.testing.assert_array_equal(names[(False, 1)][1], names[(True, 0)][1])
    with pytest.raises(ValueError):
        ra.ctx.set_option('no_such_option', 1)
    other.close()


def test_destruction_order_is_refused_not_undefined(ra):
    """include/raoteh_hip.h, ownership: rt_model_destroy while a batch of the model lives and
    rt_ctx_destroy while a model or chain batch of the context lives return RT_ERR_INVALID and
    destroy nothing; in the right order everything goes.  Straight through ctypes."""
    from ctypes import byref, c_void_p, c_double, c_int32, c_int64, c_uint64, POINTER
    L = ra.lib.lib()
    ctx = c_void_p()
    assert L.rt_ctx_create(0, byref(ctx)) == 0
    idx = np.array([1, 2], dtype=np.int64)
    ptr = np.array([0, 2, 2, 2], dtype=np.int64)
    p64 = lambda a: a.ctypes.data_as(POINTER(c_int64))
    pf = lambda a: a.ctypes.data_as(POINTER(c_double))
    model = c_void_p()
    assert L.rt_model_create(ctx, 3, 4, p64(idx), p64(ptr), byref(model)) == 0
    esd = np.tile(np.full((4, 4), 0.25), (3, 1, 1))
    assert L.rt_model_set_transitions(model, pf(esd)) == 0
    obs = np.array([1, 2], dtype=np.int64)
    data = np.array([[0, 1], [2, 255]], dtype=np.uint8)
    sites, twin = c_void_p(), c_void_p()
    assert L.rt_sites_create(model, 2, ra.lib.RT_OBS_STATE, 2, p64(obs), data.ctypes.data_as(c_void_p),
                             byref(sites)) == 0
    assert L.rt_sites_clone(sites, byref(twin)) == 0
    assert L.rt_prune(model, sites) == 0
    # chains on the same context
    parent = np.array([-1, 0, 0], dtype=np.int32)
    branch = np.array([0.0, 0.3, 0.2])
    P = np.full((4, 4), 0.25)
    rates = np.full(4, 1.5)
    masks = np.full((5, 3), 15, dtype=np.uint64)
    chains = c_void_p()
    assert L.rt_chains_create(ctx, 3, parent.ctypes.data_as(POINTER(c_int32)), pf(branch), 4, pf(P),
                              pf(rates), None, 5, masks.ctypes.data_as(POINTER(c_uint64)),
                              c_uint64(1), byref(chains)) == 0
    # wrong order: refused, nothing destroyed, everything still works
    assert L.rt_model_destroy(model) == ra.lib.RT_ERR_INVALID
    assert b'2 site batch' in L.rt_last_error()
    assert L.rt_ctx_destroy(ctx) == ra.lib.RT_ERR_INVALID
    assert b'1 model' in L.rt_last_error() and b'1 chain' in L.rt_last_error()
    assert L.rt_prune(model, twin) == 0
    ll = np.zeros(2)
    assert L.rt_sites_get_logliks(twin, pf(ll), None) == 0
    # root weights of one (not a distribution): 4 x 0.25 x 0.25 and 4 x 0.25 x 1
    np.testing.assert_allclose(ll, np.log([0.25, 1.0]), rtol=1e-14, atol=1e-15)
    assert L.rt_chains_sweep(chains, 2) == 0
    # one child left: still refused
    assert L.rt_sites_destroy(sites) == 0
    assert L.rt_model_destroy(model) == ra.lib.RT_ERR_INVALID
    assert L.rt_sites_destroy(twin) == 0
    assert L.rt_model_destroy(model) == 0
    assert L.rt_ctx_destroy(ctx) == ra.lib.RT_ERR_INVALID       # the chains
    assert L.rt_chains_destroy(chains) == 0
    assert L.rt_ctx_destroy(ctx) == 0
    assert L.rt_model_destroy(None) == 0 and L.rt_sites_destroy(None) == 0


def test_objects_in_a_reference_cycle_are_finalised_in_a_safe_order(ra):
    """A context, a model, two batches and a chain batch that die in ONE garbage-collection
    cycle (a cycle through a traceback kept them alive in round 2: the model was finalised
    before its batch and the process segfaulted): the Python classes close children first."""
    import gc
    from raoteh_amd import _sampler
    T, root, leaves = ra.synth.balanced_tree(4)
    Q, distn = ra.synth.hky85()

    def build():
        ctx = ra.device.Context(0)
        model = ra.device.TreeModel(T, root, 4, ctx=ctx)
        model.set_rates(Q_default=Q)
        batch = model.upload_sites(leaves, np.zeros((3, len(leaves)), dtype=np.uint8), kind='state')
        twin = batch.clone()
        model.prune(batch)
        chains = _sampler.DeviceHistoryBatch(T, root, Q, nchains=4, seed=2, ctx=ctx)
        chains.sweep(1)
        cycle = {'ctx': ctx, 'model': model, 'batches': [batch, twin], 'chains': chains}
        cycle['self'] = cycle                       # only the cycle collector can free this
        ctx.cycle = cycle
        return ctx._h.value

    for _ in range(3):
        gc.collect()
        assert build()
        gc.collect()
    # an explicit close of the context closes what hangs on it
    ctx = ra.device.Context(0)
    model = ra.device.TreeModel(T, root, 4, ctx=ctx)
    model.set_rates(Q_default=Q)
    batch = model.upload_sites(leaves, np.zeros((3, len(leaves)), dtype=np.uint8), kind='state')
    ctx.close()
    assert not batch._h and not model._h and not ctx._h


def test_background_compiles_are_bounded_over_many_trees(ra, tmp_path, monkeypatch):
    """A caller that creates batches over ever new trees with "jit" on automatic (a search
    over topologies) gets at most RAOTEH_JIT_MAX_JOBS (default 2) compile threads at a time:
    the batches created meanwhile stay on the interpreter kernel -- same numbers -- and a
    tree met again later gets its kernel then (from the cache if it was compiled before)."""
    monkeypatch.setenv('RAOTEH_JIT_CACHE_DIR', str(tmp_path / 'jit'))
    rng = np.random.RandomState(4242)
    n, nsites = 61, 3000
    ctx = ra.device.Context(0)
    ctx.set_option('jit_async', 1)
    made = []
    for k in range(6):
        T, root, obs_nodes, w = _random_case(ra, rng, n, 40 + k, nsites)
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        dense = rng.uniform(0.05, 1.0, size=(nsites, len(obs_nodes), n))
        model = ra.device.TreeModel(T, root, n, ctx=ctx)
        model.set_transitions(esd)
        model.set_root_distn(w)
        batch = model.upload_sites(obs_nodes, dense, kind='dense')
        ll0, st0 = model.log_likelihoods(batch)
        made.append((model, batch, ll0, (T, root, obs_nodes, w, esd, dense),
                     (idx, ptr, [pre.index(v) for v in obs_nodes])))
    # six creates take milliseconds each, a 61-state compile a second or more: the first two
    # trees got a job, not every later one did
    kinds = []
    for model, batch, ll0, case, (idx, ptr, oidx) in made:
        batch.wait_for_kernel()
        ll1, _ = model.log_likelihoods(batch)
        np.testing.assert_array_equal(ll0, ll1)
        want, wst = orc.batch_log_likelihoods(idx, ptr, case[4], oidx, case[5], case[3])
        np.testing.assert_allclose(ll0, want, rtol=RTOL_LL)
        assert batch.kernel_name.startswith(('prune_tree_jit', 'prune_mfma')), batch.kernel_name
        kinds.append(batch.kernel_name.startswith('prune_tree_jit'))
    assert kinds[0] and kinds[1], kinds
    assert not all(kinds), kinds
    # every job has finished: a tree that stayed on the interpreter gets its kernel on the
    # next visit, the first tree's kernel comes from the cache
    late = kinds.index(False)
    for k in (late, 0):
        T, root, obs_nodes, w, esd, dense = made[k][3]
        model = ra.device.TreeModel(T, root, n, ctx=ctx)
        model.set_transitions(esd)
        model.set_root_distn(w)
        batch = model.upload_sites(obs_nodes, dense, kind='dense')
        batch.wait_for_kernel()
        ll2, _ = model.log_likelihoods(batch)
        assert batch.kernel_name.startswith('prune_tree_jit'), (k, batch.kernel_name)
        np.testing.assert_array_equal(made[k][2], ll2)
        batch.close()
        model.close()
    for model, batch, _, _, _ in made:
        batch.close()
        model.close()
    # RAOTEH_JIT_MAX_JOBS=0: never compile in the background
    monkeypatch.setenv('RAOTEH_JIT_MAX_JOBS', '0')
    T, root, obs_nodes, w = _random_case(ra, rng, n, 33, nsites)
    pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
    dense = rng.uniform(0.05, 1.0, size=(nsites, len(obs_nodes), n))
    model = ra.device.TreeModel(T, root, n, ctx=ctx)
    model.set_transitions(esd)
    model.set_root_distn(w)
    batch = model.upload_sites(obs_nodes, dense, kind='dense')
    batch.wait_for_kernel()
    ll, st = model.log_likelihoods(batch)
    assert batch.kernel_name.startswith('prune_mfma'), batch.kernel_name
    want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in obs_nodes],
                                          dense, w)
    np.testing.assert_allclose(ll, want, rtol=RTOL_LL)
    batch.close()
    model.close()
    ctx.close()


def test_background_compile_and_persistent_code_object_cache(ra, tmp_path, monkeypatch):
    """jit_async: rt_sites_create returns without waiting for hiprtc, the batch (and a clone
    made meanwhile) runs the interpreter kernel and switches to the tree-specialised one when
    the compile thread is done -- same log-likelihoods bit for bit; the code object lands in
    the cache directory and a second context (standing in for a second process) loads it from
    there instead of compiling."""
    import time
    monkeypatch.setenv('RAOTEH_JIT_CACHE_DIR', str(tmp_path / 'jit'))
    rng = np.random.RandomState(77)
    for n, nnodes in ((20, 37), (61, 29)):
        nsites = 6000
        T, root, obs_nodes, w = _random_case(ra, rng, n, nnodes, nsites)
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        dense = rng.uniform(0.05, 1.0, size=(nsites, len(obs_nodes), n))
        ctx = ra.device.Context(0)
        ctx.set_option('jit_async', 1)
        model = ra.device.TreeModel(T, root, n, ctx=ctx)
        model.set_transitions(esd)
        model.set_root_distn(w)
        t0 = time.perf_counter()
        batch = model.upload_sites(obs_nodes, dense, kind='dense')
        create_s = time.perf_counter() - t0
        twin = batch.clone()
        ll0, st0 = model.log_likelihoods(batch)
        first = batch.kernel_name
        tot0 = model.fetch_totals(batch)
        batch.wait_for_kernel()
        ll1, st1 = model.log_likelihoods(batch)
        assert batch.kernel_name.startswith('prune_tree_jit'), batch.kernel_name
        np.testing.assert_array_equal(ll0, ll1)
        np.testing.assert_array_equal(st0, st1)
        tot1 = model.fetch_totals(batch)
        assert tot1[2] == nsites and tot1[0] == pytest.approx(tot0[0], rel=1e-13)
        assert tot1[0] == pytest.approx(ll1.sum(), rel=1e-12)
        cold = batch.jit_compile_seconds
        assert cold > 0
        # (on a box whose compiler cache is warm the job may finish before the first launch)
        assert first.startswith(('prune_mfma', 'prune_tree_jit')), first
        for _ in range(200):                      # the clone swaps on its own, without waiting
            ll2, _ = model.log_likelihoods(twin)
            if twin.kernel_name.startswith('prune_tree_jit'):
                break
        assert twin.kernel_name.startswith('prune_tree_jit')
        np.testing.assert_array_equal(ll0, ll2)
        files = sorted(p.name for p in (tmp_path / 'jit').glob('*.hsaco'))
        assert files, 'no code object was written to the cache directory'
        # a second context: nothing in its memory cache, the code object on disk
        ctx2 = ra.device.Context(0)
        ctx2.set_option('jit_async', 0)
        model2 = ra.device.TreeModel(T, root, n, ctx=ctx2)
        model2.set_transitions(esd)
        model2.set_root_distn(w)
        batch2 = model2.upload_sites(obs_nodes, dense, kind='dense')
        ll3, _ = model2.log_likelihoods(batch2)
        assert batch2.kernel_name == batch.kernel_name
        np.testing.assert_array_equal(ll0, ll3)
        warm = batch2.jit_compile_seconds
        assert sorted(p.name for p in (tmp_path / 'jit').glob('*.hsaco')) == files
        assert warm < 0.25, (warm, cold)
        print('n=%d: rt_sites_create %.3f s, background compile %.3f s, from the disk cache %.4f s'
              % (n, create_s, cold, warm))
        ctx2.close()
        ctx.close()
    # the lane family (n <= 4): the kernel's resident layout differs from the interpreter's
    # (sites per wave, one byte per observed state), so the batch keeps the caller's
    # observations on the device and is packed again when the kernel arrives
    for n, kind in ((4, 'state'), (3, 'dense'), (4, 'mask')):
        nsites = 30000
        T, root, obs_nodes, w = _random_case(ra, rng, n, 41, nsites)
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        states = rng.randint(0, n, size=(nsites, len(obs_nodes))).astype(np.uint8)
        states[rng.uniform(size=states.shape) < 0.2] = 255
        dense = np.ones((nsites, len(obs_nodes), n))
        seen = states != 255
        dense[seen] = 0.0
        ii, kk = np.nonzero(seen)
        dense[ii, kk, states[ii, kk]] = 1.0
        masks = np.zeros((nsites, len(obs_nodes)), dtype=np.uint64)
        for sidx in range(n):
            masks |= dense[..., sidx].astype(np.uint64) << np.uint64(sidx)
        data = {'state': states, 'dense': dense, 'mask': masks}[kind]
        ctx = ra.device.Context(0)
        ctx.set_option('jit_async', 1)
        model = ra.device.TreeModel(T, root, n, ctx=ctx)
        model.set_transitions(esd)
        model.set_root_distn(w)
        batch = model.upload_sites(obs_nodes, data, kind=kind)
        bytes0 = batch.device_bytes
        ll0, st0 = model.log_likelihoods(batch)
        first = batch.kernel_name
        batch.wait_for_kernel()
        ll1, st1 = model.log_likelihoods(batch)
        assert batch.kernel_name.startswith('prune_tree_jit<%d' % n), batch.kernel_name
        assert first.startswith(('prune_lane', 'prune_tree_jit')), first
        np.testing.assert_array_equal(ll0, ll1)
        np.testing.assert_array_equal(st0, st1)
        if kind != 'dense':
            assert batch.device_bytes < bytes0 / 8          # one byte per leaf now
            assert 'states' in batch.kernel_name or 'masks' in batch.kernel_name
        twin = batch.clone()
        ll2, _ = model.log_likelihoods(twin)
        np.testing.assert_array_equal(ll0, ll2)
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in obs_nodes],
                                              dense[:200], w)
        np.testing.assert_allclose(ll1[:200][wst == 0], want[wst == 0], rtol=RTOL_LL)
        tot = model.fetch_totals(batch)
        ok = np.isfinite(ll1)
        assert tot[2] == nsites and tot[0] == pytest.approx(ll1[ok].sum(), rel=1e-12)
        # the expectation step works on the switched batch too
        if kind == 'state':
            Q = rng.exponential(size=(n, n))
            np.fill_diagonal(Q, 0.0)
            Q -= np.diag(Q.sum(axis=1))
            model.set_rates(Q_default=Q)
            d1 = model.expected_history_statistics(batch)
            ctx.set_option('jit', 0)
            plain = model.upload_sites(obs_nodes, data, kind=kind)
            ctx.set_option('jit', None)
            d2 = model.expected_history_statistics(plain)
            for a, b in zip(d1, d2):
                np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)
        ctx.close()
    # RAOTEH_JIT_CACHE=0: nothing is read or written
    monkeypatch.setenv('RAOTEH_JIT_CACHE', '0')
    monkeypatch.setenv('RAOTEH_JIT_CACHE_DIR', str(tmp_path / 'off'))
    ctx = ra.device.Context(0)
    ctx.set_option('jit_async', 0)
    model = ra.device.TreeModel(T, root, n, ctx=ctx)
    model.set_transitions(esd)
    batch = model.upload_sites(obs_nodes, dense, kind='dense')
    model.log_likelihoods(batch)
    assert batch.kernel_name.startswith('prune_tree_jit')
    assert not (tmp_path / 'off').exists()
    ctx.close()


def test_timing_and_clone(ra):
    cfg = ra.synth.make_config('c2', nsites=2000)
    model = ra.device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'])
    model.set_root_distn(cfg['root_distn'])
    ctx = model.ctx
    ctx.set_timing(True)
    ctx.reset_timing()
    model.set_rates(Q_default=cfg['Q_default'])
    batch = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8),
                               kind='state')
    twin = batch.clone()
    for _ in range(3):
        model.prune(batch)
        model.prune(twin)
    a, _ = model.fetch_log_likelihoods(batch)
    b, _ = model.fetch_log_likelihoods(twin)
    np.testing.assert_array_equal(a, b)
    ms, cnt, name = ctx.kernel_time(1)
    assert cnt == 6 and ms > 0 and name.startswith(('prune_lane', 'prune_tree_jit'))
    ms, cnt, name = ctx.kernel_time(0)
    assert cnt == 1 and name.startswith('expm')
    ctx.set_timing(False)
    assert batch.device_bytes == 32 * 64 * 64 * 4 * 8


def test_rccl_reduce_single_rank(ra):
    """The N > 1 data path on one GPU: a 1-rank RCCL communicator; per-batch and
    grouped all-reduce of the totals (csrc/comm.hip) leave a single rank's totals
    unchanged, in neighbouring and in scattered slots of the totals arena, and a
    batch can be pruned again right after."""
    cfg = ra.synth.make_config('c2', nsites=700)
    model = ra.device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'])
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    states = cfg['leaf_states'].astype(np.uint8)
    first = model.upload_sites(cfg['leaves'], states, kind='state')
    batches = [first] + [first.clone() for _ in range(4)]
    want = []
    for b in batches:
        model.step(b)
        want.append(model.fetch_totals(b))
    assert want[0][2] == 700 and want[0][1] == 0
    ctx = ra.ctx
    ctx.comm_init(1, 0, ra.device.Context.comm_unique_id())
    try:
        for b in batches:
            model.step(b)
        model.allreduce(batches[0])
        model.allreduce_group(batches[1:])             # neighbouring slots: one collective
        for b, w in zip(batches, want):
            np.testing.assert_array_equal(model.fetch_totals(b), w)
        del batches[2]                                  # leaves a hole in the arena
        import gc
        gc.collect()
        extra = first.clone()                           # reuses the freed slot
        group = [batches[3], extra, batches[0]]         # not ascending neighbours
        for b in group:
            model.step(b)
        model.allreduce_group(group)
        for b in group:
            model.step(b, recompute_transitions=False)  # waits for the collective
            np.testing.assert_array_equal(model.fetch_totals(b), want[0])
    finally:
        ctx.comm_destroy()
    # after the communicator is gone the batches still work
    model.step(first)
    np.testing.assert_array_equal(model.fetch_totals(first), want[0])


def test_p53_alignment_from_files(ra):
    """The reference's p53 example (examples/p53/p53.py:62-100) end to end: PHYLIP
    alignment + newick tree + genetic code -> MG94 -> per-column log-likelihoods,
    against the oracle; with and without site-pattern compression."""
    from test_io_cpu import p53_problem
    T, root, leaves, states, Q, distn, _, _ = p53_problem()
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, root, 61, Q_default=Q)
    dense = np.zeros((393, 25, 61))
    ii, kk = np.indices(states.shape)
    dense[ii, kk, states] = 1.0
    want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in leaves],
                                          dense, distn)
    ll, st = ra.mjp.get_log_likelihoods(T, root, 61, leaves, states, kind='state',
                                        root_distn=distn, Q_default=Q)
    np.testing.assert_array_equal(st, wst)
    np.testing.assert_allclose(ll, want, rtol=RTOL_LL)
    llc, stc = ra.mjp.get_log_likelihoods(T, root, 61, leaves, states, kind='state',
                                          root_distn=distn, Q_default=Q, compress=True)
    np.testing.assert_allclose(llc, want, rtol=RTOL_LL)
    np.testing.assert_array_equal(stc, wst)
    tot = ra.mjp.get_total_log_likelihood(T, root, 61, leaves, states, kind='state',
                                          root_distn=distn, Q_default=Q)
    totc = ra.mjp.get_total_log_likelihood(T, root, 61, leaves, states, kind='state',
                                           root_distn=distn, Q_default=Q, compress=True)
    assert tot == pytest.approx(-11202.4288003113, rel=1e-11)
    assert totc == pytest.approx(tot, rel=1e-12)
    # the reference's own single-site call on one column (p53.py:88-97)
    col = 7
    allowed = dict((v, set(range(61))) for v in T)
    for leaf, s0 in zip(leaves, states[col]):
        allowed[leaf] = {int(s0)}
    lk = ra.mjp.get_likelihood(T, allowed, root, 61, root_distn=distn, Q_default=Q)
    assert np.log(lk) == pytest.approx(want[col], rel=RTOL_LL)


def test_tree_specialised_kernel_large_tree(ra):
    """A 256-leaf tree: the step-ordered P table (65 KB at 4 states) no longer fits
    one copy per wave, so the tree-specialised kernel shares it between the waves of
    a workgroup; results stay bit-identical with the interpreter kernel."""
    set_option = ra.lib.lib().rt_set_option
    T, root, leaves = ra.synth.balanced_tree(256, seed=5)
    Q, pi = ra.synth.hky85()
    rng = np.random.RandomState(77)
    states = rng.randint(0, 4, size=(3000, 256)).astype(np.uint8)
    states[rng.uniform(size=states.shape) < 0.05] = 255
    out = {}
    for jit in (0, 1):
        ra.lib.check(set_option(b'jit', jit))
        try:
            out[jit] = ra.mjp.get_log_likelihoods(T, root, 4, leaves, states, kind='state',
                                                  root_distn=pi, Q_default=Q)
            out[jit] += (ra.ctx.kernel_time(1)[2],)
        finally:
            ra.lib.check(set_option(b'jit', -1))
    assert out[0][2].startswith('prune_lane') and out[1][2].startswith('prune_tree_jit')
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    # and against the oracle on a few sites
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, root, 4, Q_default=Q)
    dense = np.ones((40, 256, 4))
    sub = states[:40]
    obs = sub != 255
    dense[obs] = 0.0
    ii, kk = np.nonzero(obs)
    dense[ii, kk, sub[ii, kk]] = 1.0
    want, _ = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in leaves],
                                        dense, pi)
    np.testing.assert_allclose(out[1][0][:40], want, rtol=RTOL_LL)


def test_randomised_soak_short(ra, monkeypatch):
    """A few seconds of tests/soak/soak.py and tests/soak/soak_passes.py (random trees, state
    counts, encodings, tilings; specialised vs interpreter kernel bit for bit, both
    against the oracle).  The long runs are recorded in DESIGN.md section 5."""
    import importlib.util
    tools = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'soak')
    for name, seed in (('soak', 4242), ('soak_passes', 4243)):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        monkeypatch.setattr(sys, 'argv', [name, '6', str(seed)])
        mod.main()          # exits the process with status 1 on any mismatch


@pytest.mark.parametrize('n', [1, 2, 3, 4])
def test_compact_state_batches_match_dense(ra, n):
    """uint8 states stay states on the device when a tree-specialised lane kernel runs
    them (64 B per site instead of 2 KB); the kernel expands them to 0/1 vectors in
    registers, so the results equal the dense encoding's bit for bit."""
    rng = np.random.RandomState(500 + n)
    set_option = ra.lib.lib().rt_set_option
    for nnodes, nsites in ((2, 70), (11, 129), (40, 3000), (130, 700)):
        T, root, obs_nodes, w = _random_case(ra, rng, n, nnodes, nsites)
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        states = rng.randint(0, n, size=(nsites, len(obs_nodes))).astype(np.uint8)
        states[rng.uniform(size=states.shape) < 0.2] = 255
        dense = np.ones((nsites, len(obs_nodes), n))
        seen = states != 255
        dense[seen] = 0.0
        ii, kk = np.nonzero(seen)
        dense[ii, kk, states[ii, kk]] = 1.0
        model = ra.device.TreeModel(T, root, n)
        model.set_transitions(esd)
        model.set_root_distn(w)
        out = {}
        for key, jit, bs, data, kind in (('dense', 1, 64, dense, 'dense'),
                                         ('state', 1, 64, states, 'state'),
                                         ('state51', 1, 51, states, 'state'),
                                         ('interp', 0, 0, states, 'state')):
            ra.lib.check(set_option(b'jit', jit))
            ra.lib.check(set_option(b'jit_block_sites', bs))
            try:
                batch = model.upload_sites(obs_nodes, data, kind=kind)
                ll, st = model.log_likelihoods(batch)
                out[key] = (ll, st, ra.ctx.kernel_time(1)[2], batch.device_bytes)
                ll2, _ = model.log_likelihoods(batch.clone())
                np.testing.assert_array_equal(ll, ll2)
            finally:
                ra.lib.check(set_option(b'jit', -1))
                ra.lib.check(set_option(b'jit_block_sites', 0))
        assert out['state'][2].endswith(',states>'), out['state'][2]
        assert not out['dense'][2].endswith(',states>')
        assert out['state'][3] * 8 <= out['dense'][3]
        for key in ('state', 'state51', 'interp'):
            np.testing.assert_array_equal(out['dense'][0], out[key][0])
            np.testing.assert_array_equal(out['dense'][1], out[key][1])
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in obs_nodes],
                                              dense[:64], w)
        np.testing.assert_array_equal(out['state'][1][:64] & 1, wst)
        ok = wst == 0
        np.testing.assert_allclose(out['state'][0][:64][ok], want[ok], rtol=RTOL_LL)
        # allowed-set masks (type y; the IUPAC codes of a DNA alignment): one byte per
        # leaf as well, a 2^n-column table instead of the n + 1 columns of states
        masks = rng.randint(1, 1 << n, size=(nsites, len(obs_nodes))).astype(np.uint64)
        masks[rng.uniform(size=masks.shape) < 0.02] = 0          # a few impossible sites
        dm = ((masks[..., None] >> np.arange(n, dtype=np.uint64)) & 1).astype(np.float64)
        outm = {}
        for key, jit, data, kind in (('dense', 1, dm, 'dense'), ('mask', 1, masks, 'mask'),
                                     ('interp', 0, masks, 'mask')):
            ra.lib.check(set_option(b'jit', jit))
            try:
                batch = model.upload_sites(obs_nodes, data, kind=kind)
                ll, st = model.log_likelihoods(batch)
                outm[key] = (ll, st, ra.ctx.kernel_time(1)[2], model.fetch_totals(batch))
            finally:
                ra.lib.check(set_option(b'jit', -1))
        assert outm['mask'][2].endswith(',masks>'), outm['mask'][2]
        for key in ('mask', 'interp'):
            np.testing.assert_array_equal(outm['dense'][0], outm[key][0])
            np.testing.assert_array_equal(outm['dense'][1], outm[key][1])
        np.testing.assert_array_equal(outm['dense'][3], outm['mask'][3])
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in obs_nodes],
                                              dm[:64], w)
        np.testing.assert_array_equal(outm['mask'][1][:64] & 1, wst)
        ok = wst == 0
        if ok.any():
            np.testing.assert_allclose(outm['mask'][0][:64][ok], want[ok], rtol=RTOL_LL)


# ---------------------------------------------------------------------------
# Rao-Teh sweep core: ragged batches of trees with one shared matrix (csrc/forest.hip)
# ---------------------------------------------------------------------------

def _forest_cases():
    fx = load_golden('forest')
    out = []
    for c in fx['cases']:
        T = nx.Graph()
        T.add_nodes_from(c['chunk_nodes'])
        T.add_edges_from((a, b) for a, b in c['chunk_edges'])
        out.append((c, T))
    return out


def test_forest_passes_match_the_reference(ra):
    """pset / set / pmap of every chunk tree of the fixture in ONE ragged batch against
    the reference's un-accelerated functions with P_default = the uniformized matrix
    (tests/golden/forest.json, tools/gen_golden.py fixture_forest)."""
    from raoteh_amd import _forest
    cases = _forest_cases()
    # each fixture case has its own P: a batch per case = its tree with its observations
    # plus up to three other trees of the same state count without observations, so that
    # the batch is ragged
    for c, T in cases:
        n = c['nstates']
        P = np.array(c['P'])
        allowed = dict((int(v), set(ss)) for v, ss in c['allowed'].items())
        other = [(T2, c2) for c2, T2 in cases if c2['nstates'] == n][:3]
        trees = [(T, c['root'])] + [(T2, c2['root']) for T2, c2 in other]
        obs = [allowed] + [None] * len(other)
        forest = _forest.Forest(trees)
        sets, pmaps = _forest.get_node_to_set_and_pmap(forest, P, obs)
        for v in T:
            assert sets[0][v] == set(c['set'][str(v)]), (v, sets[0][v], c['set'][str(v)])
            np.testing.assert_allclose(pmaps[0][v], c['pmap'][str(v)], rtol=1e-12, atol=0)
        # unrestricted trees: every state everywhere, pmap = 1 (P is stochastic)
        for k in range(1, len(trees)):
            for v in trees[k][0]:
                assert sets[k][v] == set(range(n))
                np.testing.assert_allclose(pmaps[k][v], 1.0, rtol=1e-12)


def test_pyfelscore_shared_matrix_passes_and_lower_bound(ra):
    """The pyfelscore names the Rao-Teh sweep and examples/p53/liwen.py call, with the
    reference's own argument lists: mcy_get_node_to_pset / get_node_to_set (_mcy.py:158,168,
    259: the tree CSR, the transition matrix as a boolean CSR, the int state mask in place)
    against the reference's un-accelerated twins (tests/golden/forest.json), and
    get_lb_transition_matrix (liwen.py:45) against the restated getp_lb (liwen.py:47-82)."""
    from raoteh_amd._tree import TreeArrays
    pyf = ra.pyf
    done = 0
    for c, T in _forest_cases():
        if c.get('single'):
            continue
        n = c['nstates']
        P = np.array(c['P'])
        ta = TreeArrays(T, c['root'])
        # _density.digraph_to_bool_csr of the matrix over its sorted states (_mcy.py:148-149)
        tptr = np.concatenate([[0], np.cumsum((P != 0).sum(axis=1))]).astype(np.int64)
        tidx = np.nonzero(P != 0)[1].astype(np.int64)
        mask = np.zeros((ta.nnodes, n), dtype=np.int64)
        for i, v in enumerate(ta.preorder_nodes):
            mask[i, sorted(c['allowed'][str(v)])] = 1
        pyf.mcy_get_node_to_pset(ta.indices, ta.indptr, tidx, tptr, mask)
        for i, v in enumerate(ta.preorder_nodes):
            assert set(np.nonzero(mask[i])[0]) == set(c['pset'][str(v)]), (v, mask[i])
        tmp = np.zeros(n, dtype=np.int64)
        pyf.get_node_to_set(ta.indices, ta.indptr, tidx, tptr, mask, tmp)
        for i, v in enumerate(ta.preorder_nodes):
            assert set(np.nonzero(mask[i])[0]) == set(c['set'][str(v)]), (v, mask[i])
        # the forward pass alone does not repeat the backward pass: from the unrestricted
        # mask it only removes what the ROOT's set cannot reach
        raw = np.ones((ta.nnodes, n), dtype=np.int64)
        raw[0] = 0
        raw[0, 0] = 1
        want = raw.copy()
        for i in range(ta.nnodes):
            for j in range(ta.indptr[i], ta.indptr[i + 1]):
                k = ta.indices[j]
                want[k] &= ((P != 0)[want[i] != 0].any(axis=0)).astype(np.int64)
        pyf.get_node_to_set(ta.indices, ta.indptr, tidx, tptr, raw, None)
        np.testing.assert_array_equal(raw, want)
        done += 1
    assert done >= 15
    rng = np.random.RandomState(8)
    for n in (3, 61, 122):
        Q = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) < 0.3)
        np.fill_diagonal(Q, 0.0)
        Q -= np.diag(Q.sum(axis=1))
        Q[1, 1] = Q[0, 0]
        for t in (0.01, 0.4):
            out = np.empty((n, n))
            pyf.get_lb_transition_matrix(t, Q, out)
            # (exp(-ra t) - exp(-rb t) cancels where two exit rates are close)
            np.testing.assert_allclose(out, orc.getp_lb(Q, t), rtol=1e-11, atol=1e-300)
            assert ((Q == 0) == (out == 0))[~np.eye(n, dtype=bool)].all()


def test_forest_sampling_follows_the_exact_posterior(ra):
    """Sampled states against the exact posterior node marginals of the reference
    (_mc0.get_node_to_distn): many sweeps of the same forest, frequencies within
    sampling error; zero-likelihood trees are flagged, not sampled; draws are
    reproducible and independent of the position of a tree in the batch."""
    from raoteh_amd import _forest
    cases = _forest_cases()
    reps = 4000
    for c, T in cases[:12]:
        n = c['nstates']
        P = np.array(c['P'])
        allowed = dict((int(v), set(ss)) for v, ss in c['allowed'].items())
        distn = np.array(c['root_distn'])
        forest = _forest.Forest([(T, c['root'])] * reps)
        obs = [allowed] * reps
        states, status = _forest.resample_states(forest, P, obs, root_distn=distn,
                                                 seed=99, sweep=3, return_status=True)
        if c['zero']:
            assert (status == 1).all()
            assert all(s == -1 for d in states for s in d.values())
            with pytest.raises(ra.pkg.StructuralZeroProb):
                _forest.resample_states(forest, P, obs, root_distn=distn, seed=99, sweep=3)
            continue
        assert not status.any()
        for v in T:
            want = np.array(c['distn'][str(v)])
            got = np.bincount([d[v] for d in states], minlength=n) / float(reps)
            # never a state outside the posterior support; frequencies within 5 sigma
            assert not got[want == 0].any(), (v, got, want)
            sigma = np.sqrt(np.maximum(want * (1 - want), 1e-12) / reps)
            assert np.all(np.abs(got - want) <= 5 * sigma + 1e-9), (v, got, want)
        # every parent -> child pair drawn is a transition P allows
        for d in states[:200]:
            for a, b in nx.bfs_edges(T, c['root']):
                assert P[d[a], d[b]] > 0
        # same (seed, sweep) -> same draws; another sweep -> other draws
        again, _ = _forest.resample_states(forest, P, obs, root_distn=distn, seed=99, sweep=3,
                                           return_status=True)
        assert again == states
        other, _ = _forest.resample_states(forest, P, obs, root_distn=distn, seed=99, sweep=4,
                                           return_status=True)
        if any(max(c['distn'][str(v)]) < 0.9 for v in T):     # not a degenerate posterior
            assert other != states


def test_forest_rejects_bad_layouts(ra):
    from raoteh_amd import _forest
    T = nx.path_graph(4)
    forest = _forest.Forest([(T, 0), (T, 2)])
    assert forest.node_offset.tolist() == [0, 4, 8] and forest.total == 8
    P = np.full((3, 3), 1.0 / 3)
    with pytest.raises(ValueError):
        _forest.get_node_to_set_and_pmap(forest, np.ones((3, 4)))
    with pytest.raises(ValueError):
        _forest.Forest([(T, 17)])
    with pytest.raises(ValueError):
        _forest.resample_states(forest, P, [{0: {5}}, None])
    # a corrupted CSR is caught by the library, not run
    bad = _forest.Forest([(T, 0)])
    bad.indices = np.array([0, 2, 3], dtype=np.int64)
    with pytest.raises(ValueError):
        _forest.get_node_to_set_and_pmap(bad, P)
    sets, pmaps = _forest.get_node_to_set_and_pmap(forest, P, [{3: {1}}, {0: {0, 2}}])
    assert sets[0][3] == {1} and sets[1][0] == {0, 2}
    np.testing.assert_allclose(pmaps[0][0], 1.0 / 3, rtol=1e-13)


@pytest.mark.parametrize('where', ['device', 'host'])
def test_rao_teh_sweeps_reproduce_the_posterior_expectations(ra, where):
    """The batched Rao-Teh sampler (raoteh_amd/_sampler.py around the device's forest
    sampler) is a Gibbs sampler whose stationary law is the posterior over histories:
    dwell times, transition counts and root states averaged over replicate chains must
    match the expected history statistics of the same observation (the expectation path,
    itself pinned to the oracle above).  Deterministic: fixed seeds, counter-based draws."""
    from raoteh_amd import _mjp_dense, _sampler
    cfg = ra.synth.make_config('c1', nsites=3)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    Q = cfg['Q_default']
    full = set(range(n))
    allowed = dict((v, full) for v in T)
    for leaf, s in zip(cfg['leaves'], cfg['leaf_states'][1]):
        allowed[leaf] = {int(s)}
    allowed[cfg['leaves'][2]] = {0, 3}                     # an ambiguous leaf
    want_dwell, want_root, want_trans = _mjp_dense.get_expected_history_statistics(
        T, allowed, root, n, root_distn=cfg['root_distn'], Q_default=Q)
    B, burn, keep = 3000, 8, 24
    cls = _sampler.DeviceHistoryBatch if where == 'device' else _sampler.HistoryBatch
    batch = cls(T, root, Q, node_to_allowed_states=allowed, nchains=B,
                root_distn=cfg['root_distn'], seed=11, ctx=ra.ctx)
    total = sum(d['weight'] for _, _, d in T.edges(data=True))
    dwell = np.zeros((B, n))
    trans = np.zeros((B, n, n))
    roots = np.zeros((B, n))
    for it in range(burn + keep):
        batch.sweep()
        if it < burn:
            continue
        d = batch.dwell_times()
        np.testing.assert_allclose(d.sum(axis=1), total, rtol=1e-12)
        dwell += d
        trans += batch.transition_counts()
        roots[np.arange(B), batch.root_states()] += 1
        # leaves keep to their allowed sets, whatever the sweep did
        node_states = batch.node_states
        for leaf in cfg['leaves']:
            st = node_states[:, batch.tree.node_to_index[leaf]]
            assert set(np.unique(st).tolist()) <= allowed[leaf]
    dwell /= keep
    trans /= keep
    roots /= keep

    def close(sample, expected, what):
        mean = sample.mean(axis=0)
        se = sample.std(axis=0, ddof=1) / np.sqrt(B)
        assert abs(mean - expected) <= 5 * se + 1e-3 * max(abs(expected), 1e-2), \
            '%s: %.5f vs %.5f (se %.5f)' % (what, mean, expected, se)

    for s in range(n):
        close(dwell[:, s], want_dwell[s], 'dwell %d' % s)
        close(roots[:, s], want_root[s], 'root %d' % s)
    for a in range(n):
        for b in range(n):
            if a != b:
                close(trans[:, a, b], want_trans[a][b]['weight'], 'transitions %d->%d' % (a, b))
    assert batch.last_chunks >= B                          # a chunk tree per chain went by


@pytest.mark.parametrize('where', ['device', 'host'])
def test_rao_teh_batch_of_different_sites_and_the_generator(ra, where):
    """One chain per site, sites with different observations: the sums over the batch
    against the batched expectation call; and the reference's generator interface
    (sparse rate matrix with labelled states) on top of a batch of one."""
    import networkx as nx
    from raoteh_amd import _mjp_dense, _sampler
    cfg = ra.synth.make_config('c1', nsites=1500)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    Q = cfg['Q_default']
    ta_index = _sampler.TreeArrays(T, root).node_to_index
    masks = np.full((1500, len(ta_index)), (1 << n) - 1, dtype=np.uint64)
    cols = [ta_index[v] for v in cfg['leaves']]
    masks[:, cols] = np.uint64(1) << cfg['leaf_states'].astype(np.uint64)
    want_d, want_i, want_t = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, root_distn=cfg['root_distn'], Q_default=Q, obs_nodes=cfg['leaves'],
        data=cfg['leaf_states'], kind='state')
    cls = _sampler.DeviceHistoryBatch if where == 'device' else _sampler.HistoryBatch
    batch = cls(T, root, Q, node_masks=masks, root_distn=cfg['root_distn'], seed=5, ctx=ra.ctx)
    burn, keep = 8, 40
    dwell = np.zeros((1500, n))
    trans = np.zeros((1500, n, n))
    for it in range(burn + keep):
        batch.sweep()
        if it >= burn:
            dwell += batch.dwell_times()
            trans += batch.transition_counts()
    dwell /= keep
    trans /= keep
    # sums over sites: independent chains, so the error adds in quadrature
    for s in range(n):
        se = np.sqrt(dwell[:, s].var(ddof=1) * 1500)
        assert abs(dwell[:, s].sum() - want_d[s]) <= 5 * se + 1e-3 * want_d[s]
    offdiag = ~np.eye(n, dtype=bool)
    se = np.sqrt(trans.var(axis=0, ddof=1) * 1500)
    assert (np.abs(trans.sum(axis=0) - want_t)[offdiag] <= (5 * se + 1e-3 * want_t + 0.05)[offdiag]).all()
    # generator interface: states labelled by strings, rates as a digraph without loops
    labels = ['A', 'C', 'G', 'T']
    Qg = nx.DiGraph()
    for a in range(n):
        for b in range(n):
            if a != b:
                Qg.add_edge(labels[a], labels[b], weight=float(Q[a, b]))
    obs = dict((leaf, {labels[int(s)]}) for leaf, s in zip(cfg['leaves'], cfg['leaf_states'][0]))
    for v in T:
        obs.setdefault(v, set(labels))
    rd = dict((labels[s], float(p)) for s, p in enumerate(cfg['root_distn']))
    total = sum(d['weight'] for _, _, d in T.edges(data=True))
    count = 0
    for h in _sampler.gen_restricted_histories(T, Qg, obs, root, root_distn=rd, nhistories=4,
                                               seed=3, ctx=ra.ctx):
        count += 1
        assert nx.is_tree(h) and set(T) <= set(h)
        assert sum(d['weight'] for _, _, d in h.edges(data=True)) == pytest.approx(total, rel=1e-12)
        for v in h:
            states = set(d['state'] for d in h[v].values())
            if v in T:
                assert len(states) == 1 and states <= obs[v]   # not an event node
            else:
                assert h.degree(v) == 2 and len(states) == 2   # a real transition
    assert count == 4
    with pytest.raises(ValueError):
        next(_sampler.gen_restricted_histories(T, Qg, obs, root, uniformization_factor=1.0))
    with pytest.raises(ValueError):
        next(_sampler.gen_restricted_histories(T, Qg, {-5: {'A'}}, root))
    impossible = dict(obs)
    impossible[cfg['leaves'][0]] = set()
    with pytest.raises(ra.pkg.StructuralZeroProb):
        next(_sampler.gen_restricted_histories(T, Qg, impossible, root))


@pytest.mark.parametrize('where', ['device', 'host'])
def test_rao_teh_stationary_law_on_a_pure_cycle_is_exact(ra, where):
    """A pure 4-cycle 0 -> 1 -> 2 -> 3 -> 0 (unit rates) on a 3-node tree whose two leaves are
    observed in state 0: a history is its root state s and the numbers of changes on the two
    edges, k_a = k_b = -s (mod 4), and the posterior -- the stationary law of a Rao-Teh sweep
    at ANY uniformization factor -- is known in closed form:
        p(s, k_a, k_b)  ~  pi_s  Poisson(k_a; t_a)  Poisson(k_b; t_b).
    Histories that differ by a full turn on an edge are separate modes; a sweep adds a turn
    only when four new virtual events fall on that edge.  Round 2's soak saw such a case 22
    standard errors off after 3 000 sweeps at factor 2 and called it slow mixing; here the
    claim is a test: at factor 2 the law of (root state, total changes) over many chains
    is (i) visibly short of the turned modes after a few sweeps on short branches and
    (ii) the exact law, by chi-square, once the chains have mixed -- on long and on short
    branches, as at factor 16."""
    from math import exp, factorial
    from raoteh_amd import _sampler
    n = 4
    Q = np.zeros((n, n))
    for i in range(n):
        Q[i, (i + 1) % n] = 1.0
    Q -= np.diag(Q.sum(axis=1))
    cls = _sampler.DeviceHistoryBatch if where == 'device' else _sampler.HistoryBatch
    C = 20000 if where == 'device' else 3000

    def exact(ta, tb, kmax=40):
        law = {}
        for s in range(n):
            for ka in range((-s) % n, kmax, n):
                for kb in range((-s) % n, kmax, n):
                    w = 0.25 * exp(-ta) * ta ** ka / factorial(ka) * exp(-tb) * tb ** kb / factorial(kb)
                    law[(s, ka + kb)] = law.get((s, ka + kb), 0.0) + w
        z = sum(law.values())
        return dict((k, v / z) for k, v in law.items())

    def observed(batch):
        root = batch.root_states()
        total = batch.transition_counts().reshape(C, -1).sum(axis=1)
        return root, total

    def chi_square(root, total, law):
        cells = sorted(law, key=lambda k: -law[k])
        stat, dof, rest_e, rest_o = 0.0, -1, 0.0, 0
        seen = 0
        for key in cells:
            e = law[key] * C
            o = int(((root == key[0]) & (total == key[1])).sum())
            seen += o
            if e >= 8.0:
                stat += (o - e) ** 2 / e
                dof += 1
            else:
                rest_e += e
                rest_o += o
        rest_o += C - seen
        if rest_e >= 8.0:
            stat += (rest_o - rest_e) ** 2 / rest_e
            dof += 1
        return stat, dof

    for (ta, tb), factor, sweeps in (((3.0, 2.0), 2.0, 400), ((0.9, 0.7), 2.0, 3000),
                                     ((0.9, 0.7), 16.0, 300)):
        if where == 'host' and sweeps > 400:
            continue                         # (the numpy-orchestrated batch: the quick case only)
        T = nx.Graph()
        T.add_edge(0, 1, weight=ta)
        T.add_edge(0, 2, weight=tb)
        law = exact(ta, tb)
        # a full turn on some edge: more changes than the 2 (-s mod 4) the root state needs
        base = lambda st: 2 * ((-st) % n)
        turned = sum(p for (st, k), p in law.items() if k >= base(st) + 4)
        b = cls(T, 0, Q, node_to_allowed_states={1: {0}, 2: {0}}, nchains=C,
                root_distn=np.full(n, 0.25), uniformization_factor=factor, seed=11, ctx=ra.ctx)
        def run(k):
            if where == 'device':
                b.sweep(k)
            else:
                for _ in range(k):
                    b.sweep()
        if (ta, factor) == (0.9, 2.0):
            run(3)
            root, total = observed(b)
            early = float((total >= 2 * ((-root) % n) + 4).mean())
            # the start-up histories have no turn; the turned modes fill up from below
            assert early <= turned + 4.0 * np.sqrt(turned / C), (early, turned)
            run(sweeps - 3)
        else:
            run(sweeps)
        root, total = observed(b)
        # structure: both edges carry -s (mod 4) changes, so the total is -2 s (mod 4)
        assert ((total + 2 * root) % n == 0).all()
        stat, dof = chi_square(root, total, law)
        # chi-square upper tail: mean dof, sd sqrt(2 dof); a biased sampler is off by hundreds
        assert stat < dof + 6.0 * np.sqrt(2.0 * dof) + 10.0, (ta, tb, factor, stat, dof)
        late = float((total >= 2 * ((-root) % n) + 4).mean())
        assert abs(late - turned) < 6.0 * np.sqrt(turned * (1 - turned) / C) + 1e-3, \
            (ta, tb, factor, late, turned)


def test_forest_trees_beyond_the_lds_image(ra):
    """Trees with more nodes than the wave-private LDS image holds (csrc/forest.hip,
    FOREST_CAP = 1024) take the coherent global path: the boolean passes and the upward
    pass against a plain numpy statement of _mcy.py:396-470, :611-682, the draws against
    the support and the root's exact posterior; small trees ride in the same batch."""
    from raoteh_amd import _forest
    rng = np.random.RandomState(17)
    n = 5
    # two states that cannot be left for each other directly: structural zeros, but every
    # state reaches every other within two steps, so sparse observations stay feasible
    Q = rng.exponential(size=(n, n)) + 0.1
    Q[0, 1] = Q[1, 0] = Q[3, 4] = 0.0
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(axis=1))
    P = np.identity(n) + Q / (2.0 * (-np.diag(Q)).max())
    nn = 1500
    big = nx.Graph()
    big.add_node(0)
    for k in range(1, nn):
        big.add_edge(int(rng.randint(max(0, k - 40), k)), k)
    small = nx.path_graph(6)
    allowed = dict((int(v), {int(rng.randint(n))}) for v in rng.choice(nn, size=25, replace=False))
    allowed[0] = {0, 1, 2}
    reps = 300
    forest = _forest.Forest([(big, 0), (small, 0)] + [(big, 0)] * (reps - 1))
    obs = [allowed, {5: {2}}] + [allowed] * (reps - 1)
    sets, pmaps = _forest.get_node_to_set_and_pmap(forest, P, obs)
    # numpy statement on the big tree
    order = list(nx.dfs_preorder_nodes(big, 0))
    par = dict((b, a) for a, b in nx.bfs_edges(big, 0))
    S = dict((v, set(allowed.get(v, range(n)))) for v in big)
    nz = P > 0
    for v in reversed(order[1:]):
        S[par[v]] &= set(s for s in range(n) if any(nz[s, t] for t in S[v]))
    for v in order[1:]:
        S[v] &= set(t for t in range(n) if any(nz[s, t] for s in S[par[v]]))
    L = dict((v, np.array([1.0 if s in S[v] else 0.0 for s in range(n)])) for v in big)
    for v in reversed(order[1:]):
        L[par[v]] = L[par[v]] * P.dot(L[v])
    for v in big:
        assert sets[0][v] == S[v]
        np.testing.assert_allclose(pmaps[0][v], L[v], rtol=1e-11, atol=0)
    distn = rng.dirichlet(np.ones(n))
    states, status = _forest.resample_states(forest, P, obs, root_distn=distn, seed=4, sweep=1,
                                             return_status=True)
    assert not status.any()
    post = distn * L[0]
    post /= post.sum()
    got = np.bincount([states[k][0] for k in range(len(states)) if k != 1], minlength=n) / float(reps)
    assert not got[post == 0].any()
    assert np.all(np.abs(got - post) <= 5 * np.sqrt(np.maximum(post * (1 - post), 1e-12) / reps) + 1e-9)
    for k in (0, 2, 7):
        d = states[k]
        for v in big:
            assert d[v] in S[v]
        for a, b in nx.bfs_edges(big, 0):
            assert P[d[a], d[b]] > 0
    assert states[1][5] == 2
    again, _ = _forest.resample_states(forest, P, obs, root_distn=distn, seed=4, sweep=1,
                                       return_status=True)
    assert again == states


def test_device_resident_histories_are_consistent_and_reproducible(ra):
    """rt_chains_*: the rows of every chain stay sorted by edge, their lengths add up to
    the branch lengths, neighbouring rows of an edge differ in state (self transitions are
    removed), transitions are ones Q allows, the node states agree with the rows, the same
    seed gives the same histories, and the statistics kernels agree with numpy on the rows."""
    from raoteh_amd import _sampler
    cfg = ra.synth.make_config('c2', nsites=700)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    Q = cfg['Q_default'].copy()
    Q[0, 3] = Q[3, 0] = 0.0                                  # a structural zero in Q
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(axis=1))
    index = _sampler.TreeArrays(T, root).node_to_index
    masks = np.full((700, len(index)), (1 << n) - 1, dtype=np.uint64)
    cols = [index[v] for v in cfg['leaves']]
    masks[:, cols] = np.uint64(1) << cfg['leaf_states'].astype(np.uint64)
    masks[::9, cols[3]] = 0b0110
    batches = [_sampler.DeviceHistoryBatch(T, root, Q, node_masks=masks,
                                           root_distn=cfg['root_distn'], seed=21, ctx=ra.ctx)
               for _ in range(2)]
    for b in batches:
        b.sweep(7)
    a, b = batches
    rows_a, rows_b = a.rows(), b.rows()
    for x, y in zip(rows_a, rows_b):
        np.testing.assert_array_equal(x, y)
    chain, edge, length, state = rows_a
    assert a.sizes()[0] == chain.shape[0] and a.sizes()[1] >= 700
    N = len(index)
    # sorted by (chain, edge); every edge of every chain present; lengths add up
    key = chain * N + edge
    assert (np.diff(key) >= 0).all()
    per_edge = np.bincount(key, weights=length, minlength=700 * N).reshape(700, N)
    np.testing.assert_allclose(per_edge[:, 1:], np.broadcast_to(a.branch[1:], (700, N - 1)),
                               rtol=1e-12)
    assert (length > 0).all() and ((state >= 0) & (state < n)).all()
    same_edge = key[1:] == key[:-1]
    assert (state[1:][same_edge] != state[:-1][same_edge]).all()
    assert (Q[state[:-1][same_edge], state[1:][same_edge]] > 0).all()
    # node states: the lower node of an edge has the state of the edge's last row, the
    # upper node that of its first row
    ns = a.node_states
    last = np.ones(chain.shape[0], dtype=bool)
    last[:-1] = ~same_edge
    first = np.ones(chain.shape[0], dtype=bool)
    first[1:] = ~same_edge
    np.testing.assert_array_equal(ns[chain[last], edge[last]], state[last])
    np.testing.assert_array_equal(ns[chain[first], a.parent[edge[first]]], state[first])
    assert ((masks[np.arange(700)[:, None], np.arange(N)[None, :]] >> ns.astype(np.uint64)) & 1).all()
    # statistics kernels against numpy on the rows
    dwell = np.bincount(chain * n + state, weights=length, minlength=700 * n).reshape(700, n)
    np.testing.assert_allclose(a.dwell_times(), dwell, rtol=1e-13)
    at = np.nonzero(same_edge)[0] + 1
    trans = np.bincount((chain[at] * n + state[at - 1]) * n + state[at],
                        minlength=700 * n * n).reshape(700, n, n)
    np.testing.assert_array_equal(a.transition_counts(), trans)
    # another seed: other histories
    c = _sampler.DeviceHistoryBatch(T, root, Q, node_masks=masks, root_distn=cfg['root_distn'],
                                    seed=22, ctx=ra.ctx)
    c.sweep(7)
    assert c.rows()[2].shape != length.shape or not np.array_equal(c.rows()[2], length)
    # a chain without a feasible history is an error at creation, as in the reference
    bad = masks.copy()
    bad[5, cols[0]] = 0
    with pytest.raises(ra.pkg.StructuralZeroProb):
        _sampler.DeviceHistoryBatch(T, root, Q, node_masks=bad, ctx=ra.ctx)


@pytest.mark.parametrize('where', ['device', 'host'])
def test_metropolis_hastings_corrects_rao_teh_to_another_process(ra, where):
    """_sampler.gen_mh_histories (:393-551) on a batch: proposals are Rao-Teh sweeps under
    Q, the target density is the trajectory log-likelihood under ANOTHER rate matrix and
    root distribution; the corrected chains must reproduce the expected history
    statistics of that other process (expectation path, pinned to the oracle).  With the
    target equal to the proposal process every proposal is accepted."""
    from raoteh_amd import _mjp_dense, _sampler
    cfg = ra.synth.make_config('c1', nsites=2)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    Q = cfg['Q_default']
    rng = np.random.RandomState(12)
    Q2 = Q * rng.uniform(0.4, 2.5, size=Q.shape)
    np.fill_diagonal(Q2, 0.0)
    Q2 -= np.diag(Q2.sum(axis=1))
    rd2 = rng.dirichlet(np.ones(n) * 3)
    allowed = dict((v, set(range(n))) for v in T)
    for leaf, s in zip(cfg['leaves'], cfg['leaf_states'][0]):
        allowed[leaf] = {int(s)}
    want_dwell, want_root, _ = _mjp_dense.get_expected_history_statistics(
        T, allowed, root, n, root_distn=rd2, Q_default=Q2)
    B = 3000
    cls = _sampler.DeviceHistoryBatch if where == 'device' else _sampler.HistoryBatch
    batch = cls(T, root, Q, node_to_allowed_states=allowed, nchains=B,
                root_distn=cfg['root_distn'], seed=31, ctx=ra.ctx)

    def same(b):
        return b.trajectory_log_likelihoods()

    def other(b):
        return _sampler.trajectory_log_likelihoods(b.dwell_times(), b.transition_counts(),
                                                   b.root_states(), Q2, rd2)

    assert batch.mh_sweep(same, cache={}).all()
    cache = {}
    dwell = np.zeros((B, n))
    roots = np.zeros((B, n))
    rate = 0.0
    burn, keep = 25, 40
    for it in range(burn + keep):
        accepted = batch.mh_sweep(other, cache=cache)
        if it >= burn:
            dwell += batch.dwell_times()
            roots[np.arange(B), batch.root_states()] += 1
            rate += accepted.mean()
    dwell /= keep
    roots /= keep
    assert 0.1 < rate / keep < 0.999                       # some proposals are rejected
    for s in range(n):
        for sample, expected in ((dwell[:, s], want_dwell[s]), (roots[:, s], want_root[s])):
            se = sample.std(ddof=1) / np.sqrt(B)
            assert abs(sample.mean() - expected) <= 5 * se + 1e-3 * max(abs(expected), 1e-2)
    # the proposal process alone does NOT give these numbers (the test has power)
    _, plain_root, _ = _mjp_dense.get_expected_history_statistics(
        T, allowed, root, n, root_distn=cfg['root_distn'], Q_default=Q)
    assert np.abs(np.asarray(plain_root) - np.asarray(want_root)).max() > 0.02
    if where == 'device':
        # a snapshot goes back exactly one sweep
        batch.snapshot()
        batch.sweep(2)
        with pytest.raises(ValueError):
            batch.restore(np.ones(B, dtype=bool))
        # the reference's generator: (history, accepted) pairs
        count = 0
        total = sum(d['weight'] for _, _, d in T.edges(data=True))
        for h, ok in _sampler.gen_mh_histories(
                T, Q, allowed, lambda tree: -0.5 * sum(d['weight'] for _, _, d in tree.edges(data=True)
                                                       if d['state'] == 0),
                root, root_distn=cfg['root_distn'], nhistories=6, seed=2, ctx=ra.ctx):
            count += 1
            assert isinstance(ok, bool)
            assert sum(d['weight'] for _, _, d in h.edges(data=True)) == pytest.approx(total, rel=1e-12)
        assert count == 6


def test_device_chains_at_the_limits(ra):
    """rt_chains_* at the edges of what it accepts: a two-node tree, one chain, 64 states
    (every lane a state, full 64-bit masks), and a base tree of 1 024 nodes (98 KB of LDS
    tables in the split kernel); one node more is an error, not a fault."""
    from raoteh_amd import _sampler
    rng = np.random.RandomState(3)
    # 64 states on an edge
    n = 64
    Q = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) < 0.2)
    for i in range(n):
        Q[i, (i + 1) % n] += 0.5
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(axis=1))
    T = nx.Graph()
    T.add_edge(7, 3, weight=0.8)
    b = _sampler.DeviceHistoryBatch(T, 7, Q, node_to_allowed_states={7: {63}, 3: {0, 5, 63}},
                                    nchains=1, seed=1, ctx=ra.ctx)
    b.sweep(5)
    chain, edge, length, state = b.rows()
    assert (chain == 0).all() and (edge == 1).all() and length.sum() == pytest.approx(0.8, rel=1e-12)
    assert state[0] == 63 and state[-1] in (0, 5, 63)
    assert (Q[state[:-1], state[1:]] > 0).all()
    # 1 024 base nodes: a caterpillar with random leaf states, 4 states
    n = 4
    Q4, pi = ra.synth.hky85()
    N = 1024
    big = nx.Graph()
    for v in range(1, N):
        big.add_edge(v - 1 if v % 2 else v - 2, v,          # spine 0, 2, 4, ...; a leaf on each
                     weight=0.02 + 0.05 * rng.uniform())
    assert nx.is_tree(big) and big.number_of_nodes() == N
    leaves = [v for v in big if big.degree(v) == 1 and v != 0]
    allowed = dict((v, {int(rng.randint(n))}) for v in leaves[::3])
    c = _sampler.DeviceHistoryBatch(big, 0, Q4, node_to_allowed_states=allowed, nchains=40,
                                    root_distn=pi, seed=2, ctx=ra.ctx)
    c.sweep(3)
    chain, edge, length, state = c.rows()
    per_edge = np.bincount(chain * N + edge, weights=length, minlength=40 * N).reshape(40, N)
    np.testing.assert_allclose(per_edge[:, 1:], np.broadcast_to(c.branch[1:], (40, N - 1)), rtol=1e-11)
    ns = c.node_states
    for v, ss in allowed.items():
        assert set(np.unique(ns[:, c.tree.node_to_index[v]]).tolist()) <= ss
    big.add_edge(0, N, weight=0.1)
    with pytest.raises(ValueError):
        _sampler.DeviceHistoryBatch(big, 0, Q4, nchains=2, ctx=ra.ctx)
    with pytest.raises(ValueError):
        _sampler.DeviceHistoryBatch(T, 7, np.zeros((3, 3)), nchains=1, ctx=ra.ctx)


def test_device_sweep_virtual_events_are_poisson_beyond_255_per_row(ra):
    """The virtual events of a sweep are a Poisson process of rate omega - q(state) on every
    segment (_sample_mjp_dense.py:47-61), however many fall on one row: with a uniformization
    factor of 400 a row of length 1 carries ~ 399 of them (round 2's kernel stopped at 255
    silently).  A row that would pass the 16-bit cap is an error, not a wrong draw; a chunk
    tree of zero likelihood in a sweep raises what the host batch raises."""
    from raoteh_amd import _sampler
    from raoteh_amd._util import StructuralZeroProb
    Q = np.array([[-1.0, 1.0], [1.0, -1.0]])
    T = nx.Graph()
    T.add_edge(0, 1, weight=1.0)
    C = 2000
    b = _sampler.DeviceHistoryBatch(T, 0, Q, nchains=C, uniformization_factor=400, seed=3,
                                    ctx=ra.ctx)
    assert b.poisson_rates == pytest.approx([399.0, 399.0])
    for _ in range(3):
        rows_before = b.sizes()[0]
        expect = float((b.dwell_times() * b.poisson_rates[None, :]).sum(axis=1).mean())
        b.sweep(1)
        events = (b.sizes()[1] - rows_before) / float(C)      # two-node tree: chunks = new rows
        assert expect == pytest.approx(399.0, rel=1e-9)
        assert abs(events - expect) < 5.0 * np.sqrt(expect / C), (events, expect)
    chain, edge, length, state = b.rows()
    np.testing.assert_allclose(np.bincount(chain, weights=length, minlength=C), 1.0, rtol=1e-11)
    # past the cap of 65 535 events per row: refused
    far = _sampler.DeviceHistoryBatch(T, 0, Q, nchains=4, uniformization_factor=70000, seed=1,
                                      ctx=ra.ctx)
    with pytest.raises(ra.lib.RaotehHipError) as err:
        far.sweep(1)
    assert 'virtual events' in str(err.value) and far.nsweeps == 0
    # root restricted to a state the data then rule out in mid-run cannot happen through the
    # constructor (it checks feasibility), so the zero-likelihood error path of a sweep is
    # reached through the C ABI: masks that exclude every state at a node after creation are
    # not expressible either -- the creation-time error is the one a caller sees
    with pytest.raises(StructuralZeroProb):
        _sampler.DeviceHistoryBatch(T, 0, np.array([[-1.0, 1.0], [0.0, 0.0]]),
                                    node_to_allowed_states={0: {1}, 1: {0}}, nchains=2,
                                    ctx=ra.ctx)


def test_codon_scale_expectation_weights_on_the_matrix_pipe(ra, monkeypatch):
    """rt_mjp_esd_expectation_weights_obs for 8 < n <= 64 (csrc/expect_mfma.hip: upward pass
    with L and M kept, downward pass with P^T fragments, site sums as GEMMs over the sites)
    against the reference-format per-pass kernels on the same input: 61-state codon model
    and a 40-state random model (three row tiles), site weights, ambiguous and impossible
    observations, a batch that is not a multiple of 16 sites; and the per-pass form is the
    one the oracle tests pin."""
    from raoteh_amd import _mjp_dense
    from raoteh_amd._tree import TreeArrays
    rng = np.random.RandomState(21)
    cfg = ra.synth.make_config('c3', nsites=203)
    cases = [(cfg['T'], cfg['root'], cfg['nstates'], cfg['leaves'], cfg['Q_default'],
              cfg['root_distn'], cfg['leaf_states'])]
    n2 = 40
    T2, root2, leaves2 = ra.synth.balanced_tree(8, seed=5)
    R = rng.exponential(size=(n2, n2)) * (rng.uniform(size=(n2, n2)) < 0.5)
    np.fill_diagonal(R, 0.0)
    Q2 = R - np.diag(R.sum(axis=1))
    cases.append((T2, root2, n2, leaves2, Q2, None, rng.randint(n2, size=(77, len(leaves2)))))
    for n3 in (9, 20, 31):                                   # one and two row tiles
        R = rng.exponential(size=(n3, n3))
        np.fill_diagonal(R, 0.0)
        cases.append((T2, root2, n3, leaves2, R - np.diag(R.sum(axis=1)), rng.dirichlet(np.ones(n3)),
                      rng.randint(n3, size=(130, len(leaves2)))))
    for T, root, n, leaves, Q, rd, states in cases:
        nb = states.shape[0]
        T_aug = _mjp_dense.get_expm_augmented_tree(T, root, Q_default=Q)
        ta = TreeArrays(T_aug, root)
        esd = ta.esd_transitions(n)
        cols = [ta.node_to_index[v] for v in leaves]
        masks = (np.uint64(1) << states.astype(np.uint64))
        masks[::7, 1] |= np.uint64(1) << np.uint64((states[::7, 1] + 1) % n)      # ambiguous
        masks[5, 0] = 0                                                           # impossible
        w = rng.uniform(0.5, 2.0, size=nb)
        got = ra.ctx.expectation_weights_obs(ta.indices, ta.indptr, esd, rd, cols, masks, 'mask',
                                             site_weights=w)
        monkeypatch.setenv('RAOTEH_EXPECT_LEGACY', '1')
        want = ra.ctx.expectation_weights_obs(ta.indices, ta.indptr, esd, rd, cols, masks, 'mask',
                                              site_weights=w)
        monkeypatch.delenv('RAOTEH_EXPECT_LEGACY')
        W, rp, st = got
        W0, rp0, st0 = want
        np.testing.assert_array_equal(st, st0)
        assert st[5] == 2 and (st != 0).sum() == 1
        scale = np.abs(W0).max()
        np.testing.assert_allclose(W, W0, rtol=1e-10, atol=1e-13 * scale)
        np.testing.assert_allclose(rp, rp0, rtol=1e-12)
        assert not W[0].any()
        # uint8 states: the same numbers as the one-hot masks
        ok = np.ones(nb, dtype=bool)
        ok[5] = False
        a = ra.ctx.expectation_weights_obs(ta.indices, ta.indptr, esd, rd, cols,
                                           states[ok][:, :].astype(np.uint8), 'state')
        b = ra.ctx.expectation_weights_obs(ta.indices, ta.indptr, esd, rd, cols,
                                           np.uint64(1) << states[ok].astype(np.uint64), 'mask')
        np.testing.assert_array_equal(a[0], b[0])
