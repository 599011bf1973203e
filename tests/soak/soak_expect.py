#!/usr/bin/env python
"""
Randomised soak test of the expected-history-statistics path (run on a GPU box):
random trees, state counts up to 64, per-edge rate matrices with structural zeros,
allowed-state sets at random nodes, site weights.  Per case
  * the device site sums (rt_mjp_esd_expectation_weights_obs) against J / P assembled
    on the host from the reference-format joint endpoint distributions;
  * dwell times summing to (sum of weights) x (tree length) -- exercises the Frechet
    block exponentials at every order;
  * for small cases, every statistic against the oracle's reference-faithful
    restatement (scipy expm_frechet per direction, edge and site).
    python tests/soak/soak_expect.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raoteh_amd import _mjp_dense, device, synth     # noqa: E402
from raoteh_amd._tree import TreeArrays              # noqa: E402
from oracle import oracle_numpy as orc               # noqa: E402  (the checker)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2468)
    ctx = device.get_context(0)
    t0 = time.time()
    cases = oracle_cases = 0
    worst_w = worst_o = worst_len = 0.0
    while time.time() - t0 < budget:
        # (n > 32: the matrix-pipe passes of csrc/expect_mfma.hip, three and four row tiles)
        n = int(rng.choice([2, 3, 4, 5, 6, 7, 9, 11, 12, 16, 20, 31, 33, 40, 48, 49, 61, 64]))
        nnodes = int(rng.randint(2, 24))
        nsites = int(rng.choice([1, 2, 3, 63, 64, 65, 130, 700]))
        T, root, leaves = synth.random_tree(nnodes, seed=int(rng.randint(1 << 30)),
                                            max_children=int(rng.randint(2, 5)))

        def rates():
            R = rng.exponential(size=(n, n))
            R[rng.uniform(size=(n, n)) < 0.25] = 0.0
            np.fill_diagonal(R, 0.0)
            for i in range(n):
                if R[i, (i + 1) % n] == 0:
                    R[i, (i + 1) % n] = rng.uniform(0.2, 1.0)
            return R - np.diag(R.sum(axis=1))
        mats = [rates(), rates()]
        for na, nb in T.edges():
            T[na][nb]['weight'] = float(rng.uniform(0.02, 1.5))
            if rng.uniform() < 0.3:
                T[na][nb]['Q'] = mats[1]
        obs_nodes = [v for v in T if T.degree(v) == 1 or rng.uniform() < 0.2]
        full = (1 << n) - 1
        data = np.full((nsites, len(obs_nodes)), full, dtype=np.uint64)
        for k in range(nsites):
            for j in range(len(obs_nodes)):
                if rng.uniform() < 0.8:
                    pick = rng.permutation(n)[:int(rng.randint(1, min(n, 3) + 1))]
                    data[k, j] = int(sum(1 << int(x) for x in pick))
        distn = rng.exponential(size=n)
        distn /= distn.sum()
        w = rng.uniform(0.5, 3.0, size=nsites)
        kw = dict(root_distn=distn, Q_default=mats[0])
        dwell, init, trans = _mjp_dense.get_expected_history_statistics_batch(
            T, root, n, weights=w, obs_nodes=obs_nodes, data=data, kind='mask', **kw)
        length = sum(d['weight'] for _, _, d in T.edges(data=True))
        worst_len = max(worst_len, abs(dwell.sum() / (w.sum() * length) - 1.0))
        assert worst_len < 1e-10, (n, nnodes, nsites, worst_len)
        assert abs(init.sum() / w.sum() - 1.0) < 1e-12

        # site sums against the reference-format passes
        T_aug = _mjp_dense.get_expm_augmented_tree(T, root, Q_default=mats[0])
        ta = TreeArrays(T_aug, root)
        esd = ta.esd_transitions(n)
        cols = [ta.node_to_index[v] for v in obs_nodes]
        mask = np.ones((nsites, ta.nnodes, n), dtype=np.int64)
        mask[:, cols, :] = ((data[:, :, None] >> np.arange(n, dtype=np.uint64)) & np.uint64(1)).astype(np.int64)
        W, rp, st = ctx.expectation_weights_obs(ta.indices, ta.indptr, esd, distn, cols, data,
                                                'mask', site_weights=w)
        assert not st.any()
        pm = np.empty(mask.shape)
        ctx.passes(ta.indices, ta.indptr, esd, mask.copy(), pm)
        dn, _ = ctx.node_to_distn(ta.indices, ta.indptr, esd, distn, pm)
        J = ctx.joint_endpoint_distn(ta.indices, ta.indptr, esd, pm, dn)
        for i in range(1, ta.nnodes):
            ratio = np.where(J[:, i] != 0, J[:, i] / np.where(esd[i] != 0, esd[i], 1.0), 0.0)
            want = np.tensordot(w, ratio, axes=(0, 0))
            err = np.max(np.abs(W[i] - want)) / max(np.abs(want).max(), 1e-300)
            worst_w = max(worst_w, err)
            assert err < 1e-11, (n, nnodes, nsites, i, err)

        if n <= 6 and nsites <= 3:
            wd, wi, wt = np.zeros(n), np.zeros(n), np.zeros((n, n))
            for k in range(nsites):
                allowed = dict((v, set(range(n))) for v in T)
                for j, v in enumerate(obs_nodes):
                    allowed[v] = set(s for s in range(n) if (data[k, j] >> s) & 1)
                od, oi, ot = orc.mjp_dense_get_expected_history_statistics(
                    T, allowed, root, n, **kw)
                wd += w[k] * od
                wi += w[k] * oi
                wt += w[k] * ot
            scale = np.abs(wd).max()
            err = max(np.max(np.abs(dwell - wd)) / scale, np.max(np.abs(trans - wt)) / scale,
                      np.max(np.abs(init - wi)))
            worst_o = max(worst_o, err)
            assert err < 1e-10, (n, nnodes, nsites, err)
            oracle_cases += 1
        cases += 1
        if cases % 25 == 0:
            print('%d cases (%d against the oracle), %d s: worst site-sum difference %.2e, '
                  'oracle difference %.2e, dwell-length defect %.2e'
                  % (cases, oracle_cases, time.time() - t0, worst_w, worst_o, worst_len),
                  flush=True)
    print('soak_expect: %d cases clean (%d against the oracle): site sums %.2e, oracle %.2e, '
          'dwell-length %.2e' % (cases, oracle_cases, worst_w, worst_o, worst_len))


if __name__ == '__main__':
    main()
