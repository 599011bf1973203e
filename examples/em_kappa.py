#!/usr/bin/env python
"""
EM for the transition/transversion ratio kappa of an HKY85 model from expected
history statistics -- the use the reference's get_expected_history_statistics
(raoteh/sampler/_mjp_dense.py:410-539) is made for, over a whole alignment per
iteration instead of one site per call:

  (The alignment is uploaded ONCE; an iteration sets the new rates and calls
  model.expected_history_statistics(batch) -- rt_expect_step, 2 n + n^2 numbers back.  The
  codon-scale version of this loop is examples/em_codon.py.)

  E step  expected dwell time D_i per state and expected number N_ij of i -> j
          changes, summed over the sites (one device call: passes, downward pass,
          per-edge site sums, one Frechet block exponential per edge);
  M step  for rates q_ij = mu * kappa * pi_j (transitions) or mu * pi_j
          (transversions):  mu*kappa = sum_ts N_ij / sum_ts pi_j D_i,
          mu = sum_tv N_ij / sum_tv pi_j D_i.

    python examples/em_kappa.py [nsites]

Synthetic data: configuration 2 of the benchmark (64-leaf tree, sites simulated with
kappa = 2); examples/optimise_kappa.py finds the same estimate by direct
maximisation of the likelihood.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from raoteh_amd import device, synth          # noqa: E402


def main(argv):
    nsites = int(argv[1]) if len(argv) > 1 else 20000
    cfg = synth.make_config('c2', nsites=nsites)
    T, root, n, pi = cfg['T'], cfg['root'], cfg['nstates'], cfg['root_distn']
    states = cfg['leaf_states'].astype(np.uint8)
    ts = np.zeros((n, n), dtype=bool)
    for i, j in ((0, 2), (2, 0), (1, 3), (3, 1)):
        ts[i, j] = True
    tv = ~ts & ~np.eye(n, dtype=bool)

    def rate_matrix(mu, kappa):
        R = mu * np.where(ts, kappa, 1.0) * pi[None, :]
        np.fill_diagonal(R, 0.0)
        return R - np.diag(R.sum(axis=1))

    model = device.TreeModel(T, root, n)
    model.set_root_distn(pi)
    model.set_rates(Q_default=rate_matrix(1.0, 1.0))
    batch = model.upload_sites(cfg['leaves'], states, kind='state')
    mu, kappa = 1.0, 1.0
    for it in range(12):
        t0 = time.perf_counter()
        model.set_rates(Q_default=rate_matrix(mu, kappa))
        dwell, _, trans = model.expected_history_statistics(batch, recompute_transitions=False)
        dt = time.perf_counter() - t0
        exposure = dwell[:, None] * pi[None, :]
        mu_kappa = trans[ts].sum() / exposure[ts].sum()
        mu = trans[tv].sum() / exposure[tv].sum()
        kappa = mu_kappa / mu
        print('iteration %2d: kappa = %.5f  mu = %.5f  (E step over %d sites: %.1f ms)'
              % (it + 1, kappa, mu, nsites, dt * 1e3))
    print('kappa_hat = %.4f (simulated with 2.0)' % kappa)


if __name__ == '__main__':
    main(sys.argv)
