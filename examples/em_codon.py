#!/usr/bin/env python
"""
ECM for kappa (transition / transversion ratio) and omega (nonsynonymous / synonymous
ratio) of the MG94 codon model from expected history statistics on a RESIDENT batch --
the use the reference's get_expected_history_statistics
(raoteh/sampler/_mjp_dense.py:410-539) is made for, over a whole alignment per iteration:

  E step  model.expected_history_statistics(batch) (rt_expect_step): per-edge expm,
          upward pass, downward pass, per-edge site sums and one Frechet block
          exponential per edge on the device; the alignment was uploaded once, an
          iteration moves 2 n + n^2 numbers.
  CM steps  rates q_ij = mu * pi[target nt] * (kappa if transition) * (omega if the amino
          acid changes): with omega fixed, mu*kappa = N_ts / E_ts and mu = N_tv / E_tv
          (N: expected counts, E = sum base_ij * dwell_i, the exposure); then the same for
          omega with kappa fixed.

    python examples/em_codon.py [nsites]

Synthetic data: configuration 3 of the benchmark (61 states, 64-leaf tree, simulated with
kappa = 3.17632, omega = 0.21925).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from raoteh_amd import device, synth          # noqa: E402


def main(argv):
    nsites = int(argv[1]) if len(argv) > 1 else 10000
    cfg = synth.make_config('c3', nsites=nsites)
    T, root, n, pi = cfg['T'], cfg['root'], cfg['nstates'], cfg['root_distn']
    code = synth.genetic_code()
    nt = dict(zip('ACGT', (0.25039, 0.30126, 0.25952, 0.18883)))
    is_ts = {('A', 'G'), ('G', 'A'), ('C', 'T'), ('T', 'C')}
    base = np.zeros((n, n))          # pi[target nucleotide] where one nucleotide differs
    ts = np.zeros((n, n), dtype=bool)
    nonsyn = np.zeros((n, n), dtype=bool)
    for a, (ca, ra) in enumerate(code):
        for b, (cb, rb) in enumerate(code):
            diff = [(x, y) for x, y in zip(ca, cb) if x != y]
            if len(diff) == 1:
                base[a, b] = nt[diff[0][1]]
                ts[a, b] = diff[0] in is_ts
                nonsyn[a, b] = ra != rb
    single = base > 0

    def rate_matrix(mu, kappa, omega):
        R = mu * base * np.where(ts, kappa, 1.0) * np.where(nonsyn, omega, 1.0)
        return R - np.diag(R.sum(axis=1))

    model = device.TreeModel(T, root, n)
    model.set_root_distn(pi)
    model.set_rates(Q_default=rate_matrix(1.0, 1.0, 1.0))
    batch = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8), kind='state')
    mu, kappa, omega = 1.0, 1.0, 1.0
    for it in range(15):
        model.set_rates(Q_default=rate_matrix(mu, kappa, omega))
        t0 = time.perf_counter()
        dwell, _, trans = model.expected_history_statistics(batch, recompute_transitions=False)
        dt = time.perf_counter() - t0
        # kappa | omega
        expo = dwell[:, None] * base * np.where(nonsyn, omega, 1.0)
        mu_k = trans[single & ts].sum() / expo[single & ts].sum()
        mu = trans[single & ~ts].sum() / expo[single & ~ts].sum()
        kappa = mu_k / mu
        # omega | kappa
        expo = dwell[:, None] * base * np.where(ts, kappa, 1.0)
        mu_w = trans[single & nonsyn].sum() / expo[single & nonsyn].sum()
        mu_s = trans[single & ~nonsyn].sum() / expo[single & ~nonsyn].sum()
        omega = mu_w / mu_s
        mu = mu_s
        ll = model.total_log_likelihood(batch)[0]
        print('iteration %2d: kappa = %.5f  omega = %.5f  mu = %.5f  log-lik = %.4f  '
              '(E step over %d sites: %.2f ms)' % (it + 1, kappa, omega, mu, ll, nsites, dt * 1e3))
    print('kappa_hat = %.4f (simulated with 3.17632), omega_hat = %.4f (0.21925)' % (kappa, omega))


if __name__ == '__main__':
    main(sys.argv)
