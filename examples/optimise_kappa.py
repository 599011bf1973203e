#!/usr/bin/env python
"""
The repeated-evaluation loop the resident objects are for: maximum-likelihood
estimate of the transition/transversion ratio kappa of an HKY85 model on a fixed
tree and alignment (the shape of the reference's examples/p53/liwen-opt.py and
jeffopt.py objectives: new rate matrix -> expm of every edge -> likelihood of every
site -> sum).  Sites and tree stay on the GPU; one iteration uploads 16 doubles.

    python examples/optimise_kappa.py [nsites]

Synthetic data: configuration 2 of the benchmark (64-leaf tree, sites simulated with
kappa = 2), so the estimate should come back near 2.
"""
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize_scalar

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from raoteh_amd import synth                      # noqa: E402
from raoteh_amd.device import TreeModel           # noqa: E402


def main(argv):
    nsites = int(argv[1]) if len(argv) > 1 else 100000
    cfg = synth.make_config('c2', nsites=nsites)
    model = TreeModel(cfg['T'], cfg['root'], cfg['nstates'])
    model.set_root_distn(cfg['root_distn'])
    batch = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8),
                               kind='state')                  # once
    calls = []

    def negative_log_likelihood(kappa):
        Q, _ = synth.hky85(kappa=kappa, pi=cfg['root_distn'])
        t0 = time.perf_counter()
        model.set_rates(Q_default=Q)              # expm of every edge on the device
        total, nzero = model.total_log_likelihood(batch)     # prune + reduce
        calls.append(time.perf_counter() - t0)
        return -total

    res = minimize_scalar(negative_log_likelihood, bounds=(0.2, 10.0), method='bounded',
                          options=dict(xatol=1e-6))
    lat = np.array(calls[2:]) * 1e6
    print('%d sites: kappa_hat = %.5f (simulated with 2.0), log-likelihood %.4f, '
          '%d evaluations, %.0f us per evaluation (median, host round trip included)' % (
              nsites, res.x, -res.fun, len(calls), np.median(lat)))


if __name__ == '__main__':
    main(sys.argv)
