#!/usr/bin/env python
"""Single-site call latency of the reference-shaped API (run on a GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raoteh_amd import synth, _mjp_dense, _mcy_dense

for name in ('c1', 'c2', 'c3'):
    cfg = synth.make_config(name, nsites=4)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    allowed = synth.site_node_to_allowed_states(cfg, 0)
    T_aug = _mjp_dense.get_expm_augmented_tree(T, root, Q_default=cfg['Q_default'])
    for _ in range(3):
        _mcy_dense.get_likelihood(T_aug, root, n, node_to_allowed_states=allowed,
                                  root_distn=cfg['root_distn'])
    reps = 30
    t0 = time.perf_counter()
    for _ in range(reps):
        lk = _mcy_dense.get_likelihood(T_aug, root, n, node_to_allowed_states=allowed,
                                       root_distn=cfg['root_distn'])
    t1 = time.perf_counter()
    for _ in range(reps):
        lk2 = _mjp_dense.get_likelihood(T, allowed, root, n, root_distn=cfg['root_distn'],
                                        Q_default=cfg['Q_default'])
    t2 = time.perf_counter()
    print('%s: n=%d nodes=%d  _mcy_dense.get_likelihood %.0f us/call   _mjp_dense.get_likelihood '
          '(with expm of every edge) %.0f us/call' % (name, n, T.number_of_nodes(),
          (t1 - t0) / reps * 1e6, (t2 - t1) / reps * 1e6))
