#!/usr/bin/env python
"""
Posterior expectations by sampling next to the exact ones.

For every site of a simulated alignment (4-state HKY85 on a 64-leaf tree) one Rao-Teh chain
samples substitution histories given the leaf states (raoteh_amd._sampler.DeviceHistoryBatch:
the reference's gen_restricted_histories, _sampler.py:300-390, for all sites at once, histories
resident on the device); the averages of dwell times and substitution counts over sweeps and
sites are printed next to the expected history statistics of the same data
(_mjp_dense.get_expected_history_statistics_batch: the reference's :410-539 summed over sites).

    python examples/sample_histories.py [nsites=2000] [nsweeps=60]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from raoteh_amd import _mjp_dense, _sampler, synth      # noqa: E402


def main():
    nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    nsweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    cfg = synth.make_config('c2', nsites=nsites)
    T, root, n, Q = cfg['T'], cfg['root'], cfg['nstates'], cfg['Q_default']
    index = _sampler.TreeArrays(T, root).node_to_index
    masks = np.full((nsites, len(index)), (1 << n) - 1, dtype=np.uint64)
    masks[:, [index[v] for v in cfg['leaves']]] = np.uint64(1) << cfg['leaf_states'].astype(np.uint64)
    t0 = time.time()
    batch = _sampler.DeviceHistoryBatch(T, root, Q, node_masks=masks,
                                        root_distn=cfg['root_distn'], seed=1)
    burn = max(10, nsweeps // 4)
    batch.sweep(burn)
    dwell, trans = np.zeros(n), np.zeros((n, n))
    for _ in range(nsweeps):
        batch.sweep()
        dwell += batch.dwell_times().sum(axis=0)
        trans += batch.transition_counts().sum(axis=0)
    dwell /= nsweeps
    trans /= nsweeps
    t_sample = time.time() - t0
    t0 = time.time()
    want_d, _, want_t = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, root_distn=cfg['root_distn'], Q_default=Q, obs_nodes=cfg['leaves'],
        data=cfg['leaf_states'], kind='state')
    t_exact = time.time() - t0
    np.set_printoptions(precision=2, suppress=True, linewidth=120)
    print('%d sites, %d + %d sweeps: %.2f s of sampling, %.2f s for the exact expectations'
          % (nsites, burn, nsweeps, t_sample, t_exact))
    print('time spent in each state, summed over sites (sampled / exact):')
    print('  ', dwell)
    print('  ', want_d)
    print('substitutions a -> b, summed over sites (sampled / exact):')
    print(trans)
    print(np.where(np.eye(n, dtype=bool), 0.0, want_t))
    rel = np.abs(dwell - want_d).max() / want_d.max()
    print('largest relative difference of the dwell times: %.2e' % rel)


if __name__ == '__main__':
    main()
