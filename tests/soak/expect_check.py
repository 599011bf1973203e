"""Expected history statistics at batch sizes: the device path
(raoteh_amd._mjp_dense.get_expected_history_statistics_batch: batched passes + one
Frechet block exponential per edge) against the oracle's reference-faithful
restatement (scipy expm_frechet once per direction, per edge, per site) on a
sample of the sites, with wall-clock times of both.  Needs a GPU.

    python tests/soak/expect_check.py [c2|c5] [nsites] [oracle_sample]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raoteh_amd import _mjp_dense, synth          # noqa: E402
from oracle import oracle_numpy as orc             # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    nsites = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    sample = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cfg = synth.make_config(name, nsites=nsites)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    sites = []
    for row in cfg['leaf_states']:
        if cfg['obs_kind'] == 'state':
            sites.append(dict((leaf, {int(s)}) for leaf, s in zip(cfg['leaves'], row)))
        else:
            sites.append(dict((leaf, set(cfg['leaf_allowed'][int(s)]))
                              for leaf, s in zip(cfg['leaves'], row)))
    kw = dict(root_distn=cfg['root_distn'], Q_default=cfg.get('Q_default'))
    # the array form of the same batch: allowed-set bit masks per leaf
    if cfg['obs_kind'] == 'state':
        masks = 1 << cfg['leaf_states'].astype(np.int64)
    else:
        table = np.array([sum(1 << s for s in ss) for ss in cfg['leaf_allowed']],
                         dtype=np.int64)
        masks = table[cfg['leaf_states']]
    arr = dict(obs_nodes=cfg['leaves'], data=masks, kind='mask')
    _mjp_dense.get_expected_history_statistics_batch(T, root, n, sites[:2], **kw)   # warm up
    t0 = time.perf_counter()
    dwell, init, trans = _mjp_dense.get_expected_history_statistics_batch(
        T, root, n, **kw, **arr)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()            # again: the context's scratch is in place now
    dwell2, _, trans2 = _mjp_dense.get_expected_history_statistics_batch(T, root, n, **kw, **arr)
    t_gpu = time.perf_counter() - t0
    assert np.array_equal(dwell2, dwell) and np.array_equal(trans2, trans)
    print('first call %.3f s, repeated %.3f s' % (t_first, t_gpu))
    da, _, ta = _mjp_dense.get_expected_history_statistics_batch(T, root, n, sites, **kw)
    assert np.array_equal(da, dwell) and np.array_equal(ta, trans)
    total = sum(d['weight'] for _, _, d in T.edges(data=True))
    print('%s: %d sites, %d states, %d edges: device path %.3f s (%.1f sites/s); '
          'sum of dwell times / (sites * tree length) = %.15f'
          % (name, nsites, n, T.number_of_edges(), t_gpu, nsites / t_gpu,
             dwell.sum() / (nsites * total)))
    # the same sample through both
    sub = sites[:sample]
    d2, i2, t2 = _mjp_dense.get_expected_history_statistics_batch(T, root, n, sub, **kw)
    t0 = time.perf_counter()
    wd, wi, wt = np.zeros(n), np.zeros(n), np.zeros((n, n))
    for s in sub:
        full = dict((v, set(range(n))) for v in T)
        full.update(s)
        od, oi, ot = orc.mjp_dense_get_expected_history_statistics(T, full, root, n, **kw)
        wd += od
        wi += oi
        wt += ot
    t_cpu = time.perf_counter() - t0
    err = max(np.max(np.abs(d2 - wd) / np.abs(wd).max()),
              np.max(np.abs(t2 - wt) / np.abs(wt).max()),
              np.max(np.abs(i2 - wi)))
    print('oracle (reference-faithful, 1 core): %d sites in %.2f s (%.3f sites/s); '
          'worst relative difference %.2e' % (sample, t_cpu, sample / t_cpu, err))
    assert err < 1e-10


if __name__ == '__main__':
    main()
