#!/usr/bin/env python
"""Latency of the reference-shaped single-site call
raoteh_amd._mjp_dense.get_expected_history_statistics (the drop-in for
raoteh/sampler/_mjp_dense.py:410-539) on the benchmark trees.  Needs a GPU.

    python tools/latency_expect.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from raoteh_amd import _mjp_dense, synth          # noqa: E402


def main():
    for name in ('c1', 'c2', 'c5'):
        cfg = synth.make_config(name, nsites=1)
        T, root, n = cfg['T'], cfg['root'], cfg['nstates']
        allowed = synth.site_node_to_allowed_states(cfg, 0)
        kw = dict(root_distn=cfg['root_distn'], Q_default=cfg.get('Q_default'))
        _mjp_dense.get_expected_history_statistics(T, allowed, root, n, **kw)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            _mjp_dense.get_expected_history_statistics(T, allowed, root, n, **kw)
            ts.append(time.perf_counter() - t0)
        print('%s: %d states, %d edges: single-site get_expected_history_statistics '
              '%.2f ms (median of 20)' % (name, n, T.number_of_edges(), 1e3 * np.median(ts)))


if __name__ == '__main__':
    main()
