#!/usr/bin/env python
"""
PCIe-inclusive cost of one pass: rt_sites_create from host buffers
(upload + pack, host data already built) + one step + the fetch of the totals, for the
dense and the state encodings of C3 and C2.  Run on the GPU box:
    python tools/time_upload.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raoteh_amd import device, synth          # noqa: E402


def main():
    ctx = device.get_context(0)
    ctx.set_option('jit_async', 1)
    for name, nsites in (('c3', 10000), ('c2', 100000)):
        cfg = synth.make_config(name, nsites=nsites)
        T, root, n = cfg['T'], cfg['root'], cfg['nstates']
        model = device.TreeModel(T, root, n, ctx=ctx)
        model.set_rates(Q_default=cfg['Q_default'])
        model.set_root_distn(cfg['root_distn'])
        dense = np.ascontiguousarray(synth.leaf_likelihoods(cfg))
        states = np.ascontiguousarray(cfg['leaf_states'].astype(np.uint8))
        for kind, data in (('dense', dense), ('state', states)):
            best_up, best_all = 1e9, 1e9
            for rep in range(5):
                t0 = time.perf_counter()
                b = model.upload_sites(cfg['leaves'], data, kind=kind)
                ctx.sync()
                t1 = time.perf_counter()
                model.step(b)
                tot = model.fetch_totals(b)
                t2 = time.perf_counter()
                b.close()
                if rep:
                    best_up = min(best_up, t1 - t0)
                    best_all = min(best_all, t2 - t0)
            print('%s %-5s %7.1f MB: upload + pack %.2f ms, with one step and the totals %.2f ms '
                  '= %.3g sites/s (loglik %.6f)'
                  % (name, kind, data.nbytes / 1e6, best_up * 1e3, best_all * 1e3,
                     nsites / best_all, tot[0]))
        model.close()


if __name__ == '__main__':
    main()
