#!/usr/bin/env python
"""Full-size parity of the tree-specialised kernels (run on a GPU box): the three
families at (multi-GPU-shard-sized) batches, specialised against interpreter kernel
bit for bit, a sample of the sites against the oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raoteh_amd import _lib, device, synth       # noqa: E402
from oracle import oracle_numpy as orc           # noqa: E402  (the checker)

set_option = _lib.lib().rt_set_option
ctx = device.get_context(0)
for name, nsites in (('c2', 1000000), ('c3', 125000), ('c5', 400000)):
    cfg = synth.make_config(name, nsites=nsites)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    model = device.TreeModel(T, root, n, ctx=ctx)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    if cfg['obs_kind'] == 'state':
        data, kind = cfg['leaf_states'].astype(np.uint8), 'state'
    else:
        data, kind = synth.leaf_likelihoods(cfg), 'dense'
    out = {}
    for jit in (0, 1):
        _lib.check(set_option(b'jit', jit))
        try:
            t0 = time.time()
            batch = model.upload_sites(cfg['leaves'], data, kind=kind)
            ll, st = model.log_likelihoods(batch)
            out[jit] = (ll, st, model.fetch_totals(batch), ctx.kernel_time(1)[2], time.time() - t0)
            del batch
        finally:
            _lib.check(set_option(b'jit', -1))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), name
    m = 512
    pick = np.random.RandomState(1).choice(nsites, m, replace=False)
    dense = synth.leaf_likelihoods(dict(cfg, leaf_states=cfg['leaf_states'][pick])) \
        if kind == 'state' else data[pick]
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, root, n, Q_default=cfg['Q_default'])
    want, wst = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in cfg['leaves']],
                                          dense, cfg['root_distn'])
    err = float(np.max(np.abs(out[1][0][pick] - want) / np.abs(want)))
    assert err < 1e-10 and not out[1][1].any()
    print('%s: %d sites, %s == %s bit for bit; %d sampled sites vs oracle: max rel err %.2e; '
          'totals %.6f' % (name, nsites, out[1][3], out[0][3], m, err, out[1][2][0]), flush=True)
