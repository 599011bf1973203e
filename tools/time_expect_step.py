"""Diagnostics: seconds per rt_expect_step call (resident batch) for one bench configuration;
under `rocprofv3 --kernel-trace --stats` for the kernel split.
    python tools/time_expect_step.py [c3|c5|c2] [calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raoteh_amd import device, synth
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = synth.make_config(name)
T, root, n = cfg['T'], cfg['root'], cfg['nstates']
model = device.TreeModel(T, root, n)
model.set_root_distn(cfg['root_distn'])
if cfg.get('Q_default') is not None:
    model.set_rates(Q_default=cfg['Q_default'])
else:                                       # per-edge rate matrices on the tree (C5)
    model.set_rates()
batch = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
out = None
for _ in range(3):
    out = model.expected_history_statistics(batch)
t0 = time.perf_counter()
for _ in range(calls):
    out = model.expected_history_statistics(batch)
dt = (time.perf_counter() - t0) / calls
print('%s: %.3f ms per call; dwell sum %.12g, trans sum %.12g' % (name, dt * 1e3, out[0].sum(), out[2].sum()))
