#!/usr/bin/env python
"""C4-sized single-GPU check: one GPU's share of the 1M-site codon workload
(125 000 sites, 61 states, 64 leaves) + upload (PCIe-inclusive) timing."""
import json, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raoteh_amd import synth, device
from oracle import oracle_numpy as orc

out = {}
for name, nsites in (('c2', 100000), ('c3', 125000)):
    cfg = synth.make_config(name, nsites=nsites)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    model = device.TreeModel(T, root, n)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    states = cfg['leaf_states'].astype(np.uint8)
    dense = synth.leaf_likelihoods(cfg)
    model.ctx.sync()
    t0 = time.perf_counter(); b_dense = model.upload_sites(cfg['leaves'], dense, kind='dense'); model.ctx.sync(); t_dense = time.perf_counter() - t0
    t0 = time.perf_counter(); b_state = model.upload_sites(cfg['leaves'], states, kind='state'); model.ctx.sync(); t_state = time.perf_counter() - t0
    for _ in range(3): model.prune(b_dense)
    model.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10): model.prune(b_dense)
    model.ctx.sync(); t_prune = (time.perf_counter() - t0) / 10
    ll, st = model.fetch_log_likelihoods(b_dense)
    ll2, _ = model.log_likelihoods(b_state)
    assert np.array_equal(ll, ll2) and not st.any()
    m = 2000
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(T, root, n, Q_default=cfg['Q_default'])
    want, _ = orc.batch_log_likelihoods(idx, ptr, esd, [pre.index(v) for v in cfg['leaves']], dense[:m], cfg['root_distn'])
    err = float(np.max(np.abs(ll[:m] - want) / np.abs(want)))
    out[name] = dict(nsites=nsites, dense_bytes=int(dense.nbytes), upload_dense_s=t_dense, upload_state_s=t_state,
                     prune_s=t_prune, sites_per_s_resident=nsites / t_prune,
                     sites_per_s_pcie_dense=nsites / (t_dense + t_prune), sites_per_s_pcie_state=nsites / (t_state + t_prune),
                     max_rel_err=err)
print(json.dumps(out, indent=1))
