#!/usr/bin/env python
"""
Time Rao-Teh sweeps of a batch of chains (raoteh_amd/_sampler.py):
    python tools/bench_sweep.py [workload=c2] [nchains=10000] [nsweeps=10] [device|host]
Prints, per sweep: wall time, time inside the device call (upload + three kernels +
download, rt_forest_resample_states_parents), chunks per chain, chain-sweeps per second.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raoteh_amd import _sampler, device, synth          # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    nchains = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    nsweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    cfg = synth.make_config(name, nsites=nchains)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    ctx = device.get_context(0)
    index = _sampler.TreeArrays(T, root).node_to_index
    masks = np.full((nchains, len(index)), (1 << n) - 1, dtype=np.uint64)
    cols = [index[v] for v in cfg['leaves']]
    if cfg['obs_kind'] == 'state':
        masks[:, cols] = np.uint64(1) << cfg['leaf_states'].astype(np.uint64)
    else:
        table = np.array([sum(1 << x for x in ss) for ss in cfg['leaf_allowed']], dtype=np.uint64)
        masks[:, cols] = table[cfg['leaf_states']]
    where = sys.argv[4] if len(sys.argv) > 4 else 'device'
    t0 = time.perf_counter()
    if where == 'device':
        batch = _sampler.DeviceHistoryBatch(T, root, cfg['Q_default'], node_masks=masks,
                                            root_distn=cfg['root_distn'], seed=1, ctx=ctx)
        t_init = time.perf_counter() - t0
        batch.sweep(3)
        ctx.sync()
        t = time.perf_counter()
        batch.sweep(nsweeps)
        ctx.sync()
        wall = (time.perf_counter() - t) / nsweeps
        rows, chunks, _ = batch.sizes()
        print(json.dumps(dict(
            workload=name, where='device', nchains=nchains, nstates=n, base_nodes=len(index),
            init_s=round(t_init, 3), sweep_ms=round(wall * 1e3, 3),
            chunks_per_chain=round(chunks / nchains, 1),
            segments_per_chain=round(rows / nchains, 1),
            chain_sweeps_per_s=round(nchains / wall, 1),
            chunk_nodes_per_s=round(chunks / wall, 1))))
        return
    batch = _sampler.HistoryBatch(T, root, cfg['Q_default'], node_masks=masks,
                                  root_distn=cfg['root_distn'], seed=1, ctx=ctx)
    t_init = time.perf_counter() - t0
    inner = []
    orig = batch._resample

    def timed(chain, edge, length):
        t = time.perf_counter()
        out = orig(chain, edge, length)
        inner.append(time.perf_counter() - t)
        return out

    batch._resample = timed
    for _ in range(3):
        batch.sweep()
    inner.clear()
    walls, chunks = [], []
    batch.device_seconds = 0.0
    for _ in range(nsweeps):
        t = time.perf_counter()
        batch.sweep()
        walls.append(time.perf_counter() - t)
        chunks.append(batch.last_chunks)
    wall, dev = float(np.median(walls)), float(np.median(inner))
    call = batch.device_seconds / nsweeps
    print(json.dumps(dict(
        workload=name, where='host', nchains=nchains, nstates=n, base_nodes=len(index),
        init_s=round(t_init, 3), sweep_ms=round(wall * 1e3, 2),
        chunk_trees_and_device_call_ms=round(dev * 1e3, 2),
        device_call_ms=round(call * 1e3, 2),
        chunks_per_chain=round(float(np.mean(chunks)) / nchains, 1),
        segments_per_chain=round(batch.chain.shape[0] / nchains, 1),
        chain_sweeps_per_s=round(nchains / wall, 1),
        chunk_nodes_per_s=round(float(np.mean(chunks)) / wall, 1))))


if __name__ == '__main__':
    main()
