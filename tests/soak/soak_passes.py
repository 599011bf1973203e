#!/usr/bin/env python
"""
Randomised soak test of the remaining device entry points (run on a GPU box):
 * rt_expm against scipy's expm (the oracle's custom_expm) over all Pade degrees,
   squarings and matrix orders 1..62, and rt_model_set_rates on trees;
 * the reference-format passes (pset / set / pmap with and without observation
   likelihoods / distn / joint) against the oracle, bit-exact for the masks;
 * the batched path with state (uint8) and mask (uint64) observation encodings.
    python tests/soak/soak_passes.py [seconds] [seed]
"""
import os
import sys
import time

import networkx as nx
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raoteh_amd import device, synth            # noqa: E402
from oracle import oracle_numpy as orc          # noqa: E402  (the checker)


def random_rate_matrix(rng, n, scale):
    Q = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) < rng.uniform(0.3, 1.0))
    np.fill_diagonal(Q, 0.0)
    Q *= scale / max(Q.sum(axis=1).max(), 1e-300)
    Q -= np.diag(Q.sum(axis=1))
    return Q


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
    ctx = device.get_context(0)
    t0 = time.time()
    cases = 0
    worst = dict(expm=0.0, pmap=0.0, distn=0.0, joint=0.0, batch=0.0)
    degrees = set()
    while time.time() - t0 < budget:
        # ---- expm --------------------------------------------------------------
        n = int(rng.randint(1, 63))
        count = int(rng.randint(1, 9))
        scale = float(10 ** rng.uniform(-3, 1.6))           # ||Qt|| from 1e-3 to ~40
        Q = np.stack([random_rate_matrix(rng, n, scale) for _ in range(count)])
        t = rng.uniform(0.2, 1.0, size=count)
        P, info = ctx.expm(Q, t, return_info=True)
        for k in range(count):
            want = orc.custom_expm(Q[k], t[k])
            err = np.abs(P[k] - want).max() / max(1.0, np.abs(want).max())
            worst['expm'] = max(worst['expm'], err)
            degrees.add((int(info[k, 0]), min(int(info[k, 1]), 3)))
            if not err < 1e-10:
                print('EXPM MISMATCH', dict(n=n, scale=scale, t=t[k], err=err, info=info[k]))
                sys.exit(1)
        # ---- reference-format passes -------------------------------------------
        n = int(rng.choice([1, 2, 3, 4, 5, 9, 16, 20, 33, 61, 64]))
        nnodes = int(rng.randint(2, 40))
        nsites = int(rng.choice([1, 2, 7, 33]))
        T, root, leaves = synth.random_tree(nnodes, seed=int(rng.randint(1 << 30)),
                                            max_children=int(rng.randint(2, 5)))
        for na, nb in nx.bfs_edges(T, root):
            M = rng.exponential(size=(n, n)) * (rng.uniform(size=(n, n)) > rng.uniform(0, 0.6))
            M[np.arange(n), np.arange(n)] += 0.05
            T[na][nb]['P'] = M / M.sum(axis=1, keepdims=True)
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        mask = (rng.uniform(size=(nsites, len(pre), n)) > 0.25).astype(np.int64)
        mask[:, :, 0] |= (mask.sum(axis=2) == 0)               # never an empty node
        want_mask = mask.copy()
        got_mask = mask.copy()
        for i in range(nsites):
            orc.mcy_esd_get_node_to_pset(idx, ptr, esd, want_mask[i])
        ctx.node_to_pset(idx, ptr, esd, got_mask)
        assert np.array_equal(want_mask, got_mask), 'pset'
        for i in range(nsites):
            orc.esd_get_node_to_set(idx, ptr, esd, want_mask[i])
        ctx.node_to_set(idx, ptr, esd, got_mask)
        assert np.array_equal(want_mask, got_mask), 'set'
        obs = rng.uniform(0.1, 1.0, size=mask.shape) if rng.uniform() < 0.5 else None
        pm = np.empty(mask.shape)
        ctx.node_to_pmap(idx, ptr, esd, got_mask, pm, obs_likelihood=obs)
        w = rng.uniform(0.05, 1.0, size=n)
        for i in range(nsites):
            want_pm = np.empty((len(pre), n))
            orc.mcy_esd_get_node_to_pmap(idx, ptr, esd, want_mask[i], want_pm,
                                         **({} if obs is None else dict(obs_lik=obs[i])))
            err = np.abs(pm[i] - want_pm).max() / max(np.abs(want_pm).max(), 1e-300)
            worst['pmap'] = max(worst['pmap'], err)
            assert err < 1e-11, ('pmap', err)
            if not (want_pm[0] * w).sum() > 0:
                continue
            want_dn = orc.mc0_esd_get_node_to_distn(idx, ptr, esd, w, want_pm)
            want_jt = orc.mc0_esd_get_joint_endpoint_distn(idx, ptr, esd, want_pm, want_dn)
            dn, st = ctx.node_to_distn(idx, ptr, esd, w, pm[i])
            jt = ctx.joint_endpoint_distn(idx, ptr, esd, pm[i], dn)
            e1 = np.abs(dn - want_dn).max()
            e2 = np.abs(jt - want_jt).max()
            worst['distn'] = max(worst['distn'], e1)
            worst['joint'] = max(worst['joint'], e2)
            assert e1 < 1e-11 and e2 < 1e-11, ('distn/joint', e1, e2)
        # ---- batched path, compact encodings -----------------------------------
        nsb = int(rng.choice([1, 64, 130, 2000]))
        model = device.TreeModel(T, root, n, ctx=ctx)
        model.set_transitions(esd)
        model.set_root_distn(w)
        obs_nodes = list(leaves)
        oidx = [pre.index(v) for v in obs_nodes]
        if n <= 64 and rng.uniform() < 0.5:
            bits = rng.randint(1, 1 << min(n, 30), size=(nsb, len(obs_nodes))).astype(np.uint64)
            data, kind = bits, 'mask'
            dense = ((bits[..., None] >> np.arange(n, dtype=np.uint64)) & 1).astype(np.float64)
        else:
            states = rng.randint(0, n, size=(nsb, len(obs_nodes))).astype(np.uint8)
            states[rng.uniform(size=states.shape) < 0.2] = 255
            data, kind = states, 'state'
            dense = np.ones((nsb, len(obs_nodes), n))
            seen = states != 255
            dense[seen] = 0.0
            ii, kk = np.nonzero(seen)
            dense[ii, kk, states[ii, kk]] = 1.0
        sample = np.arange(nsb) if nsb <= 64 else rng.choice(nsb, 64, replace=False)
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense[sample], w)
        ll, st = model.log_likelihoods(model.upload_sites(obs_nodes, data, kind=kind))
        assert np.array_equal(st[sample] & 1, wst), 'status'
        ok = wst == 0
        if ok.any():
            err = np.max(np.abs(ll[sample][ok] - want[ok]) / np.abs(want[ok]))
            worst['batch'] = max(worst['batch'], err)
            assert err < 1e-10, ('batch', kind, err)
        cases += 1
        if cases % 50 == 0:
            print('%d cases, %.0f s, worst %s' % (cases, time.time() - t0, worst), flush=True)
    print('soak_passes: %d cases clean; Pade (degree, squarings<=3) seen: %s; worst %s' % (
        cases, sorted(degrees), worst))


if __name__ == '__main__':
    main()
