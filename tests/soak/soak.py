#!/usr/bin/env python
"""
Randomised soak test (run on a GPU box): random trees, state counts, observation
encodings and batch sizes; every batch through the interpreter kernel and through
the tree-specialised kernel (random tiles / sites per wave), both compared bit for
bit with each other and to 1e-10 with the oracle.  Not part of the pytest suite:
    python tests/soak/soak.py [seconds] [seed]
"""
import os
import sys
import time

import networkx as nx
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raoteh_amd import _lib, device, synth           # noqa: E402
from oracle import oracle_numpy as orc               # noqa: E402  (the checker)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    rng = np.random.RandomState(seed)
    set_option = _lib.lib().rt_set_option
    ctx = device.get_context(0)
    t0 = time.time()
    cases = 0
    worst = 0.0
    while time.time() - t0 < budget:
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 20, 31, 32, 33, 40, 48, 61, 64,
                            65, 80, 97, 112, 122, 128]))
        nnodes = int(rng.randint(2, 70))
        nsites = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 200, 1000, 3000]))
        T, root, leaves = synth.random_tree(nnodes, seed=int(rng.randint(1 << 30)),
                                            max_children=int(rng.randint(2, 5)))
        for na, nb in nx.bfs_edges(T, root):
            M = rng.exponential(size=(n, n))
            if rng.uniform() < 0.3:
                M *= rng.uniform(size=(n, n)) > 0.3
                M[np.arange(n), np.arange(n)] += 0.05
            T[na][nb]['P'] = M / M.sum(axis=1, keepdims=True)
        obs_nodes = list(leaves)
        if rng.uniform() < 0.5:
            inner = [v for v in T if v not in leaves]
            obs_nodes += inner[::int(rng.randint(1, 4))]
        if rng.uniform() < 0.2 and len(obs_nodes) > 1:
            obs_nodes = obs_nodes[1:]                     # an unobserved leaf
        w = rng.uniform(0.0, 1.0, size=n)
        dense = rng.uniform(0.05, 1.0, size=(nsites, len(obs_nodes), n))
        dense[rng.uniform(size=dense.shape) < 0.15] = 0.0
        # half of the lane-family cases: observed states (the compact resident encoding
        # of the specialised kernel) against their dense 0/1 expansion
        states, skind = None, 'state'
        if n <= 4 and rng.uniform() < 0.5:
            if rng.uniform() < 0.5:
                states = rng.randint(0, n, size=(nsites, len(obs_nodes))).astype(np.uint8)
                states[rng.uniform(size=states.shape) < 0.15] = 255
                dense = np.ones((nsites, len(obs_nodes), n))
                seen = states != 255
                dense[seen] = 0.0
                ii, kk = np.nonzero(seen)
                dense[ii, kk, states[ii, kk]] = 1.0
            else:                                  # allowed-set masks
                skind = 'mask'
                states = rng.randint(0, 1 << n, size=(nsites, len(obs_nodes))).astype(np.uint64)
                dense = ((states[..., None] >> np.arange(n, dtype=np.uint64)) & 1
                         ).astype(np.float64)
        if n > 4 and rng.uniform() < 0.4:
            # observed states at every leaf and nowhere else: the kernels whose leaves are
            # gathered columns of P (one-wave 4x4x4 form; serial and pipelined split-M generators)
            obs_nodes = list(leaves)
            states = rng.randint(0, n, size=(nsites, len(obs_nodes))).astype(np.uint8)
            dense = np.zeros((nsites, len(obs_nodes), n))
            ii, kk = np.indices(states.shape)
            dense[ii, kk, states] = 1.0
            skind = 'state'
            if rng.uniform() < 0.5:
                # ... or allowed sets of one or two states per leaf, as masks
                other = rng.randint(0, n, size=states.shape)
                other[rng.uniform(size=states.shape) < 0.3] = -1
                words = (n + 63) // 64
                masks = np.zeros(states.shape + (words,), dtype=np.uint64)
                for arr in (states.astype(np.int64), other):
                    ok = arr >= 0
                    np.bitwise_or.at(masks, (ii[ok], kk[ok], arr[ok] // 64),
                                     np.uint64(1) << (arr[ok] % 64).astype(np.uint64))
                    dense[ii[ok], kk[ok], arr[ok]] = 1.0
                states = masks if words > 1 else masks[..., 0]
                skind = 'mask'
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        oidx = [pre.index(v) for v in obs_nodes]
        # the oracle on a bounded sample of the sites
        sample = np.arange(nsites) if nsites <= 64 else rng.choice(nsites, 64, replace=False)
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense[sample], w)
        only = os.environ.get('SOAK_ONLY')
        if only is not None and cases != int(only):
            for _ in range(2):
                rng.randint(1, 6); rng.choice([64, 56, 33, 5]); rng.choice(['', '0', '1'])
            cases += 1
            continue
        model = device.TreeModel(T, root, n, ctx=ctx)
        model.set_transitions(esd)
        model.set_root_distn(w)
        out = []
        # (jit, tiles, sites per wave, root halves: '' = the library's own policy; n > 32 only)
        variants = [(0, None, None, '')]
        for _ in range(2):
            variants.append((1, int(rng.randint(1, 6)), int(rng.choice([64, 56, 33, 5])),
                             str(rng.choice(['', '0', '1']))))
        if only is not None:
            variants = [(0, None, None, '')] + [(1, t, 64, h) for t in (1, 2, 3, 4, 5)
                                                for h in ('0', '1')]
            print('case', cases, dict(n=n, nnodes=nnodes, nsites=nsites, nobs=len(obs_nodes),
                  nleaves=len(leaves), depth=None))
        for jit, tiles, bs, halves in variants:
            _lib.check(set_option(b'jit', jit))
            if jit:
                os.environ['RAOTEH_JIT_TILES'] = str(tiles)
                if halves:
                    os.environ['RAOTEH_JIT_HALVES'] = halves
                    # (the combine step folded into the pruning kernel for every other cut)
                    os.environ['RAOTEH_JIT_FOLD'] = '1' if (cases + tiles) % 2 else '0'
                _lib.check(set_option(b'jit_block_sites', bs if n <= 4 else 0))
            try:
                try:
                    if jit and states is not None:
                        batch = model.upload_sites(obs_nodes, states, kind=skind)
                    else:
                        batch = model.upload_sites(obs_nodes, dense, kind='dense')
                except Exception:
                    print('UPLOAD FAILED', dict(case=cases, n=n, nnodes=nnodes, nsites=nsites,
                                                variant=(jit, tiles, bs, halves), kind=skind,
                                                root=root, obs_nodes=obs_nodes,
                                                edges=list(nx.bfs_edges(T, root))))
                    raise
                ll, st = model.log_likelihoods(batch)
                tot = model.fetch_totals(batch)
                out.append((ll, st, tot, ctx.kernel_time(1)[2], (jit, tiles, bs, halves)))
            finally:
                _lib.check(set_option(b'jit', -1))
                _lib.check(set_option(b'jit_block_sites', 0))
                os.environ.pop('RAOTEH_JIT_TILES', None)
                os.environ.pop('RAOTEH_JIT_HALVES', None)
                os.environ.pop('RAOTEH_JIT_FOLD', None)
        ref = out[0]
        if only is not None:
            ok = wst == 0
            for o in out:
                got = o[0][sample]
                err = np.max(np.abs(got[ok] - want[ok]) / np.abs(want[ok]))
                print('  variant', o[4], o[3], 'max rel err vs oracle %.3e' % err)
            return
        for o in out[1:]:
            if not (np.array_equal(ref[0], o[0]) and np.array_equal(ref[1], o[1])):
                bad = np.flatnonzero(~((ref[0] == o[0]) | (np.isnan(ref[0]) & np.isnan(o[0]))))
                print('MISMATCH interpreter vs specialised', dict(case=cases, n=n, nnodes=nnodes, nsites=nsites,
                      variant=o[4], kernels=(ref[3], o[3]), first_bad=bad[:5].tolist(),
                      a=ref[0][bad[:3]].tolist(), b=o[0][bad[:3]].tolist()))
                sys.exit(1)
            assert o[2][1] == ref[2][1] and o[2][2] == nsites
        ok = wst == 0
        got, gst = ref[0][sample], ref[1][sample]
        if not np.array_equal(gst & 1, wst):
            print('STATUS MISMATCH', dict(n=n, nnodes=nnodes, nsites=nsites))
            sys.exit(1)
        if ok.any():
            err = np.max(np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-300))
            worst = max(worst, err)
            if not err < 1e-10:
                print('ORACLE MISMATCH', dict(n=n, nnodes=nnodes, nsites=nsites, err=err,
                                              kernel=ref[3]))
                sys.exit(1)
        cases += 1
        nhalves = globals().setdefault('_nhalves', 0) + sum('halves' in o[3] for o in out)
        globals()['_nhalves'] = nhalves
        if cases % 25 == 0:
            print('%d cases, %.0f s, worst relative error vs oracle %.2e, %d root-halves kernels' % (
                cases, time.time() - t0, worst, nhalves), flush=True)
    print('soak: %d cases clean (%d root-halves kernels among them), worst relative error vs oracle %.2e' % (
        cases, globals().get('_nhalves', 0), worst))


if __name__ == '__main__':
    main()
