#!/usr/bin/env python
"""
Diagnosis of the wrong-result incident of round 1 (a tree-specialised MFMA kernel that
SPILLS: 32 states, 4 site tiles per wave, amdgpu_waves_per_eu(1,1), -amdgpu-mfma-vgpr-form=1;
commit 2e12c9b): run that configuration on a GPU box with the scratch rejection of
rt_jit_get overridden (RAOTEH_JIT_ALLOW_SCRATCH=1) in several variants and compare every
one bit for bit with the interpreter kernel and to 1e-10 with the oracle:

    base        as generated (asm pins + sched_barrier + vgpr-form, waves_per_eu(1,1))
    no_pins     without the empty asm volatile("" : "+v"(ptr...) : "v"(dep)) pins
    no_sb       without __builtin_amdgcn_sched_barrier(0)
    no_vform    without -mllvm -amdgpu-mfma-vgpr-form=1
    weu2        amdgpu_waves_per_eu(2,2): 256 registers, no AGPRs, far more spills

One process per variant (the generator reads its switches from the environment and the
kernel cache is keyed by source).  Not part of the pytest suite:
    python tests/soak/spill_probe.py            # driver: runs every variant
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# every variant but the last two runs the OLD generator output (root weights loaded at
# kernel start, RAOTEH_JIT_W_AT_START) with the scratch rejection and the probe-batch
# verification of rt_sites_create switched off
OLD = {'RAOTEH_JIT_W_AT_START': '1', 'RAOTEH_JIT_ALLOW_SCRATCH': '1',
       'RAOTEH_JIT_NO_VERIFY': '1'}
VARIANTS = {
    'base': {},
    'no_pins': {'RAOTEH_JIT_NO_PINS': '1'},
    'no_sb': {'RAOTEH_JIT_NO_SCHED_BARRIER': '1'},
    'no_pins_no_sb': {'RAOTEH_JIT_NO_PINS': '1', 'RAOTEH_JIT_NO_SCHED_BARRIER': '1'},
    'no_vform': {'RAOTEH_JIT_NO_VGPR_FORM': '1'},
    'weu2': {'RAOTEH_JIT_WAVES_EU': '2, 2'},
    't3': {'RAOTEH_JIT_TILES': '3'},
    # the old generator output with the verification ON: the wrong kernel must be caught
    # (the batch then runs the interpreter kernel) -- spills allowed so that only the
    # verification stands between the kernel and the user
    'old_verified': {'RAOTEH_JIT_NO_VERIFY': ''},
    # what ships: new generator (weights loaded at the root step), rejection + verification
    'shipping': {'RAOTEH_JIT_W_AT_START': '', 'RAOTEH_JIT_ALLOW_SCRATCH': '',
                 'RAOTEH_JIT_NO_VERIFY': ''},
}


def worker(name):
    import networkx as nx
    from raoteh_amd import _lib, device, synth
    from oracle import oracle_numpy as orc           # the checker
    ctx = device.get_context(0)
    out = dict(variant=name, cases=[])
    for seed, (n, nnodes, nsites) in enumerate([(32, 63, 3000), (32, 40, 1000), (31, 69, 200),
                                                (32, 20, 65), (28, 50, 1000)]):
        rng = np.random.RandomState(100 + seed)
        T, root, leaves = synth.random_tree(nnodes, seed=seed, max_children=3)
        for na, nb in nx.bfs_edges(T, root):
            M = rng.exponential(size=(n, n))
            T[na][nb]['P'] = M / M.sum(axis=1, keepdims=True)
        obs_nodes = list(leaves)
        w = rng.uniform(0.0, 1.0, size=n)
        dense = rng.uniform(0.05, 1.0, size=(nsites, len(obs_nodes), n))
        dense[rng.uniform(size=dense.shape) < 0.15] = 0.0
        pre, idx, ptr, esd = orc.get_esd_transitions(T, root, n)
        oidx = [pre.index(v) for v in obs_nodes]
        want, wst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense[:64], w)
        model = device.TreeModel(T, root, n, ctx=ctx)
        model.set_transitions(esd)
        model.set_root_distn(w)
        ctx.set_option('jit', 0)
        b0 = model.upload_sites(obs_nodes, dense, kind='dense')
        ll0, st0 = model.log_likelihoods(b0)
        ctx.set_option('jit', 1)
        b1 = model.upload_sites(obs_nodes, dense, kind='dense')
        ll1, st1 = model.log_likelihoods(b1)
        ok = np.isfinite(want)
        case = dict(n=n, nnodes=nnodes, nsites=nsites, kernel=b1.kernel_name,
                    interpreter=b0.kernel_name,
                    interp_vs_oracle=float(np.max(np.abs(ll0[:64][ok] - want[ok]) / np.abs(want[ok]))),
                    jit_vs_oracle=float(np.max(np.abs(ll1[:64][ok] - want[ok]) / np.abs(want[ok]))),
                    bit_identical=bool(np.array_equal(ll0, ll1) and np.array_equal(st0, st1)),
                    sites_differing=int(np.sum(ll0 != ll1)),
                    last_error=_lib.last_error())
        out['cases'].append(case)
        b0.close()
        b1.close()
        model.close()
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1:
        return worker(sys.argv[1])
    results = []
    for name, env_extra in VARIANTS.items():
        env = dict(os.environ)
        env.update(OLD)
        env.update(env_extra)
        for k in [k for k, v in env.items() if v == '']:
            del env[k]
        env.setdefault('RAOTEH_JIT_TILES', '4')
        p = subprocess.run([sys.executable, os.path.abspath(__file__), name], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        line = p.stdout.decode().strip().splitlines()
        if p.returncode != 0 or not line:
            results.append(dict(variant=name, error=p.stderr.decode()[-1500:]))
        else:
            results.append(json.loads(line[-1]))
        print(json.dumps(results[-1]), flush=True)
    path = os.path.join(ROOT, 'gpurun_out', 'spill_probe.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(results, open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
