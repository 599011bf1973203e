#!/usr/bin/env python
"""
bench.py -- site log-likelihoods / second of the batched tree-pruning hot path.

One *step* = one pass of the hot path over one resident batch of synthetic
sites: per-edge expm(Q*t) for every edge of the tree (from rates resident on
the device) -> fragment repack -> Felsenstein upward pass + root reduce + log
-> batch sum (-> RCCL all-reduce of the three totals when N > 1).  Inputs are in
HBM before the timed region starts.  Batches rotate through several HBM copies
so the 256 MiB Infinity Cache cannot hold the working set.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5|c4|c6]
                    [--also c2,c5,c4]

The headline (`value`, `roofline`, `cpu_baseline` at the top level) is the
61-state codon configuration C3 (BASELINE.json configs[2]: the largest
single-GPU configuration and the one the >= 100x target is stated on): 10 000
sites per GPU, weak scaling -- at N > 1 the global batch has N x 10 000 sites
(one seed) and rank r uploads dist.shard_range(...) of it.  The other
configurations ride in the same JSON line under "workloads" (each with its own
value / ms_per_step / roofline / cpu_baseline): C2 and C5 the same way, and C4
as STRONG scaling: the one 1 000 000-site codon batch (seed 3), rank r uploads
shard_range(1 000 000, r, N) -- at N = 1 the whole million on one GPU.

N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ...`
(one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the environment).  The
worker processes do not import torch: the control plane (barrier, max over
ranks) is a few bytes over a local TCP socket, the data-path reduce is
ncclAllReduce (RCCL) inside libraoteh_hip.so.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_DATASHEET_TFLOPS = 78.6    # MI355X datasheet dense FP64 matrix
ROOFLINE_LAUNCHES = 32         # event-timed launches per kernel after the timed region
SPINUP_SECONDS = 0.3        # untimed load before the warm-up steps (clock ramp of an idle box)


def f64_mfma_peak():
    """(TFLOP/s, source): the rate measured on an MI355X by tools/micro/mfma_f64_peak.hip
    (profiles/r02_mfma_f64_peak.json) when that file is present, else the datasheet."""
    path = os.path.join(ROOT, 'profiles', 'r02_mfma_f64_peak.json')
    try:
        v = float(json.load(open(path))['f64_mfma_16x16x4_peak_tflops'])
        if v > 0:
            return v, 'measured: profiles/r02_mfma_f64_peak.json (tools/micro/mfma_f64_peak.hip)'
    except Exception:
        pass
    return F64_MFMA_DATASHEET_TFLOPS, 'MI355X datasheet'


WORKLOADS = {
    'c2': dict(desc='4-state HKY85, 64-leaf balanced tree, 100000 sites/GPU, '
                    'dense f64 leaf likelihood vectors', bound='hbm', scaling='weak'),
    'c3': dict(desc='61-state MG94 codon, 64-leaf balanced tree, 10000 sites/GPU, '
                    'dense f64 leaf likelihood vectors', bound='mfma', scaling='weak'),
    'c4': dict(desc='61-state MG94 codon, 64-leaf balanced tree, ONE batch of 1000000 sites '
                    'sharded over the GPUs (dist.shard_range), dense f64 leaf likelihood '
                    'vectors', bound='mfma', scaling='strong'),
    'c5': dict(desc='20-state blinking compound process, 32-leaf tree, per-edge Q, '
                    '50000 sites/GPU, dense f64 0/1 leaf masks', bound='hbm', scaling='weak'),
    # not a BASELINE.json configuration: the 122-state space of examples/p53/liwen.py:599-621
    # (the caller SURVEY 8b lists that needs more than 64 states) at the shape of C3
    'c6': dict(desc='122-state switching model (MG94 x {reference, default}, '
                    'examples/p53/liwen.py:599-621), 64-leaf balanced tree, 10000 sites/GPU, '
                    'dense f64 0/1 leaf masks {c, 61+c}', bound='mfma', scaling='weak'),
}


# ---------------------------------------------------------------------------
# CPU baseline (the oracle; rank 0, N == 1 only)
# ---------------------------------------------------------------------------

def cpu_baseline(cfg, gpu_ll, budget_s):
    """Reference-faithful port: for every site, E x scipy.linalg.expm + nx
    marshalling + the three passes + root reduce, one thread -- exactly what
    raoteh's _mjp_dense.get_likelihood does per call (_mjp_dense.py:362-407)."""
    from oracle import oracle_numpy as orc
    from raoteh_amd import synth
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    t0 = time.perf_counter()
    done = 0
    worst = 0.0
    while done < cfg['leaf_states'].shape[0]:
        allowed = synth.site_node_to_allowed_states(cfg, done)
        ll = orc.reference_faithful_site_loglik(
            T, root, n, allowed, root_distn=cfg['root_distn'],
            Q_default=cfg['Q_default'])
        worst = max(worst, abs(ll - gpu_ll[done]) / abs(ll))
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = dict(value=done / dt, unit='sites/s', cores=1, kind='port',
               sample='%d sites of the same workload, reference-faithful '
                      '(per-site expm of every edge, numpy/scipy/networkx, '
                      '1 thread), %.1f s' % (done, dt),
               max_rel_err_gpu_vs_oracle=worst)
    # amortised variant: expm once per edge, numpy-vectorised over sites
    t0 = time.perf_counter()
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(
        T, root, n, Q_default=cfg['Q_default'])
    m = min(cfg['leaf_states'].shape[0], 20000 if n <= 20 else 2000)
    dense = synth.leaf_likelihoods(dict(cfg, leaf_states=cfg['leaf_states'][:m]))
    ll, _ = orc.batch_log_likelihoods(idx, ptr, esd,
                                      [pre.index(v) for v in cfg['leaves']],
                                      dense, cfg['root_distn'])
    dt = time.perf_counter() - t0
    out['amortised_numpy_sites_per_s'] = m / dt
    out['amortised_max_rel_err'] = float(np.max(
        np.abs(ll - gpu_ll[:m]) / np.abs(ll)))
    # compiled scalar C restatement (oracle/oracle.c), 1 thread, both modes
    try:
        from oracle import oracle_c
        from raoteh_amd._tree import TreeArrays
        ta = TreeArrays(T, root)
        Q, node_q = ta.rate_matrices(n, cfg['Q_default'])
        oidx = [ta.node_to_index[v] for v in cfg['leaves']]
        mf = max(2, min(m, int(3.0 * out['value'] * (40 if n > 20 else 8))))
        t0 = time.perf_counter()
        llc, _ = oracle_c.batch_loglik_faithful(ta.indices, ta.indptr, Q, node_q,
                                                ta.branch_lengths(), oidx,
                                                dense[:mf], cfg['root_distn'])
        out['c_port_faithful_sites_per_s'] = mf / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        lla, _ = oracle_c.batch_loglik(ta.indices, ta.indptr, esd, oidx, dense,
                                       cfg['root_distn'])
        out['c_port_amortised_sites_per_s'] = m / (time.perf_counter() - t0)
        out['c_port_max_rel_err'] = float(max(
            np.max(np.abs(llc - gpu_ll[:mf]) / np.abs(llc)),
            np.max(np.abs(lla - gpu_ll[:m]) / np.abs(lla))))
    except Exception as e:                       # the C port is optional here
        out['c_port_error'] = str(e)
    out['host_cpu_count'] = os.cpu_count()
    return out


# ---------------------------------------------------------------------------

def reduce_group_after_step(j, nb):
    """Ring of nb batches, step j runs batch j % nb.  The totals are all-reduced once
    per half rotation: -> (lo, hi) = the slice of the ring whose steps have just
    completed, or None.  A batch is reduced after its step and before the ring comes
    back to it, while the other half is being computed."""
    half = max(1, nb // 2)
    r = j % nb
    if r == half - 1:
        return (0, half)
    if r == nb - 1 and nb > half:
        return (half, nb)
    return None


def reduce_group_at_end(steps, nb):
    """The batches stepped since the last reduce_group_after_step group, after `steps`
    steps: -> (lo, hi) or None."""
    half = max(1, nb // 2)
    r = steps % nb
    if 0 < r < half:
        return (0, r)
    if half < r:
        return (half, r)
    return None


def shard_config(name, rank, world, sites):
    """(config of THIS rank's block of sites, global site count).  Weak workloads:
    the global batch has world x (configuration size) sites from the configuration's
    seed and the rank takes dist.shard_range of it; c4: the one million-site batch."""
    from raoteh_amd import synth
    from raoteh_amd.dist import shard_range
    if name == 'c4':
        total = synth.C4_NSITES if sites is None else sites
        lo, hi = shard_range(total, rank, world)
        return synth.make_config('c4', site_range=(lo, hi)), total
    per = {'c2': 100000, 'c3': 10000, 'c5': 50000, 'c6': 10000}[name] if sites is None else sites
    total = per * world
    cfg = synth.make_config(name, nsites=total)
    lo, hi = shard_range(total, rank, world)
    cfg['leaf_states'] = cfg['leaf_states'][lo:hi]
    return cfg, total


def run_workload(name, ctx, ctl, rank, world, reduce_kind, steps, warmup, args,
                 cpu_seconds, interpreter=True):
    """Run one workload; every rank takes part, rank 0 gets the result dict."""
    from raoteh_amd import synth, device, _lib
    wl = WORKLOADS[name]
    cfg, total_sites = shard_config(name, rank, world, args.sites if name == args.workload
                                    else None)
    nsites = cfg['leaf_states'].shape[0]
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    nleaves = len(cfg['leaves'])
    nedges = T.number_of_edges()

    model = device.TreeModel(T, root, n, ctx=ctx)
    model.set_rates(Q_default=cfg['Q_default'])
    model.set_root_distn(cfg['root_distn'])
    encoding = args.encoding if name == args.workload else 'dense'
    t_up = time.perf_counter()
    if name == 'c4' or encoding == 'state':
        # uint8 states cross PCIe; for n > 4 the pack kernel expands them on the device to
        # the same dense f64 resident layout a dense upload gives (31 GB for the million
        # codon sites: never materialised on the host).  For n <= 4 a state upload stays
        # one byte per leaf on the device (the compact encoding, --encoding state).
        if cfg['obs_kind'] != 'state':
            raise SystemExit('--encoding state: the leaves of this workload are allowed-state '
                             'sets, not states')
        # C4 is defined on dense leaf vectors: the states are only how the million sites cross
        # PCIe, so the kernel multiplies at the leaves as a dense upload's would (the kernels
        # that gather columns of P for observed states are next_rows.leaf_states_step_c3)
        if name == 'c4':
            ctx.set_option('leaf_state_kernels', 0)
        try:
            batch0 = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8),
                                        kind='state')
        finally:
            ctx.set_option('leaf_state_kernels', None)
    else:
        dense = synth.leaf_likelihoods(cfg)
        batch0 = model.upload_sites(cfg['leaves'], dense, kind='dense')
        del dense
    upload_s = time.perf_counter() - t_up
    # the tree-specialised kernel is compiled on a background thread (or loaded from the
    # persistent code-object cache) while the batch could already run the interpreter kernel;
    # the timed region measures the specialised kernel, so wait for it here
    t_wait = time.perf_counter()
    batch0.wait_for_kernel()
    jit_wait_s = time.perf_counter() - t_wait
    batches = [batch0]
    if not args.no_rotate:
        # >= 640 MB of distinct HBM copies; with RCCL a ring of 8 so that the totals of
        # one half are all-reduced while the other half is being computed
        while len(batches) < 8 and (sum(b.device_bytes for b in batches) < 640 * 2 ** 20 or
                                    (reduce_kind == 'rccl' and
                                     batch0.device_bytes * (len(batches) + 1) <= 16 * 2 ** 30)):
            batches.append(batch0.clone())

    def step(j):
        b = batches[j % len(batches)]
        model.step(b)          # expm of every edge + prune + reduce (rt_step)
        # RCCL: one all-reduce per HALF rotation of the batch ring, carrying the totals
        # of its steps (3 doubles each) while the other half is being computed: the
        # stream bookkeeping around a collective costs ~11 us of GPU time per call
        # whatever the payload, a quarter of a C2 step
        if reduce_kind == 'rccl':
            grp = reduce_group_after_step(j, len(batches))
            if grp is not None:
                model.allreduce_group(batches[grp[0]:grp[1]])
        return b

    # spin-up, untimed and before the W warm-up steps: a fresh box idles at its floor clock and
    # takes tens of milliseconds of sustained load to reach the clock it then holds -- a run of
    # 20 steps of 0.2 ms after 5 warm-up steps measured the ramp (0.231 ms per step, kernel 181 us),
    # not the path (0.210 ms, 171 us with --steps 200).  The same device step, without the
    # collectives of the N > 1 loop (each rank spins by its own clock).
    t_spin = time.perf_counter()
    j = 0
    while time.perf_counter() - t_spin < SPINUP_SECONDS:
        for _ in range(25):
            model.step(batches[j % len(batches)])
            j += 1
        ctx.sync()
    for j in range(warmup):
        step(j)
    ctx.sync()
    ll0, st0 = model.fetch_log_likelihoods(batch0)
    # inside the timed region: HIP events stamped with each kernel's own begin / end
    # (hipExtLaunchKernelGGL, on the library's stream) on every 8th launch of each kernel
    # (timing every launch would cost ~20 us per step)
    ctx.set_timing(0 if os.environ.get('RAOTEH_BENCH_NO_EVENTS') else 8)
    ctx.reset_timing()

    # The timed window -- EXACTLY `steps` steps between barrier + sync on both sides, the
    # maximum over the ranks -- is run `repeats` times back to back; the line reports the
    # MEDIAN window (ms_per_step, value) with the fastest and slowest next to it: a single
    # window of 20 steps is 4 ms of GPU time, and one hiccup of a shared box moves it by 5 %.
    windows = []
    enqueue = []
    for rep in range(max(1, args.repeats)):
        ctl.barrier()
        ctx.sync()
        t0 = time.perf_counter()
        for j in range(steps):
            last = step(j)
        if reduce_kind == 'rccl':
            grp = reduce_group_at_end(steps, len(batches))     # the incomplete group
            if grp is not None:
                model.allreduce_group(batches[grp[0]:grp[1]])
        t_enq = time.perf_counter()      # all steps enqueued (the launches are asynchronous)
        ctx.sync()
        ctl.barrier()
        t1 = time.perf_counter()
        windows.append(float(ctl.allreduce([t1 - t0], np.max)[0]))
        enqueue.append(t_enq - t0)
    order = np.argsort(windows)
    mid = int(order[len(order) // 2])
    elapsed = windows[mid]
    t_enq, t0 = enqueue[mid], 0.0

    totals = model.fetch_totals(last)
    if reduce_kind == 'host-socket-fallback':
        totals = ctl.allreduce(totals, np.sum)
    if reduce_kind == 'none':
        assert totals[2] == nsites
    else:
        assert totals[2] == total_sites, (totals, total_sites)
    assert np.isfinite(totals[0]) and totals[1] == 0, totals
    sampled = {}
    for key, kid in (('expm', _lib.RT_K_EXPM), ('prune', _lib.RT_K_PRUNE),
                     ('reduce', _lib.RT_K_REDUCE), ('combine', _lib.RT_K_COMBINE)):
        ms, cnt, _ = ctx.kernel_time(kid)
        sampled[key] = dict(avg_us=(ms / cnt * 1e3) if cnt else None, launches=cnt)

    # ---- roofline sample, decoupled from --steps: ROOFLINE_LAUNCHES more steps of the
    # same loop with EVERY launch event-timed (the step rate is no longer measured here)
    ctx.set_timing(1)
    ctx.reset_timing()
    for j in range(ROOFLINE_LAUNCHES):
        model.step(batches[j % len(batches)])
    ctx.sync()
    prune_ms, prune_cnt, prune_name = ctx.kernel_time(_lib.RT_K_PRUNE)
    expm_ms, expm_cnt, expm_name = ctx.kernel_time(_lib.RT_K_EXPM)
    red_ms, red_cnt, _ = ctx.kernel_time(_lib.RT_K_REDUCE)
    # root-halves launches (csrc/jit.hip): the pruning slot holds the first kernel (all the
    # arithmetic: what `roofline` prices), this slot the combine kernel that follows it
    comb_ms, comb_cnt, comb_name = ctx.kernel_time(_lib.RT_K_COMBINE)
    ctx.set_timing(0)

    # ---- the interpreter kernel a fresh topology gets before (or without) its
    # tree-specialised kernel: the same batch created with jit = 0
    interp = None
    if interpreter and rank == 0 and name != 'c4' and batch0.jit_compile_seconds >= 0 and \
            batch0.kernel_name.startswith('prune_tree_jit'):
        ctx.set_option('jit', 0)
        try:
            dense = synth.leaf_likelihoods(cfg)
            ib = model.upload_sites(cfg['leaves'], dense, kind='dense')
            del dense
            for _ in range(3):
                model.prune(ib)
            ctx.set_timing(1)
            ctx.reset_timing()
            for _ in range(16):
                model.prune(ib)
            ctx.sync()
            ims, icnt, iname = ctx.kernel_time(_lib.RT_K_PRUNE)
            ctx.set_timing(0)
            ill, _ = model.fetch_log_likelihoods(ib)
            interp = dict(kernel=iname, avg_kernel_us=ims / max(icnt, 1) * 1e3,
                          launches_timed=icnt,
                          bit_identical_to_specialised=bool(np.array_equal(ill, ll0)))
            ib.close()
        finally:
            ctx.set_option('jit', None)

    jit_compile_s = batch0.jit_compile_seconds
    for b in batches[1:]:
        b.close()
    # what a SECOND process pays for the same kernel: a fresh context (empty in-memory cache)
    # creates the same batch with the compile inside rt_sites_create -- the code object comes
    # from the cache directory (RAOTEH_JIT_CACHE_DIR, default ~/.cache/raoteh_amd/jit)
    jit_warm_s = None
    if rank == 0 and name != 'c4' and name == args.workload and jit_compile_s > 0 and \
            encoding == 'dense':
        try:
            ctx2 = device.Context(ctx.device)
            ctx2.set_option('jit_async', 0)
            model2 = device.TreeModel(T, root, n, ctx=ctx2)
            model2.set_rates(Q_default=cfg['Q_default'])
            dense = synth.leaf_likelihoods(cfg)
            b2 = model2.upload_sites(cfg['leaves'], dense, kind='dense')
            del dense
            jit_warm_s = b2.jit_compile_seconds
            ctx2.close()
        except Exception as exc:                       # noqa: BLE001 -- a side measurement
            jit_warm_s = repr(exc)

    if rank != 0:
        batch0.close()
        model.close()
        return None

    alg_bytes = nsites * (8.0 * n * nleaves + 8.0)
    if encoding == 'state' and batch0.device_bytes < nsites * 8 * n * nleaves / 4:
        # the batch really is resident as states (bytes/site drop to L; the
        # kernel is then bound by instruction issue, not by HBM)
        alg_bytes = nsites * (1.0 * nleaves + 8.0)
    alg_flops = nsites * (2.0 * n * n * nedges + n * nedges + 2.0 * n)
    avg_prune_s = prune_ms / max(prune_cnt, 1) * 1e-3
    # a root-halves launch is two kernels (rt_jit_prune + rt_jit_combine): the pruning work
    # of the launch is priced against both
    first_kernel_s = avg_prune_s
    if comb_cnt:
        avg_prune_s += comb_ms / comb_cnt * 1e-3
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic_%s.json' % name)
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
    if wl['bound'] == 'hbm':
        achieved = alg_bytes / avg_prune_s / 1e9
        roof = dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic)
        # a workload at the ridge (C5: 9.9 flop/B) prints the FP64 ceiling beside
        # the HBM one; whichever fraction is larger is the one that binds
        f64_peak, f64_src = f64_mfma_peak()
        roof['fp64_tflops'] = alg_flops / avg_prune_s / 1e12
        roof['fp64_frac'] = roof['fp64_tflops'] / f64_peak
        roof['fp64_peak'] = f64_peak
        peak_for = lambda s: alg_bytes / s / 1e9 / HBM_PEAK_GBS
    else:
        peak, peak_src = f64_mfma_peak()
        achieved = alg_flops / avg_prune_s / 1e12
        roof = dict(bound='mfma', achieved=achieved, peak=peak, unit='TFLOP/s',
                    frac=achieved / peak, traffic=traffic, peak_source=peak_src,
                    frac_of_datasheet_78_6=achieved / F64_MFMA_DATASHEET_TFLOPS)
        peak_for = lambda s: alg_flops / s / 1e12 / peak
    roof.update(kernel=prune_name, avg_kernel_us=avg_prune_s * 1e6,
                first_kernel_us=first_kernel_s * 1e6,
                launches_timed=prune_cnt,
                sampled_in_timed_region=sampled['prune'],
                algorithmic_bytes_per_launch=alg_bytes,
                algorithmic_flops_per_launch=alg_flops,
                hbm_gbs=alg_bytes / avg_prune_s / 1e9,
                step_level_frac=peak_for(elapsed / steps))
    if interp is not None:
        interp['frac'] = peak_for(interp['avg_kernel_us'] * 1e-6)

    out = {
        'value': total_sites * steps / elapsed if wl['scaling'] == 'strong'
                 else nsites * world * steps / elapsed,
        'unit': 'sites/s',
        'steps': steps,
        'warmup': warmup,
        'ms_per_step': elapsed / steps * 1e3,
        'ms_per_step_min': min(windows) / steps * 1e3,
        'ms_per_step_max': max(windows) / steps * 1e3,
        'timed_windows': len(windows),
        'host_enqueue_us_per_step': (t_enq - t0) / steps * 1e6,
        'scaling': wl['scaling'],
        'dtype': 'f64',
        'config': {'workload': '%s: %s' % (
                       name, wl['desc'] if encoding == 'dense' else
                       wl['desc'].replace('dense f64 leaf likelihood vectors',
                                          'uint8 leaf states (compact encoding)')),
                   'encoding': encoding,
                   'sites_per_gpu': nsites, 'sites_total': total_sites if wl['scaling'] == 'strong'
                   else nsites * world,
                   'states': n, 'leaves': nleaves,
                   'edges': nedges, 'batches_rotated': len(batches),
                   'reduce': reduce_kind,
                   # hiprtc compile of the tree-specialised kernel: once per distinct
                   # (tree, observed nodes) and context, in rt_sites_create, OUTSIDE the
                   # timed region; `interpreter_kernel` is what the same batch runs at
                   # until / without it (an MCMC over topologies lives there)
                   'spinup_s': SPINUP_SECONDS,
                   'jit_compile_s': jit_compile_s,
                   # cold: hiprtc on the background thread (rt_sites_create itself returned
                   # after upload_and_pack_s); warm: the same kernel for a fresh context, from
                   # the persistent code-object cache
                   'jit_cold_s': jit_compile_s,
                   'jit_warm_s': jit_warm_s,
                   'jit_wait_s': jit_wait_s,
                   'upload_and_pack_s': upload_s,
                   'interpreter_kernel': interp},
        'roofline': roof,
        # (expm None: the pruning launch computed the transitions itself -- n <= 4, one launch
        # per step, kernel name '...,expm>')
        'kernels_us': {'expm': (expm_ms / expm_cnt * 1e3) if expm_cnt else None,
                       'prune': first_kernel_s * 1e6,
                       'reduce': red_ms / max(red_cnt, 1) * 1e3,
                       'combine': (comb_ms / comb_cnt * 1e3) if comb_cnt else None,
                       'expm_kernel': expm_name,
                       'launches_timed': prune_cnt,
                       'sampled_in_timed_region': sampled},
        'total_log_likelihood': float(totals[0]),
    }
    if world == 1 and cpu_seconds > 0 and name != 'c4':
        out['cpu_baseline'] = cpu_baseline(cfg, ll0, cpu_seconds)
        out['speedup_vs_reference_faithful_cpu'] = out['value'] / out['cpu_baseline']['value']
    else:
        out['cpu_baseline'] = None
    batch0.close()
    model.close()
    return out


def measure_next_rows(ctx):
    """The paths either side of the hot path, one short measurement each,
    carried next to the headline; not part of any timed region above and never fatal."""
    import time
    out = {}
    try:
        from raoteh_amd import _mjp_dense, _sampler, synth
        for name, nsites in (('c3', 10000), ('c2', 100000)):
            cfg = synth.make_config(name, nsites=nsites)
            kw = dict(root_distn=cfg['root_distn'], Q_default=cfg['Q_default'],
                      obs_nodes=cfg['leaves'], data=cfg['leaf_states'], kind='state')
            _mjp_dense.get_expected_history_statistics_batch(cfg['T'], cfg['root'], cfg['nstates'], **kw)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                _mjp_dense.get_expected_history_statistics_batch(cfg['T'], cfg['root'],
                                                                 cfg['nstates'], **kw)
                best = min(best, time.perf_counter() - t0)
            out['expected_history_statistics_%s' % name] = dict(
                sites=nsites, seconds_per_call=best, sites_per_s=nsites / best,
                what='dwell times, root posteriors and transition counts summed over the batch, '
                     'one host call (expm, passes, site sums, Frechet block exponentials)')
        # observed codon STATES instead of dense leaf vectors (type x, the compact encoding of
        # SURVEY 8d -- an optimisation reported beside the headline, never in it): the
        # tree-specialised kernel's leaf steps gather columns of P, half of the matrix steps go
        from raoteh_amd import device as _device, _lib as _l
        cfg = synth.make_config('c3', nsites=10000)
        T, root, n = cfg['T'], cfg['root'], cfg['nstates']
        model = _device.TreeModel(T, root, n, ctx=ctx)
        model.set_rates(Q_default=cfg['Q_default'])
        model.set_root_distn(cfg['root_distn'])
        bs = model.upload_sites(cfg['leaves'], cfg['leaf_states'].astype(np.uint8), kind='state')
        bd = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
        bs.wait_for_kernel()
        bd.wait_for_kernel()
        lls, _ = model.log_likelihoods(bs)
        lld, _ = model.log_likelihoods(bd)
        for _ in range(20):
            model.step(bs)
        ctx.sync()
        ctx.set_timing(1)
        ctx.reset_timing()
        times = []
        for _ in range(9):
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                model.step(bs)
            ctx.sync()
            times.append((time.perf_counter() - t0) / 20)
        kms, kcnt, kname = ctx.kernel_time(_l.RT_K_PRUNE)
        ctx.set_timing(0)
        dt = float(np.median(times))
        nedges = T.number_of_edges()
        inner_edges = sum(1 for v in T if T.degree(v) > 1 and v != root)
        flops = 10000 * (2.0 * n * n * inner_edges + n * nedges + 2.0 * n)
        peak, _ = f64_mfma_peak()
        out['leaf_states_step_c3'] = dict(
            sites=10000, ms_per_step=dt * 1e3, sites_per_s=10000 / dt, kernel=bs.kernel_name,
            avg_kernel_us=kms / max(kcnt, 1) * 1e3,
            bit_identical_to_dense_batch=bool(np.array_equal(lls, lld)),
            roofline=dict(bound='mfma', achieved=flops / (kms / max(kcnt, 1) * 1e-3) / 1e12, peak=peak,
                          unit='TFLOP/s', frac=flops / (kms / max(kcnt, 1) * 1e-3) / 1e12 / peak,
                          algorithmic_flops_per_launch=flops,
                          note='products at the %d inner edges only: a leaf edge is a column gather'
                               % inner_edges),
            what='one step (expm of every edge + pruning) of the C3 batch uploaded as uint8 codon '
                 'states: NOT the headline encoding (dense f64 leaf vectors)')
        bs.close()
        bd.close()
        model.close()
        # c6 (allowed sets {c, 61 + c} at the leaves) and C5 (two compound states per observed
        # primary state) uploaded as masks: a leaf's message is the sum of two columns of P
        for wname, wsites in (('c6', 10000), ('c5', 50000)):
            cfgm = synth.make_config(wname, nsites=wsites)
            Tm, rootm, nm = cfgm['T'], cfgm['root'], cfgm['nstates']
            model = _device.TreeModel(Tm, rootm, nm, ctx=ctx)
            model.set_rates(Q_default=cfgm['Q_default'])
            model.set_root_distn(cfgm['root_distn'])
            wordsm = (nm + 63) // 64
            tablem = np.zeros((len(cfgm['leaf_allowed']), wordsm), dtype=np.uint64)
            for ci, ss in enumerate(cfgm['leaf_allowed']):
                for km in ss:
                    tablem[ci, km // 64] |= np.uint64(1) << np.uint64(km % 64)
            mk = tablem[cfgm['leaf_states']]
            bm = model.upload_sites(cfgm['leaves'], mk if wordsm > 1 else mk[..., 0], kind='mask')
            bm.wait_for_kernel()
            for _ in range(5):
                model.step(bm)
            ctx.sync()
            ctx.reset_timing()
            ctx.set_timing(4)
            times = []
            for _ in range(5):
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(20):
                    model.step(bm)
                ctx.sync()
                times.append((time.perf_counter() - t0) / 20)
            kms, kcnt, kname = ctx.kernel_time(_l.RT_K_PRUNE)
            ctx.set_timing(0)
            dtm = float(np.median(times))
            nedgesm = Tm.number_of_edges()
            innerm = sum(1 for v in Tm if Tm.degree(v) > 1 and v != rootm)
            flopsm = wsites * (2.0 * nm * nm * innerm + nm * nedgesm + 2.0 * nm)
            out['leaf_sets_step_%s' % wname] = dict(
                sites=wsites, ms_per_step=dtm * 1e3, sites_per_s=wsites / dtm, kernel=bm.kernel_name,
                avg_kernel_us=kms / max(kcnt, 1) * 1e3,
                roofline=dict(bound='mfma', achieved=flopsm / (kms / max(kcnt, 1) * 1e-3) / 1e12, peak=peak,
                              unit='TFLOP/s', frac=flopsm / (kms / max(kcnt, 1) * 1e-3) / 1e12 / peak,
                              algorithmic_flops_per_launch=flopsm,
                              note='products at the %d inner edges only: a leaf edge is two column gathers'
                                   % innerm),
                what='one step of the %s batch uploaded as allowed-set masks (one or two states per '
                     'leaf): NOT the encoding of workloads.%s (dense 0/1 vectors)' % (wname, wname))
            bm.close()
            model.close()
        # the same statistics on a RESIDENT batch (rt_expect_step): nothing marshalled or
        # uploaded per call; arithmetic of a call = upward pass + downward pass (products at
        # the internal nodes) + per-edge site sums, each 2 n^2 flops per edge and site, + one
        # order-2n block exponential per edge
        from raoteh_amd import device
        for name, nsites in (('c3', 10000), ('c5', 50000), ('c2', 100000)):
            cfg = synth.make_config(name, nsites=nsites)
            T, root, n = cfg['T'], cfg['root'], cfg['nstates']
            model = device.TreeModel(T, root, n, ctx=ctx)
            model.set_rates(Q_default=cfg['Q_default'])
            model.set_root_distn(cfg['root_distn'])
            batch = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
            for _ in range(3):
                model.expected_history_statistics(batch)
            times = []
            for _ in range(9):
                t0 = time.perf_counter()
                dwell, rootp, trans = model.expected_history_statistics(batch)
                times.append(time.perf_counter() - t0)
            best = float(np.median(times))
            nedges = T.number_of_edges()
            ninternal = sum(1 for v in T if T.degree(v) > 1 and v != root)
            flops = nsites * 2.0 * n * n * (2 * nedges + ninternal)
            peak, _ = f64_mfma_peak()
            total_len = sum(d['weight'] for _, _, d in T.edges(data=True))
            out['expected_history_statistics_resident_%s' % name] = dict(
                sites=nsites, seconds_per_call=best, seconds_min=min(times), seconds_max=max(times),
                sites_per_s=nsites / best,
                roofline=dict(bound='mfma', achieved=flops / best / 1e12, peak=peak, unit='TFLOP/s',
                              frac=flops / best / 1e12 / peak, algorithmic_flops_per_call=flops,
                              note='whole call (expm, three passes, Frechet block exponentials, '
                                   'result copy) against the f64 matrix peak'),
                dwell_sum_over_tree_length_times_sites=float(dwell.sum() / (total_len * nsites)),
                what='rt_expect_step: per-edge expm + upward pass + downward pass + site sums + '
                     'Frechet block exponentials on a resident batch; 2 n + n^2 numbers come back')
            batch.close()
            model.close()
        cfg = synth.make_config('c2', nsites=100000)
        T, root, n = cfg['T'], cfg['root'], cfg['nstates']
        index = _sampler.TreeArrays(T, root).node_to_index
        masks = np.full((100000, len(index)), (1 << n) - 1, dtype=np.uint64)
        masks[:, [index[v] for v in cfg['leaves']]] = \
            np.uint64(1) << cfg['leaf_states'].astype(np.uint64)
        batch = _sampler.DeviceHistoryBatch(T, root, cfg['Q_default'], node_masks=masks,
                                            root_distn=cfg['root_distn'], seed=1, ctx=ctx)
        batch.sweep(5)
        ctx.sync()
        t0 = time.perf_counter()
        batch.sweep(20)
        rows = batch.sizes()[0]
        dt = (time.perf_counter() - t0) / 20
        # bytes a sweep has to move (csrc/forest.hip; R rows before the split, R' = R + events
        # after it, K chunk-tree nodes, n states): count reads (len, state) and writes the
        # 16-bit counts; split reads the rows and the counts and writes (edge, len, chunk) of
        # the new rows and (parent, set) of the chunks; the forest passes read / write the
        # sets twice and write + read the messages (n doubles per chunk node) and the sampled
        # states; merge reads the new rows + chunk states and writes (edge, len, state)
        R = rows
        sizes = batch.sizes()
        K = sizes[1]
        Rn = K + 100000 * (len(index) - 2)
        sweep_bytes = (12 + 2) * R + (16 + 2) * R + 16 * Rn + 12 * K + 4 * 8 * K + \
            3 * 8 * n * K + 4 * K + (16 + 4) * Rn + 4 * K + 16 * R
        out['rao_teh_sweep_c2'] = dict(
            chains=100000, ms_per_sweep=dt * 1e3, chain_sweeps_per_s=100000 / dt,
            segments_per_chain=rows / 100000.0, chunk_nodes_per_chain=K / 100000.0,
            roofline=dict(bound='hbm', achieved=sweep_bytes / dt / 1e9, peak=HBM_PEAK_GBS,
                          unit='GB/s', frac=sweep_bytes / dt / 1e9 / HBM_PEAK_GBS,
                          algorithmic_bytes_per_sweep=sweep_bytes,
                          note='seven kernels per sweep (count, scan, split, sets, pmap, sample, '
                               'merge), one wave per chain: latency- and launch-bound, not '
                               'HBM-bound'),
            what='one Rao-Teh sweep of every chain, histories resident on the device')
        # the reference's optional spectral path (examples/p53/qtop.py): the decomposition of
        # the one reversible rate matrix is the caller's, once per matrix; a step rebuilds all
        # edges from it (csrc/spectral.hip) instead of running expm per edge
        from raoteh_amd import _lib, _spectral, device
        for name, nsites in (('c3', 10000), ('c2', 100000)):
            cfg = synth.make_config(name, nsites=nsites)
            t0 = time.perf_counter()
            A, lam, B, D = _spectral.decompose_rate_matrix(cfg['Q_default'], cfg['root_distn'])
            decompose_s = time.perf_counter() - t0
            model = device.TreeModel(cfg['T'], cfg['root'], cfg['nstates'], ctx=ctx)
            model.set_root_distn(cfg['root_distn'])
            batch = model.upload_sites(cfg['leaves'], synth.leaf_likelihoods(cfg), kind='dense')
            model.set_rates(Q_default=cfg['Q_default'])
            ll_expm, _ = model.log_likelihoods(batch)
            model.set_rates_spectral(A, lam, B, D=D)
            ll_spec, _ = model.log_likelihoods(batch)
            for _ in range(10):
                model.step(batch)
            ctx.sync()
            ctx.set_timing(1)
            ctx.reset_timing()
            t0 = time.perf_counter()
            for _ in range(50):
                model.step(batch)
            ctx.sync()
            dt = (time.perf_counter() - t0) / 50
            ems, ecnt, ename = ctx.kernel_time(_lib.RT_K_EXPM)
            ctx.set_timing(0)
            out['spectral_step_%s' % name] = dict(
                sites=nsites, ms_per_step_every_launch_timed=dt * 1e3, kernel=ename,
                avg_kernel_us=ems / max(ecnt, 1) * 1e3, host_decomposition_s=decompose_s,
                max_rel_diff_loglik_vs_expm=float(np.max(np.abs(ll_spec - ll_expm) /
                                                         np.abs(ll_expm))),
                what='one step with every edge rebuilt from the spectral decomposition of the '
                     'one reversible rate matrix (no cache between steps; the decomposition '
                     'itself is once per matrix, on the host as in the reference)')
            batch.close()
            model.close()
    except Exception as exc:                      # noqa: BLE001 -- a side measurement
        out['error'] = repr(exc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=9,
                    help='timed windows of --steps steps each; the median one is reported')
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS),
                    help='the headline workload (top-level value / roofline / cpu_baseline)')
    ap.add_argument('--also', default=None,
                    help="comma-separated workloads carried under 'workloads' (default: the "
                         "other ones of c2, c3, c5, plus c4; '' = none)")
    ap.add_argument('--sites', type=int, default=None,
                    help='sites per GPU of the headline workload (c4: total sites; default: '
                         'the configuration size)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-rotate', action='store_true')
    ap.add_argument('--encoding', default='dense', choices=('dense', 'state'),
                    help="resident observation encoding of the headline workload: 'dense' f64 "
                         "leaf vectors (8*n bytes per leaf) or 'state' (uint8 per leaf; "
                         "workloads whose leaves are observed states)")
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch N > 1 with: python -m torch.distributed.run '
                     '--nproc-per-node %d bench.py --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world

    from raoteh_amd import device, _lib     # fails loudly without the .so
    from raoteh_amd.dist import SocketControl, init_rccl
    ctl = SocketControl(rank, world)
    # RAOTEH_BENCH_ONE_DEVICE=1: every rank on device 0 -- a rehearsal of the N > 1 control flow on
    # a one-GPU box (RCCL refuses two ranks on one device: the host-socket reduce then runs)
    ctx = device.Context(0 if os.environ.get('RAOTEH_BENCH_ONE_DEVICE') else local_rank)

    if args.also is None:
        also = [w for w in ('c2', 'c5', 'c6', 'c4') if w != args.workload]
        if args.workload != 'c3':
            also.insert(0, 'c3')
    else:
        also = [w for w in args.also.split(',') if w]
        for w in also:
            if w not in WORKLOADS:
                sys.exit('unknown workload %r in --also' % w)

    # RCCL communicator for the data-path reduce.  librccl announces itself on stdout
    # ("Librccl path : ..."); stdout is for the one JSON line, so file descriptor 1
    # points at stderr while the library loads and the communicator is built.
    reduce_kind = 'none'
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1:
            reduce_kind = 'rccl' if init_rccl(ctx, ctl) else 'host-socket-fallback'
        elif os.environ.get('RAOTEH_BENCH_FORCE_RCCL'):
            # single-GPU rehearsal of the N > 1 data path: a 1-rank communicator
            ctx.comm_init(1, 0, device.Context.comm_unique_id())
            reduce_kind = 'rccl'
    finally:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    cpu_s = 0.0 if args.no_cpu_baseline else args.cpu_seconds
    head = run_workload(args.workload, ctx, ctl, rank, world, reduce_kind, args.steps,
                        args.warmup, args, cpu_s)
    extra = {}
    for w in also:
        # the carried workloads use their own short fixed step counts (c4: a step is
        # ~20 ms per million sites) so that the default run still finishes in minutes
        ksteps, kwarm = {'c2': (200, 20), 'c3': (100, 10), 'c5': (100, 10),
                         'c4': (10, 2), 'c6': (50, 5)}[w]
        extra[w] = run_workload(w, ctx, ctl, rank, world, reduce_kind, ksteps, kwarm, args,
                                min(cpu_s, 8.0))
    if rank != 0:
        return
    next_rows = measure_next_rows(ctx) if (world == 1 and also and not args.no_cpu_baseline) else None

    out = {
        'metric': 'site log-likelihoods/sec (batched tree pruning)',
        'value': head['value'],
        'unit': 'sites/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': head['ms_per_step'],
        'ms_per_step_min': head['ms_per_step_min'],
        'ms_per_step_max': head['ms_per_step_max'],
        'timed_windows': head['timed_windows'],
        'host_enqueue_us_per_step': head['host_enqueue_us_per_step'],
        'higher_is_better': True,
        'scaling': head['scaling'],
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': head['config'],
        'roofline': head['roofline'],
        'kernels_us': head['kernels_us'],
        'total_log_likelihood': head['total_log_likelihood'],
        'cpu_baseline': head['cpu_baseline'],
    }
    if head.get('speedup_vs_reference_faithful_cpu') is not None:
        out['speedup_vs_reference_faithful_cpu'] = head['speedup_vs_reference_faithful_cpu']
    out['workloads'] = extra
    if next_rows is not None:
        out['next_rows'] = next_rows
    print(json.dumps(out))


if __name__ == '__main__':
    main()
