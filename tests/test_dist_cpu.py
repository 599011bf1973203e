"""
The N > 1 path on CPU: world_size-2 (and 3) process groups over gloo exercise
the product's sharding and reduction code (raoteh_amd/dist.py) -- shard ranges,
the host-side reduce of the (sum log-lik, #zero, #sites) triple that backs up
the RCCL all-reduce, and the socket control plane bench.py uses.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from raoteh_amd.dist import shard_range


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_ranges_partition_the_sites():
    for nsites in (1, 7, 64, 1003, 100000):
        for world in (1, 2, 3, 8):
            prev = 0
            for rank in range(world):
                lo, hi = shard_range(nsites, rank, world)
                assert lo == prev and hi >= lo
                prev = hi
                for i in (lo, hi - 1):
                    if lo < hi:
                        assert i * world // nsites == rank
            assert prev == nsites
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_reduce_matches_single_process(tmp_path, world):
    out = tmp_path / 'result.json'
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', '_dist_worker.py'), str(out)]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, timeout=600)
    assert proc.returncode == 0, proc.stdout.decode()[-3000:]
    res = json.load(open(out))
    want = res['want']
    assert want[1] == 1.0 and want[2] == 1003.0
    for key in ('gloo', 'socket'):
        got = res[key]
        assert got[1:] == want[1:]
        assert got[0] == pytest.approx(want[0], rel=1e-13)
    assert res['max_rank'] == world - 1
    # expected history statistics: shard sums reduced over both control planes
    for key in ('expect_gloo', 'expect_socket'):
        for got_arr, want_arr in zip(res[key], res['expect_want']):
            np.testing.assert_allclose(got_arr, want_arr, rtol=1e-13, atol=1e-15)
    # init_rccl with a rank failing at each stage (rank 0 used to skip the
    # all-gather and desynchronise the control plane; the others hung)
    assert len(res['rccl']) == 12
    for row in res['rccl']:
        assert row['agree'] and row['after_ok'] and not row['leaked'], row
        assert row['got'] == (row['fail'] is None), row
    # Rao-Teh chains sharded over the ranks: the sample sums of the shards add up to the
    # unsharded batch's
    np.testing.assert_allclose(res['chains']['dwell'], res['chains_want']['dwell'], rtol=1e-13)
    np.testing.assert_array_equal(res['chains']['trans'], res['chains_want']['trans'])
    assert res['chains']['range'][0] == 0 and res['chains']['seed'] == 9
    ranges = res['ranges']
    assert ranges[0][0] == 0 and ranges[-1][1] == 1003
    for a, b in zip(ranges, ranges[1:]):
        assert a[1] == b[0]


def test_bench_reduce_grouping_covers_every_step_once():
    """bench.py (N > 1) all-reduces the totals of its batch ring once per half rotation:
    every step's totals are reduced exactly once, after the step and before the ring
    comes back to that batch."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'bench', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                              'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for nb in (1, 2, 3, 4, 5, 8):
        for steps in (1, 2, 7, 8, 9, 40, 41, 43):
            pending = {}                       # batch -> step whose totals await a reduce
            reduced = []
            for j in range(steps):
                b = j % nb
                assert b not in pending, (nb, steps, j)      # reduced before reuse
                pending[b] = j
                grp = bench.reduce_group_after_step(j, nb)
                if grp is not None:
                    for k in range(*grp):
                        reduced.append(pending.pop(k))
            grp = bench.reduce_group_at_end(steps, nb)
            if grp is not None:
                for k in range(*grp):
                    reduced.append(pending.pop(k))
            assert not pending, (nb, steps, pending)
            assert sorted(reduced) == list(range(steps))


def test_bench_shards_are_slices_of_one_batch():
    """bench.py at N > 1: the weak workloads shard ONE global batch of N x (configuration
    size) sites by dist.shard_range, config 4 shards the one million-site batch; the
    ranks' blocks tile the batch in order whatever the rank count."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from raoteh_amd import synth
    whole, total = bench.shard_config('c2', 0, 1, 600)
    assert total == 600 and whole['leaf_states'].shape == (600, 64)
    for world in (2, 3):
        parts = [bench.shard_config('c2', r, world, 200)[0]['leaf_states'] for r in range(world)]
        ref = synth.make_config('c2', nsites=200 * world)['leaf_states']
        np.testing.assert_array_equal(np.concatenate(parts), ref)
        assert bench.shard_config('c2', 0, world, 200)[1] == 200 * world
    # config 4 (strong scaling): total fixed, the rank blocks are slices of the same batch
    full = synth.make_config('c4', site_range=(0, 40000))['leaf_states']
    got = [bench.shard_config('c4', r, 4, 40000) for r in range(4)]
    assert all(t == 40000 for _, t in got)
    np.testing.assert_array_equal(np.concatenate([c['leaf_states'] for c, _ in got]), full)
    lo, hi = shard_range(synth.C4_NSITES, 3, 8)
    assert (lo, hi) == (375000, 500000)
    assert bench.WORKLOADS['c4']['scaling'] == 'strong' and bench.WORKLOADS['c3']['scaling'] == 'weak'
