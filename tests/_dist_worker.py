"""Worker of tests/test_dist_cpu.py: launched by torch.distributed.run with
WORLD_SIZE ranks on the CPU.  Each rank evaluates its shard of sites with the
oracle (standing in for the GPU, which this container lacks), then the product's
sharding + reduction code (raoteh_amd.dist) combines the shards over (a) a gloo
process group and (b) the socket control plane bench.py uses."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist                      # noqa: E402

from oracle import oracle_numpy as orc                # noqa: E402
from raoteh_amd import synth                          # noqa: E402
from raoteh_amd.dist import (SocketControl, TorchControl, env_rank_world,  # noqa: E402
                             init_rccl, reduce_history_statistics, reduce_totals,
                             shard_range)


class StubComm(object):
    """Stands in for device.Context in init_rccl: `fail` = (stage, rank) makes that
    stage fail on that rank only."""

    def __init__(self, rank, fail):
        self.rank, self.fail = rank, fail
        self.inited = self.destroyed = False

    def _hit(self, stage):
        return self.fail == (stage, self.rank)

    def comm_available(self):
        return not self._hit('load')

    def comm_unique_id(self):
        if self._hit('uid'):
            raise RuntimeError('stub: ncclGetUniqueId failed')
        return bytes(range(128))

    def comm_init(self, world, rank, uid):
        assert uid == bytes(range(128))
        if self._hit('init'):
            raise RuntimeError('stub: ncclCommInitRank failed')
        self.inited = True

    def comm_destroy(self):
        self.destroyed = True


def main():
    out_path = sys.argv[1]
    rank, local_rank, world = env_rank_world()
    dist.init_process_group('gloo')
    cfg = synth.make_config('c2', nsites=1003)
    T, root, n = cfg['T'], cfg['root'], cfg['nstates']
    dense = synth.leaf_likelihoods(cfg)
    dense[17, 0, :] = 0.0                              # one zero-probability site
    pre, idx, ptr, esd = orc.get_expm_augmented_transitions(
        T, root, n, Q_default=cfg['Q_default'])
    oidx = [pre.index(v) for v in cfg['leaves']]
    lo, hi = shard_range(dense.shape[0], rank, world)
    ll, st = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense[lo:hi],
                                       cfg['root_distn'])
    local = np.array([ll[st == 0].sum(), float((st != 0).sum()), float(hi - lo)])
    results = {}
    tc = TorchControl()
    assert (tc.rank, tc.world) == (rank, world)
    results['gloo'] = reduce_totals(local, tc).tolist()
    tc.barrier()
    gathered = tc.allgather(np.array([lo, hi], dtype=np.int64).tobytes())
    results['ranges'] = [np.frombuffer(g, dtype=np.int64).tolist() for g in gathered]
    # expected history statistics of a sharded alignment: per-shard sums add
    small = synth.make_config('c1', nsites=7)
    sT, sroot, sn = small['T'], small['root'], small['nstates']

    def shard_statistics(a, b):
        d, i, t = np.zeros(sn), np.zeros(sn), np.zeros((sn, sn))
        for k in range(a, b):
            od, oi, ot = orc.mjp_dense_get_expected_history_statistics(
                sT, synth.site_node_to_allowed_states(small, k), sroot, sn,
                root_distn=small['root_distn'], Q_default=small['Q_default'])
            d, i, t = d + od, i + oi, t + ot
        return d, i, t
    slo, shi = shard_range(7, rank, world)
    got = reduce_history_statistics(shard_statistics(slo, shi), tc)
    results['expect_gloo'] = [a.tolist() for a in got]
    sc = SocketControl(rank, world)
    got = reduce_history_statistics(shard_statistics(slo, shi), sc)
    results['expect_socket'] = [a.tolist() for a in got]
    if rank == 0:
        results['expect_want'] = [a.tolist() for a in shard_statistics(0, 7)]
    # RCCL bring-up with one rank failing at each stage: every rank must get the same
    # answer and the control plane must stay in step (the reduce after it is right)
    rccl = []
    for ctl in (tc, sc):
        for fail in (None, ('load', 0), ('load', world - 1), ('uid', 0), ('init', 0),
                     ('init', world - 1)):
            stub = StubComm(rank, fail)
            got = init_rccl(stub, ctl)
            after = float(ctl.allreduce([rank + 1.0], np.sum)[0])
            answers = ctl.allgather(bytes([1 if got else 0]))
            rccl.append(dict(fail=fail, got=got, agree=len(set(answers)) == 1,
                             after_ok=after == world * (world + 1) / 2.0,
                             leaked=bool(stub.inited and not got and not stub.destroyed)))
    results['rccl'] = rccl
    # chains of the Rao-Teh sampler sharded over the ranks: no collective in a sweep, the
    # sample sums add over the control plane (a stand-in batch: no GPU here)
    from raoteh_amd.dist import ShardedHistoryBatch

    class StubBatch(object):
        def __init__(self, T, root, Q, node_masks=None, root_distn=None,
                     uniformization_factor=2, seed=0, ctx=None):
            self.masks, self.n, self.sweeps, self.seed = node_masks, Q.shape[0], 0, seed

        def sweep(self):
            self.sweeps += 1

        def dwell_times(self):
            return (self.masks[:, :self.n] % 7).astype(float) + self.sweeps

        def transition_counts(self):
            c = (self.masks[:, 0] % 5).astype(np.int64)
            return c[:, None, None] * np.ones((1, self.n, self.n), dtype=np.int64)

    all_masks = (np.arange(11 * 6).reshape(11, 6) * 37 + 5).astype(np.uint64)
    sh = ShardedHistoryBatch(None, None, np.zeros((4, 4)), all_masks, sc, seed=9,
                             batch_cls=StubBatch)
    sh.sweep(3)
    d, t = sh.statistics_total()
    results['chains'] = dict(dwell=d.tolist(), trans=t.tolist(), range=list(sh.range),
                             seed=None if sh.batch is None else int(sh.batch.seed))
    if rank == 0:
        whole = StubBatch(None, None, np.zeros((4, 4)), node_masks=all_masks)
        whole.sweeps = 3
        results['chains_want'] = dict(dwell=whole.dwell_times().sum(axis=0).tolist(),
                                      trans=whole.transition_counts().sum(axis=0).tolist())
    results['socket'] = reduce_totals(local, sc).tolist()
    results['max_rank'] = float(sc.allreduce([float(rank)], np.max)[0])
    sc.barrier()
    sc.close()
    if rank == 0:
        full, fst = orc.batch_log_likelihoods(idx, ptr, esd, oidx, dense,
                                              cfg['root_distn'])
        results['want'] = [float(full[fst == 0].sum()), float((fst != 0).sum()),
                           float(dense.shape[0])]
        with open(out_path, 'w') as f:
            json.dump(results, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
