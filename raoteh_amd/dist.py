"""
Multi-GPU: sites shard, nothing else does.

Sites (alignment columns) are independent given (tree, Q, branch lengths); the
reference sums their log-likelihoods in a plain loop
(examples/p53/p53.py:88-100).  One process per GPU owns a contiguous block of
sites; tree, rates and root distribution are replicated and every GPU runs the
per-edge expm itself (deterministic, a few KB..MB).  The only data-path
exchange is the sum of three doubles (sum log-lik, #zero-probability sites,
#sites): ncclAllReduce over RCCL/xGMI inside libraoteh_hip.so
(rt_allreduce_totals).  The control plane (unique-id exchange, barriers,
max-over-ranks timing) is host-side and tiny; two interchangeable
implementations are provided:

* ``SocketControl``  -- a few bytes over a local TCP socket (single node, no
  torch import in the worker processes; used by bench.py)
* ``TorchControl``   -- an already-initialised ``torch.distributed`` process
  group (gloo on CPU in the tests, any backend elsewhere)
"""
from __future__ import annotations

import os
import socket
import struct
import time

import numpy as np

__all__ = ['shard_range', 'SocketControl', 'TorchControl', 'env_rank_world',
           'reduce_totals', 'init_rccl', 'ShardedLikelihood']


def shard_range(nsites, rank, world):
    """Contiguous block of sites of ``rank``: site i lives on GPU
    floor(i * world / nsites)."""
    if not (0 <= rank < world):
        raise ValueError('rank %d not in [0, %d)' % (rank, world))
    lo = -(-rank * nsites // world)            # ceil(rank * nsites / world)
    hi = -(-(rank + 1) * nsites // world)
    return lo, hi


def env_rank_world():
    """(rank, local_rank, world) from the torchrun environment."""
    return (int(os.environ.get('RANK', '0')),
            int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


class SocketControl(object):
    """all-gather of small equal-length byte strings among the ranks of ONE
    node.  Rendezvous: rank 0 listens on an ephemeral 127.0.0.1 port and
    publishes it in a file keyed by MASTER_PORT and the launcher's pid (all
    workers of one torchrun share their parent)."""

    def __init__(self, rank, world, token=None, timeout=300.0):
        self.rank, self.world = rank, world
        self.peers, self.sock = [], None
        if world == 1:
            return
        if token is None:
            token = '%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.getppid())
        path = '/tmp/raoteh_rdzv_%s' % token
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))
            srv.listen(world)
            srv.settimeout(timeout)
            with open(path + '.tmp', 'w') as f:
                f.write(str(srv.getsockname()[1]))
            os.rename(path + '.tmp', path)
            conns = {}
            try:
                while len(conns) < world - 1:
                    c, _ = srv.accept()
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conns[struct.unpack('i', self._recv(c, 4))[0]] = c
            finally:
                srv.close()
                try:
                    os.unlink(path)
                except OSError:
                    pass
            self.peers = [conns[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise RuntimeError('rendezvous file %s never appeared' % path)
                time.sleep(0.02)
            port = int(open(path).read())
            self.sock = socket.create_connection(('127.0.0.1', port),
                                                 timeout=timeout)
            self.sock.settimeout(None)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.sock.sendall(struct.pack('i', rank))

    @staticmethod
    def _recv(c, n):
        buf = b''
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise RuntimeError('peer closed the control connection')
            buf += chunk
        return buf

    def allgather(self, payload):
        if self.world == 1:
            return [payload]
        n = len(payload)
        if self.rank == 0:
            blob = b''.join([payload] + [self._recv(c, n) for c in self.peers])
            for c in self.peers:
                c.sendall(blob)
        else:
            self.sock.sendall(payload)
            blob = self._recv(self.sock, n * self.world)
        return [blob[i * n:(i + 1) * n] for i in range(self.world)]

    def barrier(self):
        self.allgather(b'\0')

    def allreduce(self, values, op=np.sum):
        vals = np.ascontiguousarray(values, dtype=np.float64)
        parts = self.allgather(vals.tobytes())
        return op(np.stack([np.frombuffer(p, dtype=np.float64) for p in parts]),
                  axis=0)

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None


class TorchControl(object):
    """Same interface on top of an initialised torch.distributed group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self._dist, self._group = dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def allgather(self, payload):
        import torch
        mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(outs, mine, group=self._group)
        return [bytes(o.numpy().tobytes()) for o in outs]

    def barrier(self):
        self._dist.barrier(group=self._group)

    def allreduce(self, values, op=np.sum):
        import torch
        t = torch.from_numpy(np.array(values, dtype=np.float64, copy=True))
        ops = {np.sum: self._dist.ReduceOp.SUM, np.max: self._dist.ReduceOp.MAX,
               np.min: self._dist.ReduceOp.MIN}
        self._dist.all_reduce(t, op=ops[op], group=self._group)
        return t.numpy()

    def close(self):
        pass


def reduce_totals(local_totals, control):
    """Host-side sum over ranks of the (sum log-lik, #zero, #sites) triple --
    the fallback of the RCCL reduce and what the CPU tests exercise."""
    return control.allreduce(np.asarray(local_totals, dtype=np.float64), np.sum)


def reduce_history_statistics(stats, control):
    """Sum over ranks of the (dwell f64[n], root posteriors f64[n], transitions
    f64[n, n]) triple each rank got from
    _mjp_dense.get_expected_history_statistics_batch on ITS shard of the sites
    (shard_range): the statistics are sums over sites, so the shards add.  3 small
    arrays per iteration: they go over the control plane, not RCCL."""
    dwell, init, trans = (np.asarray(a, dtype=np.float64) for a in stats)
    n = dwell.shape[0]
    if init.shape != (n,) or trans.shape != (n, n):
        raise ValueError('expected (f64[n], f64[n], f64[n, n])')
    flat = control.allreduce(np.concatenate([dwell, init, trans.ravel()]), np.sum)
    return flat[:n], flat[n:2 * n], flat[2 * n:].reshape(n, n)


def init_rccl(ctx, control):
    """Create the RCCL communicator of ``ctx`` across the ranks of ``control``.
    Returns True on every rank or False on every rank (never mixed).

    Every rank takes part in every control-plane collective below, whatever failed
    locally, so the byte streams of the control plane never desynchronise:

    1. min-reduce "librccl loads here" (``ctx.comm_available()``: dlopen only).  If any
       rank lacks it, nobody enters ``ncclCommInitRank`` -- a rank waiting there for a
       peer that never comes would hang.
    2. rank 0 creates the unique id; ALL ranks all-gather (flag byte + 128 id bytes).
       A failed id creation is a cleared flag, not a skipped collective.
    3. every rank calls ``ctx.comm_init``; min-reduce the outcome; on disagreement the
       ranks that did get a communicator destroy it again.
    (A rank dying INSIDE ncclCommInitRank is RCCL's to time out; the host side cannot
    see it.)"""
    if control.world <= 1:
        return True
    import sys

    def note(msg):
        sys.stderr.write('rank %d: RCCL unavailable: %s\n' % (control.rank, msg))

    try:
        avail = 1.0 if ctx.comm_available() else 0.0
    except Exception as e:
        note(e)
        avail = 0.0
    if float(control.allreduce([avail], np.min)[0]) < 1.0:
        if avail:
            note('another rank cannot load librccl')
        return False
    payload = bytes(129)
    if control.rank == 0:
        try:
            payload = b'\x01' + bytes(ctx.comm_unique_id())
            if len(payload) != 129:
                raise RuntimeError('unique id has %d bytes' % (len(payload) - 1))
        except Exception as e:
            note(e)
            payload = bytes(129)
    head = control.allgather(payload)[0]
    if head[:1] != b'\x01':
        return False
    ok = 1.0
    try:
        ctx.comm_init(control.world, control.rank, head[1:])
    except Exception as e:
        note(e)
        ok = 0.0
    if float(control.allreduce([ok], np.min)[0]) < 1.0:
        if ok:
            try:
                ctx.comm_destroy()
            except Exception as e:
                note(e)
        return False
    return True


class ShardedLikelihood(object):
    """Total log-likelihood of a site batch sharded over the ranks.

    Every rank passes the FULL observation array (or its own slice with
    ``presharded=True``); each uploads only its block.  ``total()`` returns the
    global (sum log-lik or -inf, #zero-probability sites, #sites) on every
    rank."""

    def __init__(self, model, control, obs_nodes, data, kind='dense',
                 presharded=False, use_rccl=True):
        self.model, self.control = model, control
        data = np.asarray(data)
        if not presharded:
            lo, hi = shard_range(data.shape[0], control.rank, control.world)
            data = data[lo:hi]
        self.batch = model.upload_sites(obs_nodes, data, kind=kind)
        self.rccl = bool(use_rccl and control.world > 1 and
                         init_rccl(model.ctx, control))

    def total(self):
        self.model.prune(self.batch)
        if self.rccl:
            self.model.allreduce(self.batch)
            tot = self.model.fetch_totals(self.batch)
        else:
            tot = reduce_totals(self.model.fetch_totals(self.batch), self.control)
        nzero = int(tot[1])
        return (-np.inf if nzero else float(tot[0])), nzero, int(tot[2])


class ShardedHistoryBatch(object):
    """Rao-Teh chains sharded over the ranks (raoteh_amd/_sampler.py).  Chains are
    independent, so a sweep needs no collective at all; only the sample sums a caller
    accumulates (dwell times per state, transition counts) cross the ranks, over the
    control plane.  Every rank passes the FULL node_masks uint64[nchains, nnodes]; each
    creates the batch of its block ``shard_range(nchains, rank, world)`` with a seed of
    its own.  ``batch_cls`` defaults to the device-resident DeviceHistoryBatch."""

    def __init__(self, T, root, Q, node_masks, control, root_distn=None,
                 uniformization_factor=2, seed=0, ctx=None, batch_cls=None):
        node_masks = np.asarray(node_masks)
        self.control = control
        self.nchains_total = int(node_masks.shape[0])
        lo, hi = shard_range(self.nchains_total, control.rank, control.world)
        self.range = (lo, hi)
        if batch_cls is None:
            from ._sampler import DeviceHistoryBatch as batch_cls
        self.batch = None
        if hi > lo:
            self.batch = batch_cls(T, root, Q, node_masks=node_masks[lo:hi],
                                   root_distn=root_distn,
                                   uniformization_factor=uniformization_factor,
                                   seed=int(seed) + (control.rank << 32), ctx=ctx)
        self.nstates = int(np.asarray(Q).shape[0])

    def sweep(self, nsweeps=1):
        if self.batch is not None:
            for _ in range(int(nsweeps)):
                self.batch.sweep()
        return self

    def statistics_total(self):
        """(dwell f64[n], transitions f64[n, n]) summed over ALL chains of all ranks."""
        n = self.nstates
        flat = np.zeros(n + n * n)
        if self.batch is not None:
            flat[:n] = self.batch.dwell_times().sum(axis=0)
            flat[n:] = self.batch.transition_counts().sum(axis=0).ravel()
        flat = self.control.allreduce(flat, np.sum)
        return flat[:n], flat[n:].reshape(n, n)

