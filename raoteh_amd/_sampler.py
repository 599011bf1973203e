"""
Rao-Teh sampling of Markov-jump-process histories on a tree, batched over chains.

The reference's sampler (raoteh/sampler/_sampler.py:300-390, ``gen_restricted_histories``)
keeps ONE history as a networkx tree and per sweep

  1. removes the redundant (self-transition) event nodes   _graph_transform.py:55-216
  2. adds Poisson events of rate omega - q(state) on every segment
                                                            _sample_mjp_dense.py:21-69
  3. cuts the tree at the event nodes into chunks of constant state and builds the
     chunk tree                                             _graph_transform.py:298-375
  4. re-samples one state per chunk from the posterior under the uniformized
     P = I + Q / omega, restricted by the allowed sets of the original nodes inside
     each chunk                                             _sample_mcy.py:19-165
  5. writes the chunk states back onto the segments.

Here a batch of chains (independent sites, or replicates of one site) goes through the
same five steps together.  The histories are flat arrays -- one row per segment, sorted
by (chain, edge, position): chain, edge (preorder index of the edge's child node),
length, state -- steps 1, 2, 3 and 5 are numpy passes over those rows, and step 4 is one
call of the device's ragged-forest sampler (``rt_forest_resample_states_parents``,
csrc/forest.hip: one wave per chunk tree, lane = state, Philox draws).  The chunk trees
come out in an order in which every chunk's parent has a smaller index, which is the
layout that entry point takes; no child lists are built.

    batch = HistoryBatch(T, root, Q, node_masks=masks)      # uint64[nchains, nnodes]
    for _ in range(nsweeps):
        batch.sweep()
        dwell, trans = batch.dwell_times(), batch.transition_counts()

``gen_restricted_histories`` is the reference's generator on top of a batch of one.
There is no CPU fallback for step 4.
"""
from __future__ import annotations

import ctypes
import time
from ctypes import c_double, c_int32, c_int64, c_uint64

import networkx as nx
import numpy as np

from . import _lib
from ._tree import TreeArrays, check_square_dense
from ._util import StructuralZeroProb
from .device import get_context

__all__ = ['HistoryBatch', 'DeviceHistoryBatch', 'gen_restricted_histories', 'gen_mh_histories',
           'trajectory_log_likelihoods', 'gen_histories', 'get_total_rates',
           'poisson_split', 'chunk_forest', 'merge_segments']


def _ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def get_total_rates(Q):
    """Rate away from each state (_mjp_dense.get_total_rates: minus the diagonal)."""
    check_square_dense(Q)
    return -np.diag(Q).astype(np.float64)


def trajectory_log_likelihoods(dwell, transitions, root_states, Q, root_distn=None):
    """_mjp_dense.get_trajectory_log_likelihood (:188-240) for a batch of histories given
    by their statistics: log prior of the root state - sum_a dwell[a] q_a + sum_{a != b}
    transitions[a, b] log Q[a, b].  root_distn None: no root term."""
    Q = np.asarray(Q, dtype=np.float64)
    n = Q.shape[0]
    rates = get_total_rates(Q)
    off = ~np.eye(n, dtype=bool)
    with np.errstate(divide='ignore'):
        logq = np.where(off & (Q > 0), np.log(np.where(Q > 0, Q, 1.0)), -np.inf)
    tr = np.asarray(transitions, dtype=np.float64)
    trans_ll = np.where(tr[:, off] > 0, tr[:, off] * logq[off][None, :], 0.0).sum(axis=1)
    ll = -np.asarray(dwell, dtype=np.float64).dot(rates) + trans_ll
    if root_distn is not None:
        with np.errstate(divide='ignore'):
            ll = ll + np.log(np.asarray(root_distn, dtype=np.float64)[np.asarray(root_states)])
    return ll


def _mh_step(batch, target, rng, cache):
    """One Metropolis-Hastings step of every chain (_sampler.py:470-520): a Rao-Teh sweep
    under the batch's own process is the proposal; ``target(batch) -> f64[nchains]`` is the
    log density it is corrected to.  Returns the bool[nchains] acceptance flags; rejected
    chains are back at their previous history."""
    if cache.get('biased') is None:
        cache['biased'] = batch.trajectory_log_likelihoods()
        cache['target'] = np.asarray(target(batch), dtype=np.float64)
    batch.snapshot()
    batch.sweep()
    biased = batch.trajectory_log_likelihoods()
    tgt = np.asarray(target(batch), dtype=np.float64)
    log_ratio = tgt - cache['target'] - biased + cache['biased']
    accept = (log_ratio > 0) | (rng.random(batch.nchains) < np.exp(np.minimum(log_ratio, 0.0)))
    accept &= ~np.isnan(log_ratio)
    batch.restore(~accept)
    cache['biased'] = np.where(accept, biased, cache['biased'])
    cache['target'] = np.where(accept, tgt, cache['target'])
    return accept


# ---------------------------------------------------------------------------
# host passes over the segment rows (pure numpy: tested on the CPU)
# ---------------------------------------------------------------------------

def poisson_split(rng, seg_len, seg_rate):
    """Step 2.  For every segment draw k ~ Poisson(rate * length) event points, uniform on
    the segment (_sample_mjp_dense.py:47-61 walks exponential gaps until they pass the
    end: the same process).  Returns (rep int64[S], sub_len f64[S']): segment i becomes
    rep[i] = k_i + 1 consecutive pieces whose lengths are sub_len (the gaps of k sorted
    uniforms are k + 1 exponentials scaled to the length)."""
    seg_len = np.asarray(seg_len, dtype=np.float64)
    k = rng.poisson(np.maximum(seg_rate, 0.0) * seg_len)
    rep = (k + 1).astype(np.int64)
    owner = np.repeat(np.arange(seg_len.shape[0]), rep)
    gaps = rng.standard_exponential(owner.shape[0])
    tot = np.bincount(owner, weights=gaps, minlength=seg_len.shape[0])
    sub = gaps / tot[owner] * seg_len[owner]
    single = rep[owner] == 1
    sub[single] = seg_len[owner[single]]                  # untouched segments stay exact
    return rep, sub


def chunk_forest(parent, nchains, chain, edge):
    """Step 3 for all chains.  ``parent`` int[nnodes] of the base tree in preorder (root
    first, parent < child); rows sorted by (chain, edge, position), every (chain, edge)
    present.  Every boundary between two rows of one (chain, edge) is an event node.  A
    chunk is a maximal region without an event inside: the first piece of an edge lies in
    the chunk of the edge's upper node, the last in the chunk of its lower node, pieces in
    between are chunks of their own.  Returns

      offset  int64[nchains + 1]  chain c's chunks are [offset[c], offset[c + 1])
      cparent int32[total]        local index of each chunk's parent (-1 at chunk 0, which
                                  holds the root), always smaller than the chunk's own
      piece   int64[S]            global chunk index of every row
      node    int64[nchains, nnodes]  local chunk index of every base node

    (_graph_transform.get_chunk_tree_type_b, :298-375, numbers chunks in BFS order of one
    tree; here ids follow the preorder of the base edges, then the position on the edge.)"""
    parent = np.asarray(parent, dtype=np.int64)
    N = parent.shape[0]
    S = chain.shape[0]
    first = np.ones(S, dtype=bool)
    first[1:] = (chain[1:] != chain[:-1]) | (edge[1:] != edge[:-1])
    # events per (chain, edge)
    M = np.bincount((chain * N + edge)[~first], minlength=nchains * N).reshape(nchains, N)
    csum = np.cumsum(M, axis=1)
    base = 1 + csum - M                                     # id of an edge's first own chunk
    node = np.zeros((nchains, N), dtype=np.int64)
    for v in range(1, N):
        node[:, v] = np.where(M[:, v] > 0, base[:, v] + M[:, v] - 1, node[:, parent[v]])
    nch = 1 + csum[:, -1] if N else np.ones(nchains, dtype=np.int64)
    offset = np.zeros(nchains + 1, dtype=np.int64)
    np.cumsum(nch, out=offset[1:])
    idx = np.arange(S)
    k = idx - np.maximum.accumulate(np.where(first, idx, 0))
    local = np.where(k == 0, node[chain, parent[edge]], base[chain, edge] + k - 1)
    piece = offset[chain] + local
    cparent = np.full(int(offset[-1]), -1, dtype=np.int32)
    later = np.nonzero(~first)[0]
    cparent[piece[later]] = local[later - 1]
    return offset, cparent, piece, node


def merge_segments(chain, edge, length, state):
    """Steps 5 + 1: neighbouring rows of one (chain, edge) with the same state are one
    segment (the event between them was a self transition: a redundant degree-two node,
    _graph_transform.py:55-83)."""
    S = chain.shape[0]
    start = np.ones(S, dtype=bool)
    start[1:] = ((chain[1:] != chain[:-1]) | (edge[1:] != edge[:-1]) |
                 (state[1:] != state[:-1]))
    at = np.nonzero(start)[0]
    return chain[at], edge[at], np.add.reduceat(length, at), state[at]


# ---------------------------------------------------------------------------
# the batch
# ---------------------------------------------------------------------------

class HistoryBatch(object):
    """``nchains`` Rao-Teh chains on the tree ``T`` rooted at ``root`` under the dense rate
    matrix ``Q`` (diagonal = minus the row sums).

    node_masks : uint64[nchains, nnodes] allowed-set bit masks by preorder index
        (``self.tree.preorder_nodes``), or give ``node_to_allowed_states``: one dict
        {node: set of states} per chain (a missing node is unrestricted), or ONE dict
        together with ``nchains`` for replicate chains of the same data.
    root_distn : f64[nstates] or None (weights of one, _sample_mc0_dense.py:53-56)
    uniformization_factor : omega = factor * max total rate (> 1, _sampler.py:344-352)
    seed : seeds the host generator (Poisson events) and keys the device's Philox draws.

    The constructor finds a first feasible history by bisecting the edges until the
    chunk trees have positive likelihood (_sampler.py:563-648); it is not a posterior
    draw, discard the first sweeps."""

    def __init__(self, T, root, Q, node_masks=None, node_to_allowed_states=None, nchains=None,
                 root_distn=None, uniformization_factor=2, seed=0, ctx=None):
        self._setup(T, root, Q, node_masks, node_to_allowed_states, nchains, root_distn,
                    uniformization_factor, seed, ctx)
        self._init_feasible()

    def _setup(self, T, root, Q, node_masks, node_to_allowed_states, nchains, root_distn,
               uniformization_factor, seed, ctx):
        if uniformization_factor <= 1:
            raise ValueError('the uniformization factor must be greater than 1')
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        check_square_dense(Q)
        n = Q.shape[0]
        if n > 64:
            raise ValueError('the forest passes hold a state per lane: nstates <= 64')
        self.nstates = n
        self.Q = Q
        self.tree = TreeArrays(T, root)
        N = self.tree.nnodes
        self.parent = self.tree.parent.copy()
        self.branch = self.tree.branch_lengths()
        rates = get_total_rates(Q)
        if not rates.max() > 0:
            raise ValueError('the rate matrix is empty')
        self.omega = float(uniformization_factor) * float(rates.max())
        self.P = np.identity(n) + Q / self.omega            # _sample_mjp_dense.py:107-114
        self.poisson_rates = self.omega - rates              # _sampler.py:356-357
        full = np.uint64((1 << n) - 1)
        if node_masks is not None:
            masks = np.ascontiguousarray(node_masks, dtype=np.uint64)
            if masks.ndim != 2 or masks.shape[1] != N:
                raise ValueError('node_masks must be [nchains, %d]' % N)
        else:
            dicts = node_to_allowed_states
            if dicts is None or isinstance(dicts, dict):
                dicts = [dicts] * int(nchains or 1)
            masks = np.full((len(dicts), N), full, dtype=np.uint64)
            cache = {}
            for c, d in enumerate(dicts):
                if d is None:
                    continue
                if id(d) not in cache:
                    bad = set(d) - set(self.tree.node_to_index)
                    if bad:
                        raise ValueError('some of the nodes which have been annotated with '
                                         'state restrictions are not even in the tree: '
                                         + str(sorted(bad)))
                    row = np.full(N, full, dtype=np.uint64)
                    for v, allowed in d.items():
                        m = 0
                        for s in allowed:
                            if not 0 <= int(s) < n:
                                raise ValueError('state %r outside [0, %d)' % (s, n))
                            m |= 1 << int(s)
                        row[self.tree.node_to_index[v]] = m
                    cache[id(d)] = row
                masks[c] = cache[id(d)]
        self.node_masks = masks & full
        self.nchains = masks.shape[0]
        self.root_distn = None
        if root_distn is not None:
            rd = np.ascontiguousarray(root_distn, dtype=np.float64)
            if rd.shape != (n,):
                raise ValueError('root shape mismatch: %s %s' % ((n,), rd.shape))
            self.root_distn = rd
        self.ctx = ctx if ctx is not None else get_context()
        self.seed = int(seed)
        self.rng = np.random.Generator(np.random.PCG64(self.seed))
        self.nsweeps = 0
        self.last_chunks = 0
        self.device_seconds = 0.0            # time inside the device call, all sweeps
        if self.tree.nnodes == 1:
            raise ValueError('the tree has no edges')

    # -- device step ------------------------------------------------------------------
    def _resample(self, chain, edge, length):
        """Steps 3-5 for rows without states: returns their states and sets node_states."""
        C, N, n = self.nchains, self.tree.nnodes, self.nstates
        offset, cparent, piece, node = chunk_forest(self.parent, C, chain, edge)
        total = int(offset[-1])
        masks = np.full(total, np.uint64((1 << n) - 1), dtype=np.uint64)
        glob = offset[:-1, None] + node
        for v in range(N):               # one base node at a time: distinct chunks per chain
            masks[glob[:, v]] &= self.node_masks[:, v]
        states = np.empty(total, dtype=np.int32)
        status = np.empty(C, dtype=np.int32)
        rd = self.root_distn
        t0 = time.perf_counter()
        _lib.check(_lib.lib().rt_forest_resample_states_parents(
            self.ctx._h, n, C, _ptr(offset, c_int64), _ptr(cparent, c_int32),
            _ptr(self.P, c_double), None if rd is None else _ptr(rd, c_double),
            _ptr(masks, c_uint64), c_uint64(self.seed & (2 ** 64 - 1)),
            c_uint64(self.nsweeps & (2 ** 64 - 1)), _ptr(states, c_int32),
            _ptr(status, c_int32)))
        self.device_seconds += time.perf_counter() - t0
        self.last_chunks = total
        self.nsweeps += 1
        node_states = states[offset[:-1, None] + node]
        return states[piece], node_states, status

    def _init_feasible(self):
        C, N = self.nchains, self.tree.nnodes
        if N == 1:
            raise ValueError('the tree has no edges')
        for k in range(0, 8):
            per = 2 ** k                                     # pieces per edge
            if per - 1 > self.nstates and k > 0:
                break
            chain = np.repeat(np.arange(C, dtype=np.int64), (N - 1) * per)
            edge = np.tile(np.repeat(np.arange(1, N, dtype=np.int64), per), C)
            length = self.branch[edge] / per
            state, node_states, status = self._resample(chain, edge, length)
            if not status.any():
                self.node_states = node_states
                self.chain, self.edge, self.length, self.state = merge_segments(
                    chain, edge, length, state.astype(np.int64))
                return
        bad = int(np.nonzero(status)[0][0])
        raise StructuralZeroProb('failed to find a feasible history for chain %d' % bad)

    # -- one sweep ----------------------------------------------------------------------
    def sweep(self):
        """One Rao-Teh sweep of every chain (_sampler.py:366-390)."""
        rep, sub = poisson_split(self.rng, self.length, self.poisson_rates[self.state])
        chain = np.repeat(self.chain, rep)
        edge = np.repeat(self.edge, rep)
        state, node_states, status = self._resample(chain, edge, sub)
        if status.any():
            bad = int(np.nonzero(status)[0][0])
            raise StructuralZeroProb('chain %d: the chunk tree has zero likelihood' % bad)
        self.node_states = node_states
        self.chain, self.edge, self.length, self.state = merge_segments(
            chain, edge, sub, state.astype(np.int64))
        return self

    # -- summaries ----------------------------------------------------------------------
    def dwell_times(self):
        """f64[nchains, nstates]: time spent in each state (rows sum to the tree length)."""
        out = np.bincount(self.chain * self.nstates + self.state, weights=self.length,
                          minlength=self.nchains * self.nstates)
        return out.reshape(self.nchains, self.nstates)

    def transition_counts(self):
        """int64[nchains, nstates, nstates]: transitions a -> b along the edges."""
        n = self.nstates
        inner = np.zeros(self.chain.shape[0], dtype=bool)
        inner[1:] = (self.chain[1:] == self.chain[:-1]) & (self.edge[1:] == self.edge[:-1])
        at = np.nonzero(inner)[0]
        key = (self.chain[at] * n + self.state[at - 1]) * n + self.state[at]
        return np.bincount(key, minlength=self.nchains * n * n).reshape(self.nchains, n, n)

    def root_states(self):
        return self.node_states[:, 0].copy()

    def trajectory_log_likelihoods(self):
        """f64[nchains]: log density of each history under the batch's own process."""
        return trajectory_log_likelihoods(self.dwell_times(), self.transition_counts(),
                                          self.root_states(), self.Q, self.root_distn)

    def snapshot(self):
        self._snap = (self.chain.copy(), self.edge.copy(), self.length.copy(), self.state.copy(),
                      self.node_states.copy())

    def restore(self, reject):
        """The chains flagged in ``reject`` (bool[nchains]) return to the snapshot."""
        reject = np.asarray(reject, dtype=bool)
        if not reject.any():
            return
        sc, se, sl, ss, sn = self._snap
        keep = ~reject[self.chain]
        back = reject[sc]
        chain = np.concatenate([self.chain[keep], sc[back]])
        order = np.argsort(chain, kind='stable')          # rows of a chain come from one side
        self.chain = chain[order]
        self.edge = np.concatenate([self.edge[keep], se[back]])[order]
        self.length = np.concatenate([self.length[keep], sl[back]])[order]
        self.state = np.concatenate([self.state[keep], ss[back]])[order]
        self.node_states = np.where(reject[:, None], sn, self.node_states)

    def mh_sweep(self, target, cache=None):
        """Metropolis-Hastings step towards ``target(batch) -> log density per chain``."""
        if cache is None:
            cache = self.__dict__.setdefault('_mh_cache', {})
        return _mh_step(self, target, self.rng, cache)

    def history(self, c=0):
        """Chain c as the reference yields it: an undirected nx tree whose edges carry
        'weight' and 'state'; the nodes of T keep their ids, event nodes are numbered from
        max(T) + 1 (_sample_mjp_dense.py:43-44)."""
        nodes = self.tree.preorder_nodes
        nxt = max(nodes) + 1
        sel = self.chain == c
        edge, length, state = self.edge[sel], self.length[sel], self.state[sel]
        out = nx.Graph()
        out.add_node(nodes[0])
        i = 0
        while i < edge.shape[0]:
            j = i
            while j < edge.shape[0] and edge[j] == edge[i]:
                j += 1
            v = int(edge[i])
            prev = nodes[int(self.parent[v])]
            for r in range(i, j):
                if r == j - 1:
                    nb = nodes[v]
                else:
                    nb = nxt
                    nxt += 1
                out.add_edge(prev, nb, weight=float(length[r]), state=int(state[r]))
                prev = nb
            i = j
        return out


class DeviceHistoryBatch(object):
    """The same batch with the histories resident on the device (csrc/forest.hip,
    ``rt_chains_*``): Poisson events, chunk trees, posterior draws and the removal of self
    transitions all run as kernels; a sweep moves two words to the host (row total, status
    flag).  Same constructor arguments, summaries and ``history(c)`` as HistoryBatch; the
    draws are counter-based on the device, so the two classes do not produce the same
    histories from the same seed -- they sample the same distribution."""

    def __init__(self, T, root, Q, node_masks=None, node_to_allowed_states=None, nchains=None,
                 root_distn=None, uniformization_factor=2, seed=0, ctx=None):
        # argument handling shared with the host batch (no device work in there)
        proto = HistoryBatch.__new__(HistoryBatch)
        proto._setup(T, root, Q, node_masks, node_to_allowed_states, nchains, root_distn,
                     uniformization_factor, seed, ctx)
        self.__dict__.update(proto.__dict__)
        parent = np.ascontiguousarray(self.parent, dtype=np.int32)
        masks = np.ascontiguousarray(self.node_masks, dtype=np.uint64)
        rd = self.root_distn
        handle = ctypes.c_void_p()
        code = _lib.lib().rt_chains_create(
            self.ctx._h, self.tree.nnodes, _ptr(parent, c_int32), _ptr(self.branch, c_double),
            self.nstates, _ptr(self.P, c_double), _ptr(self.poisson_rates, c_double),
            None if rd is None else _ptr(rd, c_double), self.nchains, _ptr(masks, c_uint64),
            c_uint64(self.seed & (2 ** 64 - 1)), ctypes.byref(handle))
        if code == _lib.RT_ERR_ZERO_PROB:
            raise StructuralZeroProb(_lib.last_error())
        _lib.check(code)
        self._h = handle
        self.nsweeps = 0
        # the C object refers to its context: the context closes its chain batches before it
        # goes (device.Context.close), whatever order the garbage collector picks
        self.ctx._children.add(self)

    def close(self):
        from .device import _shutting_down
        h = getattr(self, '_h', None)
        if h and not _shutting_down:
            _lib.lib().rt_chains_destroy(h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sweep(self, nsweeps=1):
        before = self.sizes()[2]
        code = _lib.lib().rt_chains_sweep(self._h, int(nsweeps))
        self.nsweeps += self.sizes()[2] - before          # the sweeps that completed
        if code == _lib.RT_ERR_ZERO_PROB:
            # what HistoryBatch.sweep and the reference raise for the same condition
            # (_sample_mc0_dense.py:57-62)
            raise StructuralZeroProb(_lib.last_error())
        _lib.check(code)
        return self

    def sizes(self):
        """(rows of all histories, chunks of the last sweep, device steps so far)."""
        rows, chunks, sweeps = c_int64(0), c_int64(0), c_int64(0)
        _lib.check(_lib.lib().rt_chains_get_sizes(self._h, ctypes.byref(rows),
                                                  ctypes.byref(chunks), ctypes.byref(sweeps)))
        return rows.value, chunks.value, sweeps.value

    @property
    def last_chunks(self):
        return self.sizes()[1]

    def dwell_times(self):
        out = np.empty((self.nchains, self.nstates), dtype=np.float64)
        _lib.check(_lib.lib().rt_chains_get_statistics(self._h, _ptr(out, c_double), None, None))
        return out

    def transition_counts(self):
        out = np.empty((self.nchains, self.nstates, self.nstates), dtype=np.int64)
        _lib.check(_lib.lib().rt_chains_get_statistics(self._h, None, _ptr(out, c_int64), None))
        return out

    @property
    def node_states(self):
        out = np.empty((self.nchains, self.tree.nnodes), dtype=np.int32)
        _lib.check(_lib.lib().rt_chains_get_statistics(self._h, None, None, _ptr(out, c_int32)))
        return out

    def root_states(self):
        return self.node_states[:, 0].copy()

    def trajectory_log_likelihoods(self):
        return trajectory_log_likelihoods(self.dwell_times(), self.transition_counts(),
                                          self.root_states(), self.Q, self.root_distn)

    def snapshot(self):
        _lib.check(_lib.lib().rt_chains_snapshot(self._h))

    def restore(self, reject):
        """Undo the ONE sweep since the snapshot for the chains flagged in ``reject``."""
        reject = np.ascontiguousarray(reject, dtype=np.uint8)
        if reject.shape != (self.nchains,):
            raise ValueError('one flag per chain expected')
        _lib.check(_lib.lib().rt_chains_restore(self._h, _ptr(reject, ctypes.c_ubyte)))

    def mh_sweep(self, target, cache=None):
        """Metropolis-Hastings step towards ``target(batch) -> log density per chain``."""
        if cache is None:
            cache = self.__dict__.setdefault('_mh_cache', {})
        if not hasattr(self, 'rng'):
            self.rng = np.random.Generator(np.random.PCG64(self.seed))
        return _mh_step(self, target, self.rng, cache)

    def rows(self):
        """(chain int64[S], edge int64[S], length f64[S], state int64[S]) of all histories."""
        total = self.sizes()[0]
        off = np.empty(self.nchains + 1, dtype=np.int64)
        edge = np.empty(max(total, 1), dtype=np.int32)
        length = np.empty(max(total, 1), dtype=np.float64)
        state = np.empty(max(total, 1), dtype=np.int32)
        _lib.check(_lib.lib().rt_chains_get_rows(self._h, total, _ptr(off, c_int64),
                                                 _ptr(edge, c_int32), _ptr(length, c_double),
                                                 _ptr(state, c_int32)))
        chain = np.repeat(np.arange(self.nchains, dtype=np.int64), np.diff(off))
        return (chain, edge[:total].astype(np.int64), length[:total],
                state[:total].astype(np.int64))

    def history(self, c=0):
        self.chain, self.edge, self.length, self.state = self.rows()
        try:
            return HistoryBatch.history(self, c)
        finally:
            del self.chain, self.edge, self.length, self.state


# ---------------------------------------------------------------------------
# the reference's generators
# ---------------------------------------------------------------------------

def _dense_problem(Q, node_to_allowed_states, root_distn):
    """Sparse (nx.DiGraph rate matrix, dict root distribution, arbitrary state labels:
    _sampler.py:300-352) -> dense arrays over sorted(states); returns the label list."""
    if isinstance(Q, np.ndarray):
        check_square_dense(Q)
        return Q, node_to_allowed_states, root_distn, None
    if not Q:
        raise ValueError('the rate matrix is empty')
    for a, b in Q.edges():
        if a == b:
            raise ValueError('the rate matrix should have no loops')
    labels = sorted(Q)
    index = dict((s, i) for i, s in enumerate(labels))
    dense = np.zeros((len(labels), len(labels)))
    for a, b, d in Q.edges(data=True):
        dense[index[a], index[b]] = d['weight']
    dense -= np.diag(dense.sum(axis=1))
    allowed = dict((v, set(index[s] for s in ss if s in index))
                   for v, ss in node_to_allowed_states.items())
    rd = None
    if root_distn is not None:
        rd = np.zeros(len(labels))
        for s, p in root_distn.items():
            if s in index:
                rd[index[s]] = p
    return dense, allowed, rd, labels


def gen_restricted_histories(T, Q, node_to_allowed_states, root, root_distn=None,
                             uniformization_factor=2, nhistories=None, seed=0, ctx=None):
    """raoteh.sampler._sampler.gen_restricted_histories (:300-390): yields trees whose
    edges carry 'weight' and 'state'; the first one is the feasible starting history.
    ``Q``: dense ndarray, or the reference's nx.DiGraph of rates without loops."""
    bad = set(node_to_allowed_states) - set(T)
    if bad:
        raise ValueError('some of the nodes which have been annotated with state restrictions '
                         'are not even in the tree: ' + str(sorted(bad)))
    dense, allowed, rd, labels = _dense_problem(Q, node_to_allowed_states, root_distn)
    batch = DeviceHistoryBatch(T, root, dense, node_to_allowed_states=allowed, nchains=1,
                               root_distn=rd, uniformization_factor=uniformization_factor,
                               seed=seed, ctx=ctx)
    count = 0
    while True:
        h = batch.history(0)
        if labels is not None:
            for a, b, d in h.edges(data=True):
                d['state'] = labels[d['state']]
        yield h
        count += 1
        if nhistories is not None and count >= nhistories:
            return
        batch.sweep()


def gen_histories(T, Q, node_to_state, root=None, root_distn=None, uniformization_factor=2,
                  nhistories=None, seed=0, ctx=None):
    """raoteh.sampler._sampler.gen_histories (:238-297): known states at some nodes."""
    if root is None:        # a node of known state if there is one (:268-275)
        root = next(iter(node_to_state)) if node_to_state else next(iter(T))
    allowed = dict((v, {s}) for v, s in node_to_state.items())
    for h in gen_restricted_histories(T, Q, allowed, root, root_distn=root_distn,
                                      uniformization_factor=uniformization_factor,
                                      nhistories=nhistories, seed=seed, ctx=ctx):
        yield h


def gen_mh_histories(T, Q, node_to_allowed_states, target_log_likelihood_callback, root,
                     root_distn=None, uniformization_factor=2, nhistories=None, seed=0, ctx=None):
    """raoteh.sampler._sampler.gen_mh_histories (:393-551): Rao-Teh proposals under ``Q``
    corrected by Metropolis-Hastings to the density ``target_log_likelihood_callback(T_aug)``
    gives for a history (an nx tree with 'weight' and 'state' on its edges).  Yields
    (history, accepted): a rejected proposal yields the previous history again."""
    bad = set(node_to_allowed_states) - set(T)
    if bad:
        raise ValueError('some of the nodes which have been annotated with state restrictions '
                         'are not even in the tree: ' + str(sorted(bad)))
    dense, allowed, rd, labels = _dense_problem(Q, node_to_allowed_states, root_distn)
    batch = DeviceHistoryBatch(T, root, dense, node_to_allowed_states=allowed, nchains=1,
                               root_distn=rd, uniformization_factor=uniformization_factor,
                               seed=seed, ctx=ctx)

    def labelled(b):
        h = b.history(0)
        if labels is not None:
            for _, _, d in h.edges(data=True):
                d['state'] = labels[d['state']]
        return h

    def target(b):
        return np.array([target_log_likelihood_callback(labelled(b))], dtype=np.float64)

    yield labelled(batch), True
    count = 1
    while nhistories is None or count < nhistories:
        accepted = bool(batch.mh_sweep(target)[0])
        yield labelled(batch), accepted
        count += 1

