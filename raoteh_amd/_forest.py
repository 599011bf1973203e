"""
The device core of a Rao-Teh sweep: a ragged batch of trees -- the chunk trees of many
independent chains / sites (raoteh/sampler/_graph_transform.py:298-375), each with its
own topology -- re-sampled with ONE shared transition matrix, the uniformized
P = I + Q / omega (_sample_mjp_dense.py:72-114).

    forest = Forest([(T0, root0), (T1, root1), ...])       host: CSR per tree, concatenated
    sets, pmaps = get_node_to_set_and_pmap(forest, P, allowed)
    states, status = resample_states(forest, P, allowed, root_distn, seed=.., sweep=..)

Reference functions replaced, per tree and with P_default = P on every edge:
_mcy.get_node_to_pset / get_node_to_set with one matrix (_mcy.py:139-181, 240-271;
pyfelscore.mcy_get_node_to_pset, pyfelscore.get_node_to_set), _mcy.get_node_to_pmap
(:563-607) and _sample_mc0(_dense).resample_states (_sample_mc0_dense.py:20-98) as
_sample_mcy.resample_states chains them (_sample_mcy.py:19-83).  The kernels are in
csrc/forest.hip; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import c_double, c_int32, c_int64, c_uint64

import networkx as nx
import numpy as np

from . import _lib
from ._tree import TreeArrays
from ._util import StructuralZeroProb
from .device import get_context

__all__ = ['Forest', 'get_node_to_set_and_pmap', 'resample_states']


def _ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


class Forest(object):
    """``trees``: list of (T, root), T an undirected nx tree with integer nodes.  Every
    tree is put into DFS preorder and children-CSR form (``_tree.TreeArrays``, the
    layout of _density.digraph_to_bool_csr) and the pieces are concatenated:
    ``node_offset`` int64[ntrees + 1], ``indices`` int64[total - ntrees], ``indptr``
    int64[total + ntrees]; ``preorder[k]`` lists tree k's nodes in device order."""

    def __init__(self, trees):
        self.preorder, offs, idx, ptr = [], [0], [], []
        for T, root in trees:
            if root not in T:
                raise ValueError('the specified root is not in the tree')
            ta = TreeArrays(T, root)
            self.preorder.append(list(ta.preorder_nodes))
            offs.append(offs[-1] + ta.nnodes)
            idx.append(np.asarray(ta.indices, dtype=np.int64))
            ptr.append(np.asarray(ta.indptr, dtype=np.int64))
        self.ntrees = len(self.preorder)
        if not self.ntrees:
            raise ValueError('an empty forest')
        self.node_offset = np.array(offs, dtype=np.int64)
        self.indices = (np.concatenate(idx) if idx else np.zeros(0, np.int64)).astype(np.int64)
        if self.indices.size == 0:
            self.indices = np.zeros(1, dtype=np.int64)          # a valid pointer
        self.indptr = np.concatenate(ptr).astype(np.int64)
        self.total = int(offs[-1])

    def allowed_masks(self, node_to_allowed_states, nstates):
        """list (one dict per tree, or None) of node -> allowed set; a node that is
        missing, or a None dict, is unrestricted -> uint64[total] bit masks."""
        if nstates > 64:
            raise ValueError('the forest passes hold a state per lane: nstates <= 64')
        full = (1 << nstates) - 1
        out = np.empty(self.total, dtype=np.uint64)
        for k, nodes in enumerate(self.preorder):
            d = node_to_allowed_states[k] if node_to_allowed_states is not None else None
            lo = int(self.node_offset[k])
            for i, v in enumerate(nodes):
                m = full
                if d is not None and v in d:
                    m = 0
                    for s in d[v]:
                        if not 0 <= int(s) < nstates:
                            raise ValueError('state %r outside [0, %d)' % (s, nstates))
                        m |= 1 << int(s)
                out[lo + i] = m
        return out

    def split(self, flat):
        """Per-node array [total, ...] -> list of {node: row} dicts."""
        out = []
        for k, nodes in enumerate(self.preorder):
            lo = int(self.node_offset[k])
            out.append(dict((v, flat[lo + i]) for i, v in enumerate(nodes)))
        return out


def _matrix(P, nstates):
    P = np.ascontiguousarray(P, dtype=np.float64)
    if P.ndim != 2 or P.shape != (nstates, nstates):
        raise ValueError('expected a %d x %d transition matrix' % (nstates, nstates))
    return P


def get_node_to_set_and_pmap(forest, P, node_to_allowed_states=None, ctx=None):
    """For every tree: ({node: set of states with positive posterior probability},
    {node: f64[nstates] subtree likelihoods}) -- pset, set and pmap of the reference
    with P on every edge.  Returns two lists (one entry per tree)."""
    ctx = ctx if ctx is not None else get_context()
    n = np.asarray(P).shape[0]
    P = _matrix(P, n)
    masks = forest.allowed_masks(node_to_allowed_states, n)
    L = np.empty((forest.total, n), dtype=np.float64)
    _lib.check(_lib.lib().rt_forest_passes(
        ctx._h, n, forest.ntrees, _ptr(forest.node_offset, c_int64),
        _ptr(forest.indices, c_int64), _ptr(forest.indptr, c_int64), _ptr(P, c_double),
        _ptr(masks, c_uint64), _ptr(L, c_double)))
    sets = [dict((v, set(s for s in range(n) if (int(m) >> s) & 1)) for v, m in d.items())
            for d in forest.split(masks)]
    return sets, forest.split(L)


def resample_states(forest, P, node_to_allowed_states=None, root_distn=None, seed=0,
                    sweep=0, ctx=None, return_status=False):
    """One posterior draw of a state for every node of every tree
    (_sample_mcy.resample_states, :19-83, per tree): list of {node: state}.  ``seed`` keys
    the counter-based generator, ``sweep`` is the stream within it: the same (seed, sweep,
    forest, P, observations) gives the same states.  A tree whose likelihood is zero
    raises StructuralZeroProb as the reference does -- unless ``return_status``, then
    (states, status int32[ntrees]) comes back with -1 states on such trees."""
    ctx = ctx if ctx is not None else get_context()
    n = np.asarray(P).shape[0]
    P = _matrix(P, n)
    masks = forest.allowed_masks(node_to_allowed_states, n)
    rd = None
    if root_distn is not None:
        rd = np.ascontiguousarray(root_distn, dtype=np.float64)
        if rd.shape != (n,):
            raise ValueError('root shape mismatch: %s %s' % ((n,), rd.shape))
    states = np.empty(forest.total, dtype=np.int32)
    status = np.empty(forest.ntrees, dtype=np.int32)
    _lib.check(_lib.lib().rt_forest_resample_states(
        ctx._h, n, forest.ntrees, _ptr(forest.node_offset, c_int64),
        _ptr(forest.indices, c_int64), _ptr(forest.indptr, c_int64), _ptr(P, c_double),
        None if rd is None else _ptr(rd, c_double), _ptr(masks, c_uint64),
        c_uint64(int(seed) & (2 ** 64 - 1)), c_uint64(int(sweep) & (2 ** 64 - 1)),
        _ptr(states, c_int32), _ptr(status, c_int32), None))
    out = [dict((v, int(s)) for v, s in d.items()) for d in forest.split(states)]
    if return_status:
        return out, status
    if status.any():
        k = int(np.nonzero(status)[0][0])
        raise StructuralZeroProb('tree %d of the forest has zero likelihood' % k)
    return out
