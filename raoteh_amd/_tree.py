"""
networkx tree -> flat arrays, exactly as the reference marshals them for its
native passes:

* BFS-directed tree + DFS preorder node list   raoteh/sampler/_mcy_dense.py:246-255
* children CSR in preorder index space          raoteh/sampler/_density.py:104-140
* per-edge matrices keyed by the CHILD index    raoteh/sampler/_density.py:143-180
"""
from __future__ import annotations

import numpy as np

__all__ = ['TreeArrays', 'marshal_tree', 'check_square_dense']


def check_square_dense(M):
    """Same checks and messages as raoteh/sampler/_density.py:79-101."""
    if M is None:
        raise ValueError('the matrix is None')
    try:
        shape = M.shape
    except AttributeError:
        if hasattr(M, 'number_of_nodes'):
            raise ValueError('expected an ndarray but found a graph object')
        raise ValueError('expected an ndarray')
    if len(shape) != 2:
        raise ValueError('expected len(M.shape) == 2')
    if shape[0] != shape[1]:
        raise ValueError('expected the array to be square')


class TreeArrays(object):
    """preorder_nodes, node_to_index, indices/indptr (int64 CSR of children),
    parent (int64, -1 at the root) and the nx edge data dict above each node."""

    def __init__(self, T, root):
        if root not in T:
            raise ValueError('the specified root is not in the tree')
        self.root = root
        # one depth-first walk in adjacency order: the preorder of
        # nx.dfs_preorder_nodes and, per node, the child order of nx.bfs_edges
        # (both follow the adjacency order; _density.py:104-140, _mcy_dense.py:255)
        adj = T._adj if hasattr(T, '_adj') else T.adj
        preorder, parent_of, children = [root], {root: None}, {root: []}
        stack = [(root, iter(adj[root]))]
        while stack:
            node, it = stack[-1]
            for nb in it:
                if nb in parent_of:
                    if nb != parent_of[node]:
                        raise ValueError('the graph is not a tree')
                    continue
                parent_of[nb] = node
                children[node].append(nb)
                children[nb] = []
                preorder.append(nb)
                stack.append((nb, iter(adj[nb])))
                break
            else:
                stack.pop()
        self.preorder_nodes = preorder
        n = len(preorder)
        if n != T.number_of_nodes():
            raise ValueError('the number of nodes is inconsistent')
        if T.number_of_edges() != n - 1:
            raise ValueError('the graph is not a tree')
        self.node_to_index = dict((v, i) for i, v in enumerate(preorder))
        self.parent = np.full(n, -1, dtype=np.int64)
        self.edge_data = [None] * n
        for ib in range(1, n):
            nb = preorder[ib]
            na = parent_of[nb]
            self.parent[ib] = self.node_to_index[na]
            self.edge_data[ib] = adj[na][nb]
        indices = []
        indptr = [0]
        for na in preorder:
            indices.extend(self.node_to_index[nb] for nb in children[na])
            indptr.append(len(indices))
        self.indices = np.array(indices, dtype=np.int64)
        self.indptr = np.array(indptr, dtype=np.int64)
        self.nnodes = n

    def branch_lengths(self):
        """f64[nnodes]: weight of the edge above each node (0 at the root)."""
        t = np.zeros(self.nnodes, dtype=np.float64)
        for i in range(1, self.nnodes):
            t[i] = self.edge_data[i]['weight']
        return t

    def rate_matrices(self, nstates, Q_default=None):
        """(Q f64[nq,n,n], node_q int64[nnodes]) from the per-edge 'Q'
        attribute with Q_default as fallback (_mjp_dense.py:355-356).
        Identical matrix objects share one slot."""
        slots = {}
        mats = []
        node_q = np.zeros(self.nnodes, dtype=np.int64)
        for i in range(1, self.nnodes):
            Q = self.edge_data[i].get('Q', Q_default)
            key = id(Q)
            if key not in slots:            # every distinct matrix is checked once
                check_square_dense(Q)
                if Q.shape[0] != nstates:
                    raise ValueError('rate matrix shape %s does not match nstates '
                                     '%d' % (Q.shape, nstates))
                slots[key] = len(mats)
                mats.append(np.ascontiguousarray(Q, dtype=np.float64))
            node_q[i] = slots[key]
        if not mats:
            mats.append(np.zeros((nstates, nstates)))
        return np.stack(mats), node_q

    def esd_transitions(self, nstates, P_default=None):
        """f64[nnodes,n,n] of the edges' 'P' (_density.py:143-180)."""
        esd = np.zeros((self.nnodes, nstates, nstates), dtype=np.float64)
        for i in range(1, self.nnodes):
            P = self.edge_data[i].get('P', P_default)
            check_square_dense(P)
            esd[i] = P
        return esd


def marshal_tree(T, root):
    return TreeArrays(T, root)
