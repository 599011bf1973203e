"""
CPU-side tests of the product's host logic (no GPU, no compute calls):
* the C-ABI library loads and exports every symbol include/raoteh_hip.h declares
* the post-order schedule the kernels interpret (rt_build_schedule) is a valid
  stack program: emulating it in numpy reproduces the oracle's likelihoods
* tree marshalling matches the oracle's restatement of the reference layout
* the product package never imports the oracle
"""
import ctypes
import itertools
import os
import re

import networkx as nx
import numpy as np
import pytest

from conftest import ROOT, load_golden, tree_from_edges
from oracle import oracle_numpy as orc
from raoteh_amd import _lib, synth
from raoteh_amd._tree import TreeArrays


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'raoteh_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(rt_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 30
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(_lib.SIGNATURES), (
        declared.symmetric_difference(set(_lib.SIGNATURES)))
    assert L.rt_version() >= 100


def test_no_gpu_fails_loudly():
    L = _lib.lib()
    n = ctypes.c_int(0)
    rc = L.rt_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip('a GPU is present')
    h = ctypes.c_void_p()
    assert L.rt_ctx_create(0, ctypes.byref(h)) < 0
    from raoteh_amd.device import Context
    with pytest.raises(Exception):
        Context(0)


def build_schedule(indices, indptr):
    L = _lib.lib()
    nnodes = len(indptr) - 1
    ops = np.zeros((nnodes, 4), dtype=np.int32)
    depth = ctypes.c_int32(0)
    idx = np.ascontiguousarray(indices, dtype=np.int64)
    ptr = np.ascontiguousarray(indptr, dtype=np.int64)
    rc = L.rt_build_schedule(
        nnodes, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
        ptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
        ops.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(depth))
    assert rc == 0, _lib.last_error()
    return ops, depth.value


def emulate(ops, depth, esd, obs_nodes, obs, root_w):
    """What the kernels do with the schedule, in numpy (all sites at once)."""
    nsites, n = obs.shape[0], esd.shape[1]
    slot_of = dict((int(v), k) for k, v in enumerate(obs_nodes))
    stack = [None] * max(depth, 1)
    lik = None
    for node, _, pop, dst in ops:
        x = np.ones((nsites, n))
        if pop >= 0:
            assert stack[pop] is not None
            x = stack[pop]
            stack[pop] = None
        if node in slot_of:
            x = x * obs[:, slot_of[node], :]
        if dst < 0:
            lik = np.maximum(x, 0) @ root_w
            break
        t = x @ esd[node].T
        slot, first = dst & 255, dst >> 8
        if first:
            assert stack[slot] is None
            stack[slot] = t
        else:
            stack[slot] = stack[slot] * t
    assert all(s is None for s in stack)
    return lik


@pytest.mark.parametrize('seed', range(6))
def test_schedule_is_a_valid_stack_program(seed):
    rng = np.random.RandomState(seed)
    if seed == 0:
        T, root, leaves = synth.balanced_tree(64, seed=0)
    elif seed == 1:   # caterpillar: deep tree, shallow stack
        T = nx.Graph()
        for i in range(40):
            T.add_edge(2 * i, 2 * i + 2, weight=0.1)
            T.add_edge(2 * i, 2 * i + 1, weight=0.2)
        root, leaves = 0, [2 * i + 1 for i in range(40)] + [80]
    else:
        T, root, leaves = synth.random_tree(int(rng.randint(2, 40)), seed=seed,
                                            max_children=4)
    n = int(rng.randint(2, 7))
    ta = TreeArrays(T, root)
    pre, idx, ptr = orc.tree_to_arrays(T, root)
    assert pre == ta.preorder_nodes
    np.testing.assert_array_equal(idx, ta.indices)
    np.testing.assert_array_equal(ptr, ta.indptr)
    ops, depth = build_schedule(ta.indices, ta.indptr)
    nleaves = sum(1 for v in T if T.degree(v) <= 1 or
                  (v != root and T.degree(v) == 1))
    assert depth <= int(np.floor(np.log2(max(nleaves, 1)))) + 2
    if seed == 0:
        assert depth == 6
    if seed == 1:
        assert depth <= 2
    assert sorted(ops[:, 0].tolist()) == list(range(ta.nnodes))   # every node once
    assert ops[-1, 0] == 0 and ops[-1, 3] < 0                      # root last
    esd = rng.uniform(0.0, 1.0, size=(ta.nnodes, n, n))
    esd[0] = 0
    obs_nodes = [ta.node_to_index[v] for v in leaves]
    obs = rng.uniform(0.1, 1.0, size=(5, len(obs_nodes), n))
    w = rng.uniform(0.1, 1.0, size=n)
    want, _ = orc.batch_upward(ta.indices, ta.indptr, esd, obs_nodes, obs, w)
    got = emulate(ops, depth, esd, obs_nodes, obs, w)
    np.testing.assert_allclose(got, want, rtol=1e-13)


def test_single_node_schedule():
    ops, depth = build_schedule(np.zeros(0, dtype=np.int64),
                                np.array([0, 0], dtype=np.int64))
    assert ops.tolist() == [[0, -1, -1, -1]] and depth == 0


def test_marshal_rate_matrices_and_esd():
    fx = load_golden('config_c5')
    T = tree_from_edges(fx['edges'])
    for na, nb in nx.bfs_edges(T, fx['root']):
        T[na][nb]['Q'] = np.array(fx['Q_edges'][str(nb)])
    ta = TreeArrays(T, fx['root'])
    Q, node_q = ta.rate_matrices(fx['nstates'])
    assert Q.shape == (ta.nnodes - 1, 20, 20)
    for i in range(1, ta.nnodes):
        np.testing.assert_array_equal(
            Q[node_q[i]], np.array(fx['Q_edges'][str(ta.preorder_nodes[i])]))
    with pytest.raises(ValueError):
        TreeArrays(T, 12345)
    T2 = tree_from_edges(fx['edges'])
    with pytest.raises(ValueError):          # no Q anywhere
        TreeArrays(T2, fx['root']).rate_matrices(20, None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'raoteh_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, re.M), f
                assert 'oracle_numpy' not in text, f


def test_tree_specialised_source_is_generated_on_the_host():
    """rt_jit_source (csrc/jit.hip) needs no device: one arithmetic block per
    schedule step, one load pair per observed node, every P element referenced
    at its step-ordered offset."""
    import ctypes
    import re
    from raoteh_amd import _lib, synth
    from raoteh_amd._tree import TreeArrays
    T, root, leaves = synth.balanced_tree(8, seed=0)
    ta = TreeArrays(T, root)
    obs = np.array([ta.node_to_index[v] for v in leaves], dtype=np.int64)
    p64 = ctypes.POINTER(ctypes.c_int64)
    buf = ctypes.create_string_buffer(1 << 20)
    for n in (2, 3, 4):
        _lib.check(_lib.lib().rt_jit_source(
            ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
            n, len(obs), obs.ctypes.data_as(p64), 3, buf, len(buf)))
        src = buf.value.decode()
        assert src.count('// step ') == ta.nnodes
        assert 'extern "C" __global__' in src and 'rt_jit_prune' in src
        hp = (n + 1) // 2
        # (a pair of which only .x is used -- odd n -- is an 8-byte load: '{load(...), 0.0}')
        assert len(re.findall(r'const rt_d2 o\d+_\d+ = [^;]*g\[', src)) == len(obs) * hp
        assert src.count('), 0.0};') == (len(obs) if n % 2 else 0)
        # P records of the 14 non-root steps, n*n registers each
        assert len(set(re.findall(r'\bp(\d+)_0\b', src))) == ta.nnodes - 1
    # too small a buffer is an error, not a truncation
    small = ctypes.create_string_buffer(64)
    rc = _lib.lib().rt_jit_source(
        ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
        4, len(obs), obs.ctypes.data_as(p64), 3, small, len(small))
    assert rc < 0


def test_tree_specialised_mfma_source_is_generated_on_the_host(monkeypatch):
    """MFMA families: 4 < n <= 32: NT * KS MFMAs per step and tile, T tiles per
    wave, one A-fragment load per (step, row tile, k-pair) whatever T is."""
    import ctypes
    import re
    from raoteh_amd import _lib, synth
    from raoteh_amd._tree import TreeArrays
    T, root, leaves = synth.balanced_tree(8, seed=0)
    ta = TreeArrays(T, root)
    obs = np.array([ta.node_to_index[v] for v in leaves], dtype=np.int64)
    p64 = ctypes.POINTER(ctypes.c_int64)
    buf = ctypes.create_string_buffer(1 << 22)
    # 4x4x4-block variant (the default where its MFMA count is below the 16x16x4
    # family's padded one): KS * KS block MFMAs per step and tile
    monkeypatch.setenv('RAOTEH_JIT_QUAD', '1')
    for n, tiles in ((5, 1), (20, 2), (32, 2)):
        monkeypatch.setenv('RAOTEH_JIT_TILES', str(tiles))
        _lib.check(_lib.lib().rt_jit_source(
            ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
            n, len(obs), obs.ctypes.data_as(p64), 2, buf, len(buf)))
        src = buf.value.decode()
        ks = (n + 3) // 4
        assert src.count('__builtin_amdgcn_mfma_f64_4x4x4f64') == (ta.nnodes - 1) * ks * ks * tiles
        assert '__builtin_amdgcn_mfma_f64_16x16x4f64' not in src
    monkeypatch.setenv('RAOTEH_JIT_QUAD', '0')
    for n, tiles in ((5, 1), (20, 3), (32, 2)):
        monkeypatch.setenv('RAOTEH_JIT_TILES', str(tiles))
        _lib.check(_lib.lib().rt_jit_source(
            ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
            n, len(obs), obs.ctypes.data_as(p64), 2, buf, len(buf)))
        src = buf.value.decode()
        nt, ks = (n + 15) // 16, (n + 3) // 4
        kp = (ks + 1) // 2
        steps = ta.nnodes - 1                      # the root step has no product
        assert src.count('__builtin_amdgcn_mfma_f64_16x16x4f64') == steps * nt * ks * tiles
        assert len(re.findall(r'const rt_d2 A\d+_\d+_\d+ = [^;]*ag\[', src)) == steps * nt * kp
        assert len(re.findall(r'const rt_d2 o\d+_\d+_\d+ = [^;]*g\d+\[', src)) == len(obs) * kp * tiles
        # odd number of k-steps: the last pair of every operand is an 8-byte load
        assert src.count('), 0.0};') == ((steps * nt + len(obs) * tiles) if ks % 2 else 0)
    # 32 < n <= 64: split-M family, NT waves share T tiles; wave m owns KS MFMAs per
    # step and tile and loads only its own slice of P_e (KP pairs per step)
    for n, tiles in ((33, 1), (61, 2)):
        monkeypatch.setenv('RAOTEH_JIT_TILES', str(tiles))
        _lib.check(_lib.lib().rt_jit_source(
            ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
            n, len(obs), obs.ctypes.data_as(p64), 1, buf, len(buf)))
        src = buf.value.decode()
        ks = (n + 3) // 4
        kp = (ks + 1) // 2
        steps = ta.nnodes - 1
        assert src.count('__builtin_amdgcn_mfma_f64_16x16x4f64') == steps * ks * tiles
        assert len(re.findall(r'const rt_d2 A\d+_\d+ = [^;]*ag\[', src)) == steps * kp
        assert src.count('__syncthreads()') == steps + 1      # one per step + the root's
    rc = _lib.lib().rt_jit_source(
        ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
        129, len(obs), obs.ctypes.data_as(p64), 2, buf, len(buf))
    assert rc < 0
    # 64 < n <= 128: NT = 5..8 waves per workgroup, one workgroup per CU
    rc = _lib.lib().rt_jit_source(
        ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
        122, len(obs), obs.ctypes.data_as(p64), 2, buf, len(buf))
    assert rc == 0
    src = buf.value.decode()
    assert '__launch_bounds__(512)' in src and 'amdgpu_waves_per_eu(2, 2)' in src


def test_frechet_block_assembly_is_the_adjoint_of_expm_frechet():
    """The identity the device's expected-history-statistics path rests on
    (csrc/expect.hip): with M = the upper right block of expm([[t Q^T, W], [0, t Q^T]]),
    <W, L(tQ, E_cd)> = M[c, d] for every direction E_cd -- the n + nnz(Q) expm_frechet
    calls of _mjp_dense.py:483-533 in one exponential, W of any size (it is scaled before
    it enters the block).  scipy stands in for the device here; the host plumbing
    (_edge_rates, _history_statistics_from_weights) is the product's."""
    import scipy.linalg
    from raoteh_amd import _mjp_dense

    class ScipyDevice(object):
        @staticmethod
        def frechet_statistics(Qs, q_index, t, W):
            n = Qs.shape[1]
            dwell, trans = np.zeros(n), np.zeros((n, n))
            for e in range(len(t)):
                Q = Qs[q_index[e]]
                sc = np.abs(W[e]).max() or 1.0
                B = np.zeros((2 * n, 2 * n))
                B[:n, :n] = B[n:, n:] = t[e] * Q.T
                B[:n, n:] = W[e] / sc
                M = scipy.linalg.expm(B)[:n, n:] * sc
                dwell += t[e] * np.diag(M)
                trans += np.where(Q != 0, t[e] * Q * M, 0.0)
            return dwell, trans

    rng = np.random.RandomState(5)
    for n in (2, 3, 5, 8):
        T = nx.Graph()
        mats = []
        for e in range(3):
            R = rng.exponential(size=(n, n))
            R[rng.uniform(size=(n, n)) < 0.3] = 0.0
            np.fill_diagonal(R, 0.0)
            mats.append(R - np.diag(R.sum(axis=1)))
        ts = rng.uniform(0.05, 2.0, size=4)
        T.add_edge(0, 1, weight=ts[0], Q=mats[0])
        T.add_edge(0, 2, weight=ts[1])                  # Q_default
        T.add_edge(2, 3, weight=ts[2], Q=mats[2])
        T.add_edge(2, 4, weight=ts[3], Q=mats[0])       # the same matrix object twice
        edges, distinct, q_index, tt = _mjp_dense._edge_rates(T, 0, mats[1])
        assert len(distinct) == 3 and len(edges) == 4
        for e, (na, nb) in enumerate(edges):
            assert distinct[q_index[e]] is not None
            np.testing.assert_array_equal(distinct[q_index[e]], T[na][nb].get('Q', mats[1]))
            assert tt[e] == T[na][nb]['weight']
        Ws = rng.exponential(size=(4, n, n)) * 10.0 ** rng.randint(-6, 7, size=(4, 1, 1))
        Ws[1] = 0.0                                      # an edge nothing was seen on
        dwell, trans = _mjp_dense._history_statistics_from_weights(
            ScipyDevice(), n, distinct, q_index, tt, Ws)
        # the reference's way: one expm_frechet per direction, contracted with W
        want_d, want_t = np.zeros(n), np.zeros((n, n))
        for e in range(4):
            Q = distinct[q_index[e]]
            for c in range(n):
                for d in range(n):
                    if c != d and Q[c, d] == 0:
                        continue
                    C = np.zeros((n, n))
                    C[c, d] = 1.0
                    L = scipy.linalg.expm_frechet(tt[e] * Q, C, compute_expm=False)
                    v = np.sum(Ws[e] * L)
                    if c == d:
                        want_d[c] += tt[e] * v
                    if Q[c, d] != 0:
                        want_t[c, d] += tt[e] * Q[c, d] * v
        np.testing.assert_allclose(dwell, want_d, rtol=1e-10)
        np.testing.assert_allclose(trans, want_t, rtol=1e-10, atol=1e-300)
    with pytest.raises(ValueError):
        _mjp_dense._history_statistics_from_weights(
            ScipyDevice(), 65, [np.zeros((65, 65))], np.zeros(1, dtype=np.int64), np.ones(1),
            np.zeros((1, 65, 65)))
    d0, t0 = _mjp_dense._history_statistics_from_weights(
        ScipyDevice(), 3, [], np.zeros(0, dtype=np.int64), np.zeros(0), np.zeros((0, 3, 3)))
    assert not d0.any() and not t0.any()


def test_observed_states_are_range_checked_not_wrapped():
    # a plain uint8 cast would turn state 256 into state 0 silently
    from raoteh_amd.device import _as_uint8_states
    a = _as_uint8_states(np.array([[0, 3, 255], [-1, 2, 1]]), 4)
    assert a.dtype == np.uint8 and a.tolist() == [[0, 3, 255], [255, 2, 1]]
    assert _as_uint8_states(np.array([[60, 255]], dtype=np.uint8), 61).tolist() == [[60, 255]]
    for bad in (np.array([[256]]), np.array([[4]]), np.array([[-2]]),
                np.array([[61]], dtype=np.uint8)):
        with pytest.raises(ValueError):
            _as_uint8_states(bad, 4 if bad.dtype != np.uint8 else 61)
    with pytest.raises(ValueError):
        _as_uint8_states(np.array([[0.0]]), 4)
    with pytest.raises(ValueError):
        _as_uint8_states(np.array([[0]]), 300)


def test_forest_container_and_uniformization():
    # host side of the Rao-Teh sweep core: the concatenated CSR layout rt_forest_* take
    # and the uniformized matrix (reference _sample_mjp_dense.py:72-114)
    from raoteh_amd._forest import Forest
    from raoteh_amd._sample_mjp_dense import get_uniformized_transition_matrix
    T0 = nx.Graph([(5, 7), (5, 9), (9, 2)])
    T1 = nx.Graph()
    T1.add_node(4)
    T2 = nx.path_graph(3)
    f = Forest([(T0, 5), (T1, 4), (T2, 1)])
    assert f.ntrees == 3 and f.total == 8
    assert f.node_offset.tolist() == [0, 4, 5, 8]
    assert f.preorder[0][0] == 5 and f.preorder[1] == [4] and f.preorder[2][0] == 1
    # tree k's indptr block starts at off[k] + k, its indices block at off[k] - k
    assert f.indptr.shape == (8 + 3,) and f.indices.shape == (8 - 3,)
    for k in range(3):
        lo, nn = int(f.node_offset[k]), len(f.preorder[k])
        ptr = f.indptr[lo + k:lo + k + nn + 1]
        assert ptr[0] == 0 and ptr[-1] == nn - 1
        idx = f.indices[lo - k:lo - k + nn - 1]
        assert sorted(idx.tolist()) == list(range(1, nn))
    m = f.allowed_masks([{5: {0, 2}, 2: {1}}, None, {0: {3}}], 4)
    assert m.dtype == np.uint64 and m[0] == 0b0101 and m[4] == 0b1111
    assert m[int(f.node_offset[0]) + f.preorder[0].index(2)] == 0b0010
    assert m[int(f.node_offset[2]) + f.preorder[2].index(0)] == 0b1000
    with pytest.raises(ValueError):
        f.allowed_masks([{5: {4}}, None, None], 4)
    with pytest.raises(ValueError):
        Forest([])
    Q = np.array([[-1., 1, 0], [2, -5, 3], [0, 0.5, -0.5]])
    P = get_uniformized_transition_matrix(Q)
    np.testing.assert_allclose(P, np.eye(3) + Q / 10.0)
    np.testing.assert_allclose(get_uniformized_transition_matrix(Q, uniformization_factor=3),
                               np.eye(3) + Q / 15.0)
    np.testing.assert_allclose(get_uniformized_transition_matrix(Q, omega=7.0),
                               np.eye(3) + Q / 7.0)
    with pytest.raises(ValueError):
        get_uniformized_transition_matrix(Q, uniformization_factor=2, omega=7.0)
    with pytest.raises(ValueError):
        get_uniformized_transition_matrix(np.zeros((2, 3)))


def _chunks_by_union_find(parent, edge_rows):
    """Independent statement of _graph_transform.get_chunk_tree_type_b (:298-375) for one
    chain: pieces are glued at non-event nodes; returns (piece -> chunk label, base node ->
    chunk label, set of chunk-tree edges as frozensets of labels)."""
    # elements: ('n', v) base nodes, ('p', v, k) pieces of edge v
    uf = {}

    def find(x):
        uf.setdefault(x, x)
        while uf[x] != x:
            uf[x] = uf[uf[x]]
            x = uf[x]
        return x

    def union(a, b):
        uf[find(a)] = find(b)

    for v, npieces in edge_rows.items():
        union(('n', int(parent[v])), ('p', v, 0))             # the upper node is not an event
        union(('n', v), ('p', v, npieces - 1))
        for k in range(npieces):
            find(('p', v, k))
    events = set()
    for v, npieces in edge_rows.items():
        for k in range(1, npieces):
            events.add(frozenset((find(('p', v, k - 1)), find(('p', v, k)))))
    return find, events


def test_chunk_forest_matches_a_union_find_statement():
    """The vectorised chunk trees of a batch of histories (raoteh_amd/_sampler.py) against
    a per-chain union-find statement of the reference's construction: same partition of
    pieces and base nodes into chunks, same chunk adjacency, parents before children."""
    from raoteh_amd import _sampler
    rng = np.random.RandomState(5)
    for N, C in ((2, 3), (7, 5), (15, 4), (31, 6)):
        parent = np.full(N, -1, dtype=np.int64)
        for v in range(1, N):
            parent[v] = rng.randint(max(0, v - 4), v)
        counts = rng.randint(1, 5, size=(C, N))
        counts[:, 0] = 0
        counts[rng.uniform(size=counts.shape) < 0.4] = 1      # many edges without events
        counts[:, 0] = 0
        chain = np.concatenate([np.full(counts[c, 1:].sum(), c) for c in range(C)]).astype(np.int64)
        edge = np.concatenate([np.repeat(np.arange(1, N), counts[c, 1:]) for c in range(C)]).astype(np.int64)
        offset, cparent, piece, node = _sampler.chunk_forest(parent, C, chain, edge)
        assert offset[0] == 0 and offset[-1] == cparent.shape[0]
        row = 0
        for c in range(C):
            lo, hi = int(offset[c]), int(offset[c + 1])
            local_parent = cparent[lo:hi]
            assert local_parent[0] == -1
            assert all(0 <= local_parent[i] < i for i in range(1, hi - lo))
            find, events = _chunks_by_union_find(parent, dict((v, int(counts[c, v])) for v in range(1, N)))
            label_of = {}
            for v in range(1, N):
                for k in range(int(counts[c, v])):
                    label_of.setdefault(find(('p', v, k)), set()).add(int(piece[row]) - lo)
                    row += 1
            for v in range(N):
                label_of.setdefault(find(('n', v)), set()).add(int(node[c, v]))
            # one chunk id per union-find class, and distinct classes get distinct ids
            ids = [next(iter(s)) for s in label_of.values()]
            assert all(len(s) == 1 for s in label_of.values())
            assert sorted(ids) == list(range(hi - lo))
            # the chunk tree's edges are the events
            got = set(frozenset((i, int(local_parent[i]))) for i in range(1, hi - lo))
            want = set(frozenset(next(iter(label_of[x])) for x in e) for e in events)
            assert got == want
        assert row == chain.shape[0]


def test_poisson_split_and_merge_keep_lengths_and_order():
    from raoteh_amd import _sampler
    rng = np.random.Generator(np.random.PCG64(3))
    seg_len = np.array([0.5, 2.0, 1e-3, 4.0, 0.25])
    rate = np.array([3.0, 0.0, 10.0, 2.5, 8.0])
    rep, sub = _sampler.poisson_split(rng, seg_len, rate)
    assert rep.shape == (5,) and rep[1] == 1 and sub.shape[0] == rep.sum()
    owner = np.repeat(np.arange(5), rep)
    np.testing.assert_allclose(np.bincount(owner, weights=sub), seg_len, rtol=1e-13)
    assert (sub > 0).all()
    assert sub[owner == 1][0] == 2.0                      # an untouched segment is not rounded
    # the number of events is Poisson(rate * length)
    big = np.full(20000, 1.5)
    rep, _ = _sampler.poisson_split(rng, big, np.full(20000, 2.0))
    assert abs((rep - 1).mean() - 3.0) < 0.05 and abs((rep - 1).var() - 3.0) < 0.15
    # merging: equal neighbours of one (chain, edge) fuse, nothing else does
    chain = np.array([0, 0, 0, 0, 0, 1, 1])
    edge = np.array([1, 1, 1, 2, 2, 1, 1])
    length = np.array([.1, .2, .3, .4, .5, .6, .7])
    state = np.array([2, 2, 1, 1, 1, 0, 0])
    c, e, l, s = _sampler.merge_segments(chain, edge, length, state)
    assert c.tolist() == [0, 0, 0, 1] and e.tolist() == [1, 1, 2, 1]
    np.testing.assert_allclose(l, [.3, .3, .9, 1.3])
    assert s.tolist() == [2, 1, 1, 0]


def test_spectral_host_mirror_decomposition():
    """raoteh_amd/_spectral.py (host side of examples/p53/qtop.py:104-150) against the
    reference's own factors in tests/golden/spectral.json; no device call."""
    from conftest import load_golden
    from oracle import oracle_numpy as orc
    from raoteh_amd import _spectral
    fx = load_golden('spectral')
    for c in fx['cases']:
        if 'S' not in c:
            continue
        S, D = np.array(c['S']), np.array(c['D'])
        A, lam, B = _spectral.decompose_spectral_v2(S, D)
        np.testing.assert_allclose(lam, np.array(c['lam']), rtol=1e-10, atol=1e-12)
        assert not A[D == 0].any()
        for k, t in enumerate(c['t']):
            np.testing.assert_allclose(orc.spectral_getp_v2(D, A, lam, B, t),
                                       np.array(c['P_spectral'][k]), rtol=1e-10, atol=1e-13)
    # a reversible matrix given with its distribution; and a non-reversible one
    mg = load_golden('p53_mg94')
    Q = np.array(mg['Q_offdiagonal'])
    Q -= np.diag(Q.sum(axis=1))
    A, lam, B, D = _spectral.decompose_rate_matrix(Q, mg['distn'])
    np.testing.assert_allclose((A * lam[None, :]) @ B, Q, rtol=0, atol=1e-12)
    Qbad = np.array([[-1.0, 1.0, 0.0], [0.0, -1.0, 1.0], [1.0, 0.0, -1.0]])
    with pytest.raises(ValueError):
        _spectral.decompose_rate_matrix(Qbad, [1 / 3.0] * 3)
    with pytest.raises(ValueError):
        _spectral.decompose_spectral(np.eye(3), [0.5, -0.1, 0.6])


def test_root_halves_source_covers_every_step_once(monkeypatch):
    """Split-M family, root halves (csrc/jit.hip split_at_root): the two root programs
    together hold every product of the tree exactly once, each reads its own leaves only,
    each ends by storing its share of the root's accumulator, and the module's second
    kernel multiplies the shares in child order.  Random trees (roots with 1..4 children,
    observed internal nodes), no device."""
    import ctypes
    import re
    from raoteh_amd import _lib, synth
    from raoteh_amd._tree import TreeArrays
    p64 = ctypes.POINTER(ctypes.c_int64)
    buf = ctypes.create_string_buffer(1 << 24)
    monkeypatch.setenv('RAOTEH_JIT_HALVES', '1')
    seen_cut = 0
    for fold, (seed, nnodes, n, tiles) in itertools.product(
            (False, True), ((1, 9, 33, 1), (2, 14, 61, 1), (3, 23, 64, 5), (4, 31, 48, 3),
                            (5, 12, 61, 2), (6, 5, 40, 1))):
        # RAOTEH_JIT_FOLD=1: the pair's second workgroup finishes the sites in the pruning
        # kernel itself; the default: the separate combine launch
        monkeypatch.setenv('RAOTEH_JIT_FOLD', '1' if fold else '0')
        T, root, leaves = synth.random_tree(nnodes, seed=seed, max_children=4)
        ta = TreeArrays(T, root)
        inner = [v for v in T if v not in leaves]
        obs_nodes = list(leaves) + inner[::2]
        obs = np.array(sorted(ta.node_to_index[v] for v in obs_nodes), dtype=np.int64)
        monkeypatch.setenv('RAOTEH_JIT_TILES', str(tiles))
        rc = _lib.lib().rt_jit_source(
            ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64),
            n, len(obs), obs.ctypes.data_as(p64), 2, buf, len(buf))
        nchildren = T.degree(root)
        if nchildren < 2:
            # nothing to cut: the generator returns no source and the caller keeps the
            # whole-tree form
            assert rc < 0 or buf.value == b''
            continue
        _lib.check(rc)
        seen_cut += 1
        src = buf.value.decode()
        ks = (n + 3) // 4
        kp = (ks + 1) // 2
        steps = ta.nnodes - 1
        main, combine = src.split('rt_jit_combine(')
        assert main.count('__builtin_amdgcn_mfma_f64_16x16x4f64') == steps * ks * tiles
        assert len(re.findall(r'const rt_d2 A\d+_\d+ = [^;]*ag\[', main)) == steps * kp
        progA, progB = main.split('    } else {\n', 1)
        assert 'if (half == 0) {' in progA
        # every observed non-root node is fetched by exactly one of the two programs
        nonroot_obs = len(obs) - (1 if ta.node_to_index[root] in obs else 0)
        loads = [set(re.findall(r'const rt_d2 o(\d+)_0_0 = ', p)) for p in (progA, progB)]
        assert not (loads[0] & loads[1])
        assert len(loads[0] | loads[1]) == nonroot_obs
        for k, prog in enumerate((progA, progB)):
            assert len(re.findall(r'double \*hb = halfbuf \+ \(\(size_t\)tile\d+ \* 2 \+ %d\)' % k,
                                  prog)) == tiles
        # the second kernel: product of the two shares (times the root's own observation)
        assert len(re.findall(r'const double xr_0_\d = ha\[\d+\] \* ha\[\d+\]', combine)) == 4
        assert ('ob_0' in combine) == (ta.node_to_index[root] in obs)
        assert 'loglik[site]' in combine
        assert ('loglik[site]' in main) == fold
        assert ('int *__restrict__ counters' in main) == fold
        if fold:
            tail = progB.split("the pair's second workgroup", 1)[1]
            assert tail.count('__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")') == 1
            assert progB.count('__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")') == 1
            assert len(re.findall(r'const double xr_\d+_\d = __hip_atomic_load\(&hx', tail)) == 4 * tiles
            assert main.count('__hip_atomic_store(&hb[') == 2 * 4 * tiles
    assert seen_cut >= 8


def test_pyfelscore_compat_exports_every_name_the_reference_calls():
    """Container only (the reference tree is not on the GPU box): every ``pyfelscore.<name>``
    the reference's sampler modules and its p53 example call is exported by the shim."""
    import ast
    import glob
    ref = '/root/reference'
    if not os.path.isdir(ref):
        pytest.skip('the reference tree is only present in the build container')
    called = set()
    for pattern in ('raoteh/sampler/*.py', 'examples/p53/*.py'):
        for path in glob.glob(os.path.join(ref, pattern)):
            with open(path, errors='replace') as f:
                called.update(re.findall(r'pyfelscore\.(\w+)', f.read()))
    assert {'mcy_get_node_to_pset', 'get_node_to_set', 'tmjp_get_inhomogeneous_mjp',
            'get_lb_transition_matrix', 'mcy_esd_get_node_to_pmap'} <= called
    tree = ast.parse(open(os.path.join(ROOT, 'raoteh_amd', 'pyfelscore_compat.py')).read())
    exported, defined = set(), set()
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], 'id', '') == '__all__':
            exported = set(ast.literal_eval(node.value))
        if isinstance(node, ast.FunctionDef):
            defined.add(node.name)
    assert called <= exported, sorted(called - exported)
    assert exported <= defined, sorted(exported - defined)


def test_tmjp_get_inhomogeneous_mjp_matches_the_reference_twin():
    """pyfelscore.tmjp_get_inhomogeneous_mjp (_tmjp_dense.py:1039-1054): host-side entry
    point of the library, against what the reference's sparse twin (_tmjp.py:803-903)
    returns for the same trajectories (tests/golden/tmjp_inhomogeneous.json)."""
    from raoteh_amd import pyfelscore_compat as pyf
    fx = load_golden('tmjp_inhomogeneous')
    assert len(fx['cases']) >= 10
    for c in fx['cases']:
        n = c['nprimary']
        Q = np.array(c['Q_primary_offdiagonal'])
        Q = Q - np.diag(Q.sum(axis=1))
        nnodes = len(c['edge_to_primary_state'])
        for cl in c['classes']:
            allowed = np.ones((nnodes, 2), dtype=np.int64)
            mats = np.full((nnodes, 3, 3), np.nan)
            pyf.tmjp_get_inhomogeneous_mjp(
                np.array(c['tree_csr_indices'], dtype=np.int64),
                np.array(c['tree_csr_indptr'], dtype=np.int64),
                np.array(c['edge_to_primary_state'], dtype=np.int64),
                np.array(c['primary_to_part'], dtype=np.int64), Q, c['rate_on'], c['rate_off'],
                cl['tolerance_class'], allowed, mats)
            np.testing.assert_array_equal(allowed, np.array(cl['node_to_allowed_tolerances']))
            np.testing.assert_allclose(mats, np.array(cl['tol_rate_matrices']), rtol=1e-14,
                                       atol=0)
    with pytest.raises(ValueError):
        pyf.tmjp_get_inhomogeneous_mjp(
            np.array([1], dtype=np.int64), np.array([0, 1, 1], dtype=np.int64),
            np.array([0, 7], dtype=np.int64), np.array([0, 0], dtype=np.int64), np.zeros((2, 2)),
            1.0, 1.0, 0, np.ones((2, 2), dtype=np.int64), np.zeros((2, 3, 3)))


def _hiprtc_compile(src, vgpr_form):
    """hiprtc with the options rt_jit_get uses (csrc/jit.hip); returns the code object."""
    import ctypes
    rtc = ctypes.CDLL('libhiprtc.so')
    prog = ctypes.c_void_p()
    assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src, b'rt_jit_prune.hip', 0, None,
                                   None) == 0
    opts = [b'--offload-arch=gfx950', b'-O3', b'-std=c++17']
    if vgpr_form:
        opts += [b'-mllvm', b'-amdgpu-mfma-vgpr-form=1']
    arr = (ctypes.c_char_p * len(opts))(*opts)
    rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
    if rc != 0:
        sz = ctypes.c_size_t()
        rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(sz))
        log = ctypes.create_string_buffer(sz.value + 1)
        rtc.hiprtcGetProgramLog(prog, log)
        raise AssertionError('hiprtc: ' + log.value.decode()[:3000])
    sz = ctypes.c_size_t()
    rtc.hiprtcGetCodeSize(prog, ctypes.byref(sz))
    code = ctypes.create_string_buffer(sz.value)
    rtc.hiprtcGetCode(prog, code)
    rtc.hiprtcDestroyProgram(ctypes.byref(prog))
    return code.raw


def test_every_generator_emits_source_that_compiles_for_gfx950(tmp_path, monkeypatch):
    """The text of every tree-specialised family -- lane, one-wave MFMA (16x16x4 and 4x4x4
    blocks), split-M serial / pipelined / root halves / folded combine, NT = 8 waves above 64
    states, and the leaf-state / leaf-set forms of each -- goes through hiprtc for gfx950 with
    the product's options, here, without a device: no compiler error, no scratch (the
    rejection rule of rt_jit_get), the kernel symbols rt_sites_create looks up."""
    import ctypes
    import re
    import subprocess
    from raoteh_amd import _lib, synth
    from raoteh_amd._tree import TreeArrays
    p64 = ctypes.POINTER(ctypes.c_int64)
    buf = ctypes.create_string_buffer(1 << 24)
    readelf = '/opt/rocm/lib/llvm/bin/llvm-readelf'
    knobs = ('RAOTEH_JIT_TILES', 'RAOTEH_JIT_QUAD', 'RAOTEH_JIT_HALVES', 'RAOTEH_JIT_FOLD',
             'RAOTEH_JIT_SOURCE_SPARSE', 'RAOTEH_JIT_SOURCE_STATES')
    cases = [
        # (states, leaves of the balanced tree, prefetch, environment)
        (3, 8, 3, {}),
        (4, 8, 3, {'RAOTEH_JIT_SOURCE_STATES': '1'}),
        (20, 8, 2, {'RAOTEH_JIT_TILES': '2'}),
        (20, 8, 2, {'RAOTEH_JIT_TILES': '2', 'RAOTEH_JIT_SOURCE_SPARSE': '1'}),
        (13, 8, 2, {'RAOTEH_JIT_TILES': '1', 'RAOTEH_JIT_SOURCE_SPARSE': '2'}),
        (32, 8, 2, {'RAOTEH_JIT_TILES': '2', 'RAOTEH_JIT_QUAD': '0'}),
        (61, 8, 2, {'RAOTEH_JIT_TILES': '2'}),
        (61, 8, 2, {'RAOTEH_JIT_TILES': '3', 'RAOTEH_JIT_HALVES': '1'}),
        (48, 8, 2, {'RAOTEH_JIT_TILES': '2', 'RAOTEH_JIT_HALVES': '1', 'RAOTEH_JIT_FOLD': '1'}),
        (61, 8, 2, {'RAOTEH_JIT_TILES': '2', 'RAOTEH_JIT_SOURCE_SPARSE': '1'}),
        (61, 8, 2, {'RAOTEH_JIT_TILES': '2', 'RAOTEH_JIT_HALVES': '1',
                    'RAOTEH_JIT_SOURCE_SPARSE': 'pipe'}),
        (33, 8, 2, {'RAOTEH_JIT_TILES': '1', 'RAOTEH_JIT_SOURCE_SPARSE': 'pipe2'}),
        (122, 4, 2, {'RAOTEH_JIT_TILES': '1', 'RAOTEH_JIT_HALVES': '1'}),
        (122, 4, 2, {'RAOTEH_JIT_TILES': '1', 'RAOTEH_JIT_HALVES': '1',
                     'RAOTEH_JIT_SOURCE_SPARSE': 'pipe2'}),
        (97, 4, 2, {'RAOTEH_JIT_TILES': '1', 'RAOTEH_JIT_SOURCE_SPARSE': '1'}),
    ]
    for k, (n, nleaves, prefetch, env) in enumerate(cases):
        for name in knobs:
            monkeypatch.delenv(name, raising=False)
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        T, root, leaves = synth.balanced_tree(nleaves, seed=k)
        ta = TreeArrays(T, root)
        obs = np.array(sorted(ta.node_to_index[v] for v in leaves), dtype=np.int64)
        _lib.check(_lib.lib().rt_jit_source(
            ta.nnodes, ta.indices.ctypes.data_as(p64), ta.indptr.ctypes.data_as(p64), n,
            len(obs), obs.ctypes.data_as(p64), prefetch, buf, len(buf)))
        src = buf.value
        assert src, (n, env)
        # the leaf-state forms read the leaves' state words and have fewer matrix steps
        assert (b'leafw' in src) == ('RAOTEH_JIT_SOURCE_SPARSE' in env), (n, env)
        code = _hiprtc_compile(src, vgpr_form=n > 4)
        path = tmp_path / ('k%d.co' % k)
        path.write_bytes(code)
        notes = subprocess.run([readelf, '--notes', str(path)], stdout=subprocess.PIPE,
                               check=True).stdout.decode()
        names = re.findall(r'\.name:\s+(\S+)', notes)
        assert 'rt_jit_prune' in names, (n, env, names)
        assert ('rt_jit_combine' in names) == \
            (env.get('RAOTEH_JIT_HALVES') == '1' and n > 32 and
             env.get('RAOTEH_JIT_SOURCE_SPARSE') != '1'), (n, env, names)
        scratch = [int(v) for v in re.findall(r'\.private_segment_fixed_size:\s+(\d+)', notes)]
        spills = [int(v) for v in re.findall(r'\.vgpr_spill_count:\s+(\d+)', notes)]
        assert scratch and max(scratch) == 0 and max(spills + [0]) == 0, (n, env, scratch, spills)


def test_library_kernels_have_no_scratch_beyond_the_known_few(tmp_path):
    """Register pressure is what the hot kernels are tuned against: a kernel that starts to
    spill to scratch is a performance regression no parity test sees.  The code objects inside
    libraoteh_hip.so (llvm-objdump --offloading, no device needed): scratch only in the listed
    kernels, within the listed bytes -- the order-above-64 exponentials at 7 and 8 row tiles,
    the legacy global-scratch exponential they replaced, the weighted site sums."""
    import re
    import shutil
    import subprocess
    from raoteh_amd import _lib
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    readelf = '/opt/rocm/lib/llvm/bin/llvm-readelf'
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip('llvm-objdump / llvm-readelf not found')
    if 'debug' in os.path.basename(_lib.LIB_PATH):
        pytest.skip('the sanitizer build (make debug-test) is not the tuned one')
    so = tmp_path / 'lib.so'
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, '--offloading', str(so)], cwd=tmp_path, check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    objs = sorted(tmp_path.glob('lib.so.*gfx950'))
    assert len(objs) >= 8, objs                      # one per translation unit with kernels
    allowed = {                                      # mangled-name fragment -> bytes
        'expm_taylor_kernelILi5ELb1': 1100, 'expm_taylor_kernelILi6ELb1': 1100,
        'expm_taylor_kernelILi7ELb1': 1100, 'expm_taylor_kernelILi8ELb1': 1100,
        'expm_taylor_wide_kernelILi8ELb1': 160, 'expm_taylor_wide_kernelILi7ELb0': 64,
        'expm_taylor_wide_kernelILi8ELb0': 80, 'expect_wsum_kernelILi4ELb1': 16,
    }
    seen = 0
    hot = 0
    for obj in objs:
        notes = subprocess.run([readelf, '--notes', str(obj)], stdout=subprocess.PIPE,
                               check=True).stdout.decode()
        for block in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
            name = re.search(r'\.name:\s+(\S+)', block).group(1)
            scratch = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', block).group(1))
            seen += 1
            limit = max([v for k, v in allowed.items() if k in name] + [0])
            assert scratch <= limit, (name, scratch, limit)
            if re.search(r'prune_mfma_kernel|expect_down_lds_kernel|expm_taylor_kernelILi\dELb0|'
                         r'spectral_kernel|expect_wsum_kernelILi\dELb0', name):
                hot += 1
                assert scratch == 0, (name, scratch)
    assert seen > 300 and hot >= 40, (seen, hot)
