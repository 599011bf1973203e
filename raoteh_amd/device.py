"""
Device-resident objects of the batched hot path (thin wrappers over the C ABI):

    ctx   = get_context()                      one per process / GPU
    model = TreeModel(T, root, nstates)        tree + schedule on the device
    model.set_rates(Q_default=Q)               per-edge expm(Q*t) on the device
    batch = model.upload_sites(obs_nodes, data, kind='dense')
    loglik, status = model.log_likelihoods(batch)
    total, nzero = model.total_log_likelihood(batch)

Reference path being replaced: raoteh/sampler/_mjp_dense.py:362-407 called once
per site (examples/p53/p53.py:88-100).
"""
from __future__ import annotations

import atexit
import ctypes
import weakref
from ctypes import byref, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np

from . import _lib
from ._tree import TreeArrays

__all__ = ['Context', 'get_context', 'TreeModel', 'SiteBatch', 'device_count']

# At interpreter shutdown objects are finalised in arbitrary order (a model
# after its context, say); the process is going away, so skip the native
# destructors then instead of handing the library dangling handles.
_shutting_down = []
atexit.register(_shutting_down.append, True)

_KINDS = {'dense': _lib.RT_OBS_DENSE, 'state': _lib.RT_OBS_STATE,
          'mask': _lib.RT_OBS_MASK}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def _as_uint8_states(data, nstates):
    """Observed states -> uint8 (255 = unobserved).  A plain cast would wrap a state
    >= 256 (or a negative one) onto another state silently; anything that is neither a
    state in [0, nstates) nor the 255 / -1 'unobserved' marker is an error."""
    a = np.asarray(data)
    if a.dtype == np.uint8:
        bad = (a >= nstates) & (a != 255)
    else:
        if not np.issubdtype(a.dtype, np.integer):
            raise ValueError('observed states must be integers')
        bad = ((a < 0) | (a >= nstates)) & (a != 255) & (a != -1)
        a = np.where(a == -1, 255, a)
    if nstates > 255 or bad.any():
        raise ValueError('observed state outside [0, %d) (255 or -1 = unobserved)'
                         % nstates)
    return np.ascontiguousarray(a, dtype=np.uint8)


def device_count():
    n = c_int(0)
    _lib.check(_lib.lib().rt_device_count(byref(n)))
    return n.value


class Context(object):
    """One HIP stream on one GPU.  Fails loudly without the library or a GPU."""

    def __init__(self, device=0):
        self._h = c_void_p()
        # models and chain batches of this context: the C objects refer to it, and
        # rt_ctx_destroy refuses while one of them lives, so close() closes them first
        # (objects that die in one garbage-collection cycle are finalised in any order)
        self._children = weakref.WeakSet()
        _lib.check(_lib.lib().rt_ctx_create(int(device), byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h and not _shutting_down:
            for child in list(self._children):
                child.close()
            _lib.check(_lib.lib().rt_ctx_destroy(self._h))
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(_lib.lib().rt_ctx_sync(self._h))

    def set_option(self, key, value):
        """Per-context option (rt_ctx_set_option): 'jit' (-1 automatic / 0 / 1),
        'force_generic', 'jit_block_sites', 'jit_async', 'leaf_state_kernels', 'rescale' (power-of-two rescaling
        of the messages of batches uploaded from now on: trees whose likelihood underflows
        f64); value None = back to the process default."""
        _lib.check(_lib.lib().rt_ctx_set_option(
            self._h, key.encode(), -2 if value is None else int(value)))

    def set_timing(self, enabled):
        """False/0: off; True/1: every launch; N > 1: every N-th launch of each
        kernel (each HIP event pair costs a few microseconds of stream time)."""
        _lib.check(_lib.lib().rt_ctx_set_timing(self._h, int(enabled)))

    def reset_timing(self):
        _lib.check(_lib.lib().rt_ctx_reset_timing(self._h))

    def kernel_time(self, kernel):
        """(total_ms, launches, kernel name) accumulated since the last reset."""
        ms = c_double(0.0)
        cnt = c_int64(0)
        name = c_char_p()
        _lib.check(_lib.lib().rt_ctx_kernel_time(self._h, int(kernel), byref(ms),
                                                 byref(cnt), byref(name)))
        return ms.value, cnt.value, (name.value or b'').decode()

    # ---- reference-shaped, host-pointer entry points ----------------------

    def expm(self, Q, t, q_index=None, return_info=False):
        """P[b] = expm(Q[q_index[b]] * t[b]); Q is [n,n] or [nq,n,n]."""
        Q = _f64(Q)
        if Q.ndim == 2:
            Q = Q[None]
        if Q.ndim != 3 or Q.shape[1] != Q.shape[2]:
            raise ValueError('expected the array to be square')
        t = np.atleast_1d(_f64(t))
        n, nq, count = Q.shape[1], Q.shape[0], t.shape[0]
        P = np.empty((count, n, n), dtype=np.float64)
        info = np.zeros((count, 2), dtype=np.int32)
        qi = None if q_index is None else _i64(q_index)
        _lib.check(_lib.lib().rt_expm(
            self._h, n, count, _ptr(Q, c_double), nq,
            None if qi is None else _ptr(qi, c_int64), _ptr(t, c_double),
            _ptr(P, c_double), _ptr(info, c_int32)))
        return (P, info) if return_info else P

    def expm_spectral(self, A, lam, B, t, D=None):
        """P[b] = A diag(exp(lam t[b])) B, diagonal 1 where D == 0: the reference's
        getp_spectral_v2 (examples/p53/qtop.py:76-88) at all branch lengths in one launch."""
        A, lam, B = _f64(A), _f64(lam), _f64(B)
        n = lam.shape[0]
        if A.shape != (n, n) or B.shape != (n, n):
            raise ValueError('expected the array to be square')
        t = np.atleast_1d(_f64(t))
        D = None if D is None else _f64(D)
        if D is not None and D.shape != (n,):
            raise ValueError('D must have one entry per state')
        P = np.empty((t.shape[0], n, n), dtype=np.float64)
        _lib.check(_lib.lib().rt_expm_spectral(
            self._h, n, t.shape[0], _ptr(A, c_double), _ptr(lam, c_double), _ptr(B, c_double),
            None if D is None else _ptr(D, c_double), _ptr(t, c_double), _ptr(P, c_double)))
        return P

    def _pass_args(self, indices, indptr, esd, arr):
        indices, indptr, esd = _i64(indices), _i64(indptr), _f64(esd)
        nnodes, n = esd.shape[0], esd.shape[1]
        if arr.ndim == 2:
            nsites = 1
        elif arr.ndim == 3:
            nsites = arr.shape[0]
        else:
            raise ValueError('expected [nnodes,n] or [nsites,nnodes,n]')
        if arr.shape[-2:] != (nnodes, n):
            raise ValueError('array shape %s does not match (%d, %d)' % (
                arr.shape, nnodes, n))
        return indices, indptr, esd, nnodes, n, nsites

    def node_to_pset(self, indices, indptr, esd, state_mask):
        """In place on state_mask (int64, C-contiguous)."""
        self._mask_pass('rt_mcy_esd_get_node_to_pset', indices, indptr, esd,
                        state_mask)

    def node_to_set(self, indices, indptr, esd, state_mask):
        self._mask_pass('rt_esd_get_node_to_set', indices, indptr, esd,
                        state_mask)

    def _mask_pass(self, fn, indices, indptr, esd, state_mask):
        if (state_mask.dtype != np.int64 or
                not state_mask.flags['C_CONTIGUOUS']):
            raise ValueError('state_mask must be a C-contiguous int64 array')
        indices, indptr, esd, nnodes, n, nsites = self._pass_args(
            indices, indptr, esd, state_mask)
        _lib.check(getattr(_lib.lib(), fn)(
            self._h, nnodes, n, nsites, _ptr(indices, c_int64),
            _ptr(indptr, c_int64), _ptr(esd, c_double),
            _ptr(state_mask, c_int64)))

    def node_to_pmap(self, indices, indptr, esd, state_mask, out,
                     obs_likelihood=None):
        if out.dtype != np.float64 or not out.flags['C_CONTIGUOUS']:
            raise ValueError('subtree_probability must be C-contiguous f64')
        state_mask = _i64(state_mask)
        indices, indptr, esd, nnodes, n, nsites = self._pass_args(
            indices, indptr, esd, state_mask)
        if out.shape != state_mask.shape:
            raise ValueError('shape mismatch')
        obs = None if obs_likelihood is None else _f64(obs_likelihood)
        if obs is not None and obs.shape != state_mask.shape:
            raise ValueError('shape mismatch')
        _lib.check(_lib.lib().rt_mcy_esd_get_node_to_pmap(
            self._h, nnodes, n, nsites, _ptr(indices, c_int64),
            _ptr(indptr, c_int64), _ptr(esd, c_double),
            _ptr(state_mask, c_int64),
            None if obs is None else _ptr(obs, c_double), _ptr(out, c_double)))

    def passes(self, indices, indptr, esd, state_mask, out, obs_likelihood=None):
        """pset + set + pmap in one call (rt_mcy_esd_passes): state_mask (int64,
        C-contiguous) is updated in place, ``out`` receives the pmaps."""
        if (state_mask.dtype != np.int64 or not state_mask.flags['C_CONTIGUOUS']):
            raise ValueError('state_mask must be a C-contiguous int64 array')
        if out.dtype != np.float64 or not out.flags['C_CONTIGUOUS']:
            raise ValueError('subtree_probability must be C-contiguous f64')
        indices, indptr, esd, nnodes, n, nsites = self._pass_args(
            indices, indptr, esd, state_mask)
        if out.shape != state_mask.shape:
            raise ValueError('shape mismatch')
        obs = None
        if obs_likelihood is not None:
            obs = _f64(obs_likelihood)
            if obs.shape != state_mask.shape:
                raise ValueError('obs_likelihood shape mismatch')
        _lib.check(_lib.lib().rt_mcy_esd_passes(
            self._h, nnodes, n, nsites, _ptr(indices, c_int64), _ptr(indptr, c_int64),
            _ptr(esd, c_double), _ptr(state_mask, c_int64),
            None if obs is None else _ptr(obs, c_double), _ptr(out, c_double)))

    def node_to_distn(self, indices, indptr, esd, root_distn, pmap):
        """Downward pass (mc0_esd_get_node_to_distn): returns (distn, status)."""
        pmap = _f64(pmap)
        indices, indptr, esd, nnodes, n, nsites = self._pass_args(
            indices, indptr, esd, pmap)
        out = np.empty(pmap.shape, dtype=np.float64)
        status = np.zeros(nsites, dtype=np.int32)
        rd = None if root_distn is None else _f64(root_distn)
        if rd is not None and rd.shape != (n,):
            raise ValueError('inconsistent root distribution')
        _lib.check(_lib.lib().rt_mc0_esd_get_node_to_distn(
            self._h, nnodes, n, nsites, _ptr(indices, c_int64), _ptr(indptr, c_int64),
            _ptr(esd, c_double), None if rd is None else _ptr(rd, c_double),
            _ptr(pmap, c_double), _ptr(out, c_double), _ptr(status, c_int32)))
        return out, status

    def joint_endpoint_distn(self, indices, indptr, esd, pmap, distn):
        """mc0_esd_get_joint_endpoint_distn: f64[..., nnodes, n, n] keyed by
        the child index."""
        pmap, distn = _f64(pmap), _f64(distn)
        indices, indptr, esd, nnodes, n, nsites = self._pass_args(
            indices, indptr, esd, pmap)
        if distn.shape != pmap.shape:
            raise ValueError('shape mismatch')
        out = np.empty(pmap.shape + (n,), dtype=np.float64)
        _lib.check(_lib.lib().rt_mc0_esd_get_joint_endpoint_distn(
            self._h, nnodes, n, nsites, _ptr(indices, c_int64), _ptr(indptr, c_int64),
            _ptr(esd, c_double), _ptr(pmap, c_double), _ptr(distn, c_double),
            _ptr(out, c_double)))
        return out

    def expectation_weights(self, indices, indptr, esd, root_distn, state_mask,
                            site_weights=None):
        """rt_mjp_esd_expectation_weights: upward passes + downward pass + per-edge
        site sums of J / P on the device.  Returns (W f64[nnodes, n, n] keyed by the
        child index, summed root posteriors f64[n], status int32[nsites])."""
        state_mask = _i64(state_mask)
        indices, indptr, esd, nnodes, n, nsites = self._pass_args(
            indices, indptr, esd, state_mask)
        rd = None if root_distn is None else _f64(root_distn)
        if rd is not None and rd.shape != (n,):
            raise ValueError('inconsistent root distribution')
        w = None if site_weights is None else _f64(site_weights)
        if w is not None and w.shape != (nsites,):
            raise ValueError('one weight per site expected')
        W = np.empty((nnodes, n, n), dtype=np.float64)
        status = np.zeros(nsites, dtype=np.int32)
        _lib.check(_lib.lib().rt_mjp_esd_expectation_weights(
            self._h, nnodes, n, nsites, _ptr(indices, c_int64), _ptr(indptr, c_int64),
            _ptr(esd, c_double), None if rd is None else _ptr(rd, c_double),
            _ptr(state_mask, c_int64), None if w is None else _ptr(w, c_double),
            _ptr(W, c_double), _ptr(status, c_int32)))
        root_post = W[0, :, 0].copy()
        W[0] = 0.0
        return W, root_post, status

    def expectation_weights_obs(self, indices, indptr, esd, root_distn, obs_nodes, data,
                                kind, site_weights=None):
        """rt_mjp_esd_expectation_weights_obs: as expectation_weights, from compact
        observations -- ``data`` [nsites, len(obs_nodes)] uint8 states (kind='state')
        or uint64 allowed-set masks (kind='mask'); obs_nodes are preorder indices."""
        indices, indptr, esd = _i64(indices), _i64(indptr), _f64(esd)
        nnodes, n = esd.shape[0], esd.shape[1]
        obs_nodes = _i64(obs_nodes)
        if kind == 'state':
            data = _as_uint8_states(data, n)
            code = _lib.RT_OBS_STATE
        elif kind == 'mask':
            data = np.ascontiguousarray(data, dtype=np.uint64)
            code = _lib.RT_OBS_MASK
        else:
            raise ValueError("kind must be 'state' or 'mask'")
        if data.ndim != 2 or data.shape[1] != obs_nodes.shape[0]:
            raise ValueError('data must be [nsites, len(obs_nodes)]')
        nsites = data.shape[0]
        rd = None if root_distn is None else _f64(root_distn)
        if rd is not None and rd.shape != (n,):
            raise ValueError('inconsistent root distribution')
        w = None if site_weights is None else _f64(site_weights)
        if w is not None and w.shape != (nsites,):
            raise ValueError('one weight per site expected')
        W = np.empty((nnodes, n, n), dtype=np.float64)
        status = np.zeros(nsites, dtype=np.int32)
        _lib.check(_lib.lib().rt_mjp_esd_expectation_weights_obs(
            self._h, nnodes, n, nsites, _ptr(indices, c_int64), _ptr(indptr, c_int64),
            _ptr(esd, c_double), None if rd is None else _ptr(rd, c_double),
            obs_nodes.shape[0], _ptr(obs_nodes, c_int64), code,
            data.ctypes.data_as(c_void_p), None if w is None else _ptr(w, c_double),
            _ptr(W, c_double), _ptr(status, c_int32)))
        root_post = W[0, :, 0].copy()
        W[0] = 0.0
        return W, root_post, status

    def frechet_statistics(self, Qs, q_index, t, W):
        """rt_mjp_frechet_statistics: per edge e the Frechet derivative
        M_e = L(t[e] Q[q_index[e]]^T, W[e]) on the device, contracted over the edges:
        returns (dwell f64[n], trans f64[n, n])."""
        Qs = _f64(Qs)
        if Qs.ndim == 2:
            Qs = Qs[None]
        W = _f64(W)
        t = np.atleast_1d(_f64(t))
        nq, n = Qs.shape[0], Qs.shape[1]
        nedges = t.shape[0]
        if Qs.shape[1:] != (n, n) or W.shape != (nedges, n, n):
            raise ValueError('expected Q [nq, n, n], W [nedges, n, n], t [nedges]')
        qi = _i64(q_index)
        if qi.shape != (nedges,):
            raise ValueError('one rate-matrix index per edge expected')
        dwell = np.zeros(n, dtype=np.float64)
        trans = np.zeros((n, n), dtype=np.float64)
        _lib.check(_lib.lib().rt_mjp_frechet_statistics(
            self._h, n, nedges, _ptr(Qs, c_double), nq, _ptr(qi, c_int64), _ptr(t, c_double),
            _ptr(W, c_double), _ptr(dwell, c_double), _ptr(trans, c_double)))
        return dwell, trans

    # ---- multi-GPU ---------------------------------------------------------

    @staticmethod
    def comm_available():
        """True iff librccl can be loaded (no GPU / network touched)."""
        return _lib.lib().rt_comm_available() == _lib.RT_OK

    @staticmethod
    def comm_unique_id():
        buf = (ctypes.c_ubyte * 128)()
        _lib.check(_lib.lib().rt_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, uid):
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(uid)
        _lib.check(_lib.lib().rt_comm_init(self._h, int(nranks), int(rank), buf))

    def comm_destroy(self):
        _lib.check(_lib.lib().rt_comm_destroy(self._h))


_contexts = {}


def get_context(device=0):
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx


class SiteBatch(object):
    def __init__(self, model, handle, nsites):
        self.model = model
        self._h = handle
        self.nsites = nsites
        # the C object refers to its model: when both die in one garbage-collection cycle
        # (a traceback kept them alive) the model may be finalised first, so the model closes
        # its batches before it goes
        model._batches.add(self)

    def clone(self):
        h = c_void_p()
        _lib.check(_lib.lib().rt_sites_clone(self._h, byref(h)))
        return SiteBatch(self.model, h, self.nsites)

    @property
    def device_bytes(self):
        return _lib.lib().rt_sites_device_bytes(self._h)

    @property
    def jit_compile_seconds(self):
        """hiprtc seconds spent for this batch's tree-specialised kernel (0: none
        compiled, or it came from the cache)."""
        return _lib.lib().rt_sites_jit_compile_seconds(self._h)

    @property
    def kernel_name(self):
        return (_lib.lib().rt_sites_kernel_name(self._h) or b'').decode()

    def set_weights(self, weights=None):
        """Per-site multiplicities for expected_history_statistics (site patterns);
        None = every site counts once."""
        if weights is None:
            _lib.check(_lib.lib().rt_sites_set_weights(self._h, None))
            return self
        w = _f64(weights)
        if w.shape != (self.nsites,):
            raise ValueError('one weight per site expected')
        _lib.check(_lib.lib().rt_sites_set_weights(self._h, _ptr(w, c_double)))
        return self

    def wait_for_kernel(self):
        """Block until the background compile of this batch's tree-specialised kernel (if
        one is pending: rt_set_option 'jit_async') has finished and the batch has switched
        to it; until then it runs the interpreter kernel, with the same results."""
        _lib.check(_lib.lib().rt_sites_jit_wait(self._h))
        return self

    def close(self):
        if self._h and not _shutting_down:
            _lib.lib().rt_sites_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TreeModel(object):
    """Tree + per-edge transition matrices resident on one GPU."""

    def __init__(self, T, root, nstates, ctx=None):
        self.ctx = ctx if ctx is not None else get_context()
        self.tree = T if isinstance(T, TreeArrays) else TreeArrays(T, root)
        self.nstates = int(nstates)
        self._h = c_void_p()
        self._batches = weakref.WeakSet()
        ta = self.tree
        _lib.check(_lib.lib().rt_model_create(
            self.ctx._h, ta.nnodes, self.nstates, _ptr(ta.indices, c_int64),
            _ptr(ta.indptr, c_int64), byref(self._h)))
        self.ctx._children.add(self)

    def close(self):
        for batch in list(getattr(self, '_batches', ())):
            batch.close()
        if self._h and not _shutting_down:
            _lib.check(_lib.lib().rt_model_destroy(self._h))
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def schedule_depth(self):
        return _lib.lib().rt_model_schedule_depth(self._h)

    def set_rates(self, Q_default=None, Q=None, node_q=None, t=None):
        """expm(Q*t) for every edge on the device.  Either pass nothing but
        Q_default (edge 'Q' attributes override it, _mjp_dense.py:355) or the
        explicit arrays Q [nq,n,n], node_q [nnodes], t [nnodes]."""
        if Q is None:
            Q, node_q = self.tree.rate_matrices(self.nstates, Q_default)
        else:
            Q = _f64(Q)
            if Q.ndim == 2:
                Q = Q[None]
            if Q.shape[1:] != (self.nstates, self.nstates):
                raise ValueError('expected the array to be square')
        if t is None:
            t = self.tree.branch_lengths()
        t = _f64(t)
        nq = _i64(node_q) if node_q is not None else None
        if t.shape != (self.tree.nnodes,):
            raise ValueError('t must have one entry per node')
        _lib.check(_lib.lib().rt_model_set_rates(
            self._h, _ptr(Q, c_double), Q.shape[0],
            None if nq is None else _ptr(nq, c_int64), _ptr(t, c_double)))

    def set_rates_spectral(self, A, lam, B, D=None, t=None):
        """One time-reversible rate matrix given by its spectral decomposition
        (raoteh_amd._spectral.decompose_spectral_v2, examples/p53/qtop.py:128-152): every
        edge's P = A diag(exp(lam t)) B on the device, now and at every later step()."""
        A, lam, B = _f64(A), _f64(lam), _f64(B)
        n = self.nstates
        if A.shape != (n, n) or B.shape != (n, n) or lam.shape != (n,):
            raise ValueError('expected the array to be square')
        D = None if D is None else _f64(D)
        if D is not None and D.shape != (n,):
            raise ValueError('D must have one entry per state')
        if t is None:
            t = self.tree.branch_lengths()
        t = _f64(t)
        if t.shape != (self.tree.nnodes,):
            raise ValueError('t must have one entry per node')
        _lib.check(_lib.lib().rt_model_set_rates_spectral(
            self._h, _ptr(A, c_double), _ptr(lam, c_double), _ptr(B, c_double),
            None if D is None else _ptr(D, c_double), _ptr(t, c_double)))

    def recompute_transitions(self):
        _lib.check(_lib.lib().rt_model_recompute_transitions(self._h))

    def set_transitions(self, esd):
        esd = _f64(esd)
        if esd.shape != (self.tree.nnodes, self.nstates, self.nstates):
            raise ValueError('esd_transitions has the wrong shape')
        _lib.check(_lib.lib().rt_model_set_transitions(self._h,
                                                       _ptr(esd, c_double)))

    def get_transitions(self):
        esd = np.empty((self.tree.nnodes, self.nstates, self.nstates))
        _lib.check(_lib.lib().rt_model_get_transitions(self._h,
                                                       _ptr(esd, c_double)))
        return esd

    def expm_info(self):
        info = np.zeros((self.tree.nnodes, 2), dtype=np.int32)
        _lib.check(_lib.lib().rt_model_get_expm_info(self._h,
                                                     _ptr(info, c_int32)))
        return info

    def set_root_distn(self, root_distn=None):
        if root_distn is None:
            _lib.check(_lib.lib().rt_model_set_root_distn(self._h, None))
            return
        w = _f64(root_distn)
        if w.shape != (self.nstates,):
            raise ValueError('root shape mismatch: %s %s' % (
                (self.nstates,), w.shape))
        _lib.check(_lib.lib().rt_model_set_root_distn(self._h,
                                                      _ptr(w, c_double)))

    def upload_sites(self, obs_nodes, data, kind='dense'):
        """obs_nodes: tree nodes (nx ids) carrying per-site data, in the order
        of the data's second axis.  data: dense f64[nsites,nobs,n] | state
        uint8[nsites,nobs] (255 = unobserved) | mask uint64[nsites,nobs] (nstates
        <= 64) or uint64[nsites,nobs,ceil(nstates/64)] (bit s % 64 of word s // 64)."""
        code = _KINDS[kind]
        idx = _i64([self.tree.node_to_index[v] for v in obs_nodes])
        if kind == 'dense':
            data = _f64(data)
            ok = data.ndim == 3 and data.shape[2] == self.nstates
        elif kind == 'state':
            data = _as_uint8_states(data, self.nstates)
            ok = data.ndim == 2
        else:
            # one 64-bit word per node for nstates <= 64, ceil(nstates / 64) words above
            data = np.ascontiguousarray(data, dtype=np.uint64)
            words = (self.nstates + 63) // 64
            ok = (data.ndim == 2 and words == 1) or (data.ndim == 3 and data.shape[2] == words)
        if not ok or data.shape[1] != len(idx):
            raise ValueError('observation array has the wrong shape')
        nsites = data.shape[0]
        h = c_void_p()
        _lib.check(_lib.lib().rt_sites_create(
            self._h, nsites, code, len(idx), _ptr(idx, c_int64),
            data.ctypes.data_as(c_void_p), byref(h)))
        return SiteBatch(self, h, nsites)

    def prune(self, batch):
        """Asynchronous: upward pass + root reduce + batch sum on the device."""
        _lib.check(_lib.lib().rt_prune(self._h, batch._h))

    def step(self, batch, recompute_transitions=True):
        """One iteration of a repeated-evaluation loop in one call (rt_step):
        per-edge expm from the resident rates (optional) + prune; asynchronous."""
        _lib.check(_lib.lib().rt_step(self._h, batch._h,
                                      1 if recompute_transitions else 0))

    def expected_history_statistics(self, batch, recompute_transitions=True,
                                    return_status=False):
        """rt_expect_step: the reference's get_expected_history_statistics
        (_mjp_dense.py:410-539) summed over the resident batch -- (dwell f64[n], summed
        root posteriors f64[n], transitions f64[n, n]) -- with nothing but those numbers
        crossing PCIe.  Rates must have been set with set_rates; nstates <= 64 (for
        nstates <= 4 a dense batch is read as allowed sets: likelihood != 0)."""
        n = self.nstates
        dwell = np.empty(n)
        rootp = np.empty(n)
        trans = np.empty((n, n))
        status = np.zeros(batch.nsites, dtype=np.int32) if return_status else None
        _lib.check(_lib.lib().rt_expect_step(
            self._h, batch._h, 1 if recompute_transitions else 0, _ptr(dwell, c_double),
            _ptr(rootp, c_double), _ptr(trans, c_double),
            None if status is None else _ptr(status, c_int32)))
        return (dwell, rootp, trans, status) if return_status else (dwell, rootp, trans)

    def allreduce(self, batch):
        _lib.check(_lib.lib().rt_allreduce_totals(self.ctx._h, batch._h))

    def allreduce_group(self, batches):
        """The totals of several batches in one collective (rt_allreduce_totals_group)."""
        arr = (c_void_p * len(batches))(*[b._h for b in batches])
        _lib.check(_lib.lib().rt_allreduce_totals_group(self.ctx._h, arr, len(batches)))

    def fetch_log_likelihoods(self, batch):
        ll = np.empty(batch.nsites, dtype=np.float64)
        st = np.empty(batch.nsites, dtype=np.int32)
        _lib.check(_lib.lib().rt_sites_get_logliks(
            batch._h, _ptr(ll, c_double), _ptr(st, c_int32)))
        return ll, st

    def fetch_totals(self, batch):
        tot = np.zeros(3, dtype=np.float64)
        _lib.check(_lib.lib().rt_sites_get_totals(batch._h, _ptr(tot, c_double)))
        return tot

    def log_likelihoods(self, batch):
        self.prune(batch)
        return self.fetch_log_likelihoods(batch)

    def total_log_likelihood(self, batch):
        """(sum of log-likelihoods, number of zero-probability sites); the sum
        is -inf when any site has zero probability."""
        self.prune(batch)
        tot = self.fetch_totals(batch)
        nzero = int(tot[1])
        return (-np.inf if nzero else float(tot[0])), nzero
