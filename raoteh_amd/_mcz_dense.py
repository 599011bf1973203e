"""
Type-z observations (node -> state -> likelihood), DENSE ndarray transition
matrices (the reference has no dense twin of _mcz; this is the encoding the
batched hot path uses; the reference-signature sparse API is _mcz.py).  Mirror of
raoteh/sampler/_mcz.py:94-166 (``get_node_to_pmap``): the upward pass multiplies
the per-state observation likelihood in (:159-160).  Dense ndarray transition
matrices on the edges (the batched hot path's native encoding); the allowed-set
passes run first on the support of the likelihoods, as ``_mcz.get_node_to_set``
does (:60-91).
"""
from __future__ import annotations

import numpy as np

from ._mcy_dense import _run_passes
from ._tree import TreeArrays

__all__ = ['get_node_to_pmap']


def get_node_to_pmap(T, root, nstates, node_to_state_to_likelihood=None,
                     P_default=None, node_to_set=None):
    ta = TreeArrays(T, root)
    nnodes = ta.nnodes
    obs = np.ones((nnodes, nstates), dtype=np.float64)
    state_mask = np.ones((nnodes, nstates), dtype=np.int64)
    for i, na in enumerate(ta.preorder_nodes):
        if node_to_state_to_likelihood is not None:
            # the reference indexes the dict for every node it visits
            # (_mcz.py:159): a missing node is a KeyError there too
            lik = node_to_state_to_likelihood[na]
            for s in range(nstates):
                obs[i, s] = lik.get(s, 0.0) if hasattr(lik, 'get') else lik[s]
            state_mask[i] = obs[i] != 0
        if node_to_set is not None:
            allowed = node_to_set[na]
            for s in range(nstates):
                if s not in allowed:
                    state_mask[i, s] = 0
    esd = ta.esd_transitions(nstates, P_default=P_default)
    pmap = _run_passes(ta, esd, state_mask, obs_likelihood=obs)
    return dict((na, pmap[i]) for i, na in enumerate(ta.preorder_nodes))
