"""
Type-z observations (node -> {state: likelihood}), SPARSE API: same names and
argument order as raoteh/sampler/_mcz.py (get_node_to_set :30-42,
get_node_to_pset :45-91, get_node_to_pmap :94-166, get_likelihood :169-209).
The allowed states of a node are the KEYS of its likelihood dict (:34-36); the
upward pass multiplies the per-state observation likelihood in (:159-160) and
runs on the GPU (rt_mcy_esd_get_node_to_pmap with obs_likelihood).

``get_likelihood``: the reference's signature names its observation argument
``node_to_allowed_states`` while its body reads ``node_to_state_to_likelihood``
(:169-170 against :204-206, a NameError as published); here the third
positional argument is the likelihood map and both keyword names are accepted.
"""
from __future__ import annotations

import numpy as np

from . import _mc0, _mcy
from ._sparse import SparseProblem
from .device import get_context

__all__ = ['get_node_to_set', 'get_node_to_pset', 'get_node_to_pmap',
           'get_likelihood']


def _allowed(node_to_state_to_likelihood):
    if node_to_state_to_likelihood is None:
        return None
    return dict((node, set(m)) for node, m in
                node_to_state_to_likelihood.items())


def get_node_to_set(T, root, node_to_state_to_likelihood=None, P_default=None):
    return _mcy.get_node_to_set(
        T, root, node_to_allowed_states=_allowed(node_to_state_to_likelihood),
        P_default=P_default)


def get_node_to_pset(T, root, node_to_state_to_likelihood=None,
                     P_default=None):
    return _mcy.get_node_to_pset(
        T, root, node_to_allowed_states=_allowed(node_to_state_to_likelihood),
        P_default=P_default)


def get_node_to_pmap(T, root, node_to_state_to_likelihood=None, P_default=None,
                     node_to_set=None):
    if root not in T:
        raise ValueError('unrecognized root')
    prob = SparseProblem(T, root, P_default=P_default)
    ta = prob.ta
    if node_to_set is None:
        mask = prob.mask_from_allowed(_allowed(node_to_state_to_likelihood))
        ctx = get_context()
        ctx.node_to_pset(ta.indices, ta.indptr, prob.esd, mask)
        ctx.node_to_set(ta.indices, ta.indptr, prob.esd, mask)
    else:
        mask = prob.mask_from_allowed(node_to_set)
    obs = np.ones(mask.shape, dtype=np.float64)
    for i, na in enumerate(ta.preorder_nodes):
        # the reference indexes the dict for every (node, state) it visits
        # (:159): a missing node or state is a KeyError there too
        for j, s in enumerate(prob.sorted_states):
            if mask[i, j]:
                obs[i, j] = node_to_state_to_likelihood[na][s]
    pmap = np.empty(mask.shape, dtype=np.float64)
    get_context().node_to_pmap(ta.indices, ta.indptr, prob.esd, mask, pmap,
                               obs_likelihood=obs)
    return prob.pmap_to_dict(mask, pmap)


def get_likelihood(T, root, node_to_state_to_likelihood=None, root_distn=None,
                   P_default=None, node_to_allowed_states=None):
    if node_to_state_to_likelihood is None:
        node_to_state_to_likelihood = node_to_allowed_states
    node_to_pmap = get_node_to_pmap(
        T, root, node_to_state_to_likelihood=node_to_state_to_likelihood,
        P_default=P_default)
    return _mc0.get_likelihood(node_to_pmap[root], root_distn=root_distn)
