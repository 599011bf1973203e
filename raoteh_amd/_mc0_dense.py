"""
Root reduction and its error protocol -- host-side mirror of
raoteh/sampler/_mc0_dense.py:147-212 (``get_likelihood(root_pmap, root_distn)``).
The pmap itself comes from the HIP upward pass; this function only applies the
reference's zero-probability checks and the final n-term weighted sum.
"""
from __future__ import annotations

import warnings

import numpy as np

from ._util import StructuralZeroProb

__all__ = ['get_likelihood']


def get_likelihood(root_pmap, root_distn=None):
    if root_distn is not None:
        if root_pmap.shape != root_distn.shape:
            raise ValueError('root shape mismatch: '
                             '%s %s' % (root_pmap.shape, root_distn.shape))
        prior_feasible_rstates = set(s for s, p in enumerate(root_distn) if p)
        if not prior_feasible_rstates:
            raise StructuralZeroProb(
                'no root state has nonzero prior likelihood')
    if root_pmap is None:
        raise ValueError('root_pmap is None')
    root_pmap_min = root_pmap.min()
    if root_pmap_min < 0:
        warnings.warn('root_pmap should have non-negative entries '
                      'but found minimum entry %s' % root_pmap_min)
        root_pmap = np.maximum(root_pmap, 0)
    if not root_pmap.sum():
        raise StructuralZeroProb(
            'all root states give a subtree likelihood of zero')
    feasible_rstates = set(s for s, p in enumerate(root_pmap) if p)
    if root_distn is not None:
        feasible_rstates.intersection_update(prior_feasible_rstates)
    if not feasible_rstates:
        raise StructuralZeroProb(
            'all root states have either zero prior likelihood '
            'or give a subtree likelihood of zero')
    if root_distn is not None:
        return root_distn.dot(root_pmap)
    return root_pmap.sum()
