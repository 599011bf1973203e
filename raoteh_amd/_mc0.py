"""
Root reduction of the sparse (dict) API: raoteh/sampler/_mc0.py:202-252.
Host arithmetic over at most nstates terms; the message passes that produce
``root_pmap`` run on the GPU (_mcx / _mcy / _mcz of this package).
"""
from __future__ import annotations

from ._util import StructuralZeroProb

__all__ = ['get_likelihood']


def get_likelihood(root_pmap, root_distn=None):
    if (root_distn is not None) and not root_distn:
        raise StructuralZeroProb('no root state has nonzero prior likelihood')
    if root_pmap is None:
        raise ValueError('root_pmap is None')
    if not root_pmap:
        raise StructuralZeroProb(
            'all root states give a subtree likelihood of zero')
    feasible_rstates = set(root_pmap)
    if root_distn is not None:
        feasible_rstates.intersection_update(set(root_distn))
    if not feasible_rstates:
        raise StructuralZeroProb(
            'all root states have either zero prior likelihood '
            'or give a subtree likelihood of zero')
    if root_distn is not None:
        return sum(root_pmap[s] * root_distn[s] for s in feasible_rstates)
    return sum(root_pmap.values())
