"""
Root reduction of the sparse (dict) API: raoteh/sampler/_mc0.py:202-252.
Host arithmetic over at most nstates terms; the message passes that produce
``root_pmap`` run on the GPU (_mcx / _mcy / _mcz of this package).
"""
from __future__ import annotations

from ._util import StructuralZeroProb

__all__ = ['get_likelihood']


def get_likelihood(root_pmap, root_distn=None):
    """Likelihood from the root's subtree likelihoods {state: value} and optional root
    weights {state: weight} (absent = weights of one, not a uniform prior).  Same
    checks, in the same order, with the same exception classes as the reference
    (_mc0.py:222-243): an empty prior, a missing / empty pmap, then no state common
    to both all mean the likelihood is zero by sparsity."""
    have_prior = root_distn is not None
    if have_prior and len(root_distn) == 0:
        raise StructuralZeroProb('no root state has nonzero prior likelihood')
    if root_pmap is None:
        raise ValueError('root_pmap is None')
    if len(root_pmap) == 0:
        raise StructuralZeroProb(
            'all root states give a subtree likelihood of zero')
    if not have_prior:
        return sum(root_pmap.values())
    terms = [value * root_distn[state] for state, value in root_pmap.items()
             if state in root_distn]
    if not terms:
        raise StructuralZeroProb(
            'all root states have either zero prior likelihood '
            'or give a subtree likelihood of zero')
    return sum(terms)
