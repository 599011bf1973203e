"""
Exception classes of the likelihood path -- same names and hierarchy as the
reference (raoteh/sampler/_util.py:14-21), because callers catch them
(_sampler.py:637-643, examples/p53/liwen.py:381-418).
"""

__all__ = ['ZeroProbError', 'StructuralZeroProb', 'NumericalZeroProb']


class ZeroProbError(Exception):
    pass


class StructuralZeroProb(ZeroProbError):
    pass


class NumericalZeroProb(ZeroProbError):
    pass


def get_first_element(elements):
    # raoteh/sampler/_util.py:23-25
    for x in elements:
        return x
