"""
raoteh_amd -- MI355X-native (gfx950, hand-written HIP) implementation of the
tree-structured CTMC likelihood hot path of argriffing/raoteh: per-edge
expm(Q*t) + Felsenstein leaf-to-root message pass + root reduction, batched over
independent sites.  Host code is Python over a ctypes C ABI
(include/raoteh_hip.h); there is no CPU fallback.

Modules mirror the reference's (raoteh/sampler/...):
  dense ndarray API   _mjp_dense, _mcy_dense, _mcx_dense, _mc0_dense (+ _mcz_dense)
  sparse nx/dict API  _mjp, _mcy, _mcx, _mcz, _mc0
  _util
and ``device`` holds the batched device-resident objects.
"""

from ._util import ZeroProbError, StructuralZeroProb, NumericalZeroProb

__version__ = '0.1.0'
__all__ = ['ZeroProbError', 'StructuralZeroProb', 'NumericalZeroProb']
