"""
Uniformization for the Rao-Teh sweep -- host-side mirror of
raoteh/sampler/_sample_mjp_dense.py:72-114 (``get_uniformized_transition_matrix``).
The sweep itself re-samples states on chunk trees with this one matrix on every
edge: that is the device core of ``raoteh_amd._forest``.
"""
from __future__ import annotations

import numpy as np

from ._tree import check_square_dense

__all__ = ['get_uniformized_transition_matrix']


def get_uniformized_transition_matrix(Q, uniformization_factor=None, omega=None):
    """P = I + Q / omega with omega = uniformization_factor * (largest exit rate),
    uniformization_factor 2 unless given (reference :72-114); passing both the factor
    and omega is an error there too."""
    if uniformization_factor is not None and omega is not None:
        raise ValueError('the uniformization factor and omega '
                         'should not both be provided')
    check_square_dense(Q)
    Q = np.asarray(Q, dtype=np.float64)
    if omega is None:
        factor = 2 if uniformization_factor is None else uniformization_factor
        # total rate out of a state = minus its diagonal entry (_mjp_dense.py:28-44)
        omega = factor * (-np.diag(Q)).max()
    return np.identity(Q.shape[0]) + Q / omega
