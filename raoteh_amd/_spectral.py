"""
The reference's optional spectral path for ONE time-reversible rate matrix at many
branch lengths (examples/p53/qtop.py), host side.

The decomposition is computed once per rate matrix on the host, as the reference does
(scipy.linalg.eigh there, numpy.linalg.eigh here: both LAPACK's symmetric eigensolver);
the per-branch-length reconstruction -- the part that repeats for every edge and every
optimiser step -- is the device's (csrc/spectral.hip): `getp_spectral_v2` below for host
arrays in and out, `TreeModel.set_rates_spectral` for the resident hot path.
"""
from __future__ import annotations

import numpy as np

from . import device


def pseudo_reciprocal(v):
    """1 / v with 0 where v == 0 (qtop.py:104-107)."""
    v = np.asarray(v, dtype=np.float64)
    out = np.zeros_like(v)
    nz = v != 0
    out[nz] = 1.0 / v[nz]
    return out


def decompose_spectral(S, D):
    """qtop.py:128-140.  Q = S diag(D), S symmetric, D >= 0 -> (D, U, lam) with
    diag(sqrt D) S diag(sqrt D) = U diag(lam) U^T."""
    S = np.asarray(S, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64)
    n = D.shape[0]
    if S.shape != (n, n):
        raise ValueError('expected the array to be square')
    if (D < 0).any():
        raise ValueError('D must be non-negative')
    r = np.sqrt(D)
    lam, U = np.linalg.eigh(r[:, None] * S * r[None, :])
    return D, U, lam


def decompose_spectral_v2(S, D):
    """qtop.py:142-150: the factors of the reconstruction, A = diag(1 / sqrt D) U
    (zero rows where D == 0), B = U^T diag(sqrt D)."""
    D, U, lam = decompose_spectral(S, D)
    r = np.sqrt(D)
    return pseudo_reciprocal(r)[:, None] * U, lam, U.T * r[None, :]


def decompose_rate_matrix(Q, distn):
    """(A, lam, B, D) of a reversible rate matrix given with its stationary
    distribution: S = Q diag(1 / distn), symmetrised (detailed balance makes it
    symmetric up to rounding); ValueError when Q is not reversible under distn."""
    Q = np.asarray(Q, dtype=np.float64)
    D = np.asarray(distn, dtype=np.float64)
    S = Q * pseudo_reciprocal(D)[None, :]
    scale = max(1.0, float(np.abs(S).max()))
    if np.abs(S - S.T).max() > 1e-9 * scale:
        raise ValueError('the rate matrix is not time-reversible under this distribution')
    A, lam, B = decompose_spectral_v2(0.5 * (S + S.T), D)
    return A, lam, B, D


def getp_spectral_v2(D, A, lam, B, t, ctx=None):
    """qtop.py:76-88 on the device.  t scalar -> P [n, n]; t array -> P [len(t), n, n]."""
    ctx = ctx or device.get_context()
    P = ctx.expm_spectral(A, lam, B, np.atleast_1d(t), D=D)
    return P[0] if np.ndim(t) == 0 else P


def getp_spectral(D, U, lam, t, ctx=None):
    """qtop.py:60-74: the same from (D, U, lam)."""
    r = np.sqrt(np.asarray(D, dtype=np.float64))
    U = np.asarray(U, dtype=np.float64)
    return getp_spectral_v2(D, pseudo_reciprocal(r)[:, None] * U, lam, U.T * r[None, :], t, ctx)
