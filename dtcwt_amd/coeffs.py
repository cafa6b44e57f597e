"""Wavelet tap tables: ``biort(name)`` / ``qshift(name)``.

Mirrors the reference loader's interface and error behaviour (dtcwt/coeffs.py:13-90):
returns a tuple of float64 column vectors (m, 1); unknown name -> IOError; a name of
the other family -> ValueError.  The numbers come from ``data/wavelets.npz``, a single
table built from the published tap sets by ``tools/make_wavelet_table.py``.
"""
import os

import numpy as np

_TABLE_FILE = os.path.join(os.path.dirname(__file__), 'data', 'wavelets.npz')
_TABLE = None

_BIORT_KEYS = ('h0o', 'g0o', 'h1o', 'g1o')
_BIORT_BP_KEYS = _BIORT_KEYS + ('h2o', 'g2o')
_QSHIFT_KEYS = ('h0a', 'h0b', 'g0a', 'g0b', 'h1a', 'h1b', 'g1a', 'g1b')
_QSHIFT_BP_KEYS = _QSHIFT_KEYS + ('h2a', 'h2b', 'g2a', 'g2b')


def _table():
    global _TABLE
    if _TABLE is None:
        with np.load(_TABLE_FILE) as f:
            _TABLE = {k: np.array(f[k], dtype=np.float64) for k in f.files}
    return _TABLE


def _names():
    return sorted(set(k.split('/')[0] for k in _table()))


def _lookup(name, keys):
    if not isinstance(name, str):
        # the transforms rely on this to detect "already a tuple of vectors"
        raise TypeError('wavelet name must be a string')
    tab = _table()
    if name not in _names():
        raise IOError('No such wavelet: {0}'.format(name))
    try:
        return tuple(tab[name + '/' + k].reshape(-1, 1).copy() for k in keys)
    except KeyError:
        raise ValueError('Wavelet does not define ({0}) coefficients'.format(', '.join(keys)))


def biort(name):
    """Level-1 (odd-length bi-orthogonal) filters (h0o, g0o, h1o, g1o[, h2o, g2o]).

    antonini, legall, near_sym_a, near_sym_b, near_sym_b_bp (dtcwt/coeffs.py:27-56)."""
    return _lookup(name, _BIORT_BP_KEYS if name == 'near_sym_b_bp' else _BIORT_KEYS)


def qshift(name):
    """Level>=2 (even-length quarter-shift) filters (h0a, h0b, g0a, g0b, h1a, h1b, g1a,
    g1b[, h2a, h2b, g2a, g2b]).

    qshift_06, qshift_a, qshift_b, qshift_c, qshift_d, qshift_b_bp, qshift_32
    (dtcwt/coeffs.py:58-90)."""
    return _lookup(name, _QSHIFT_BP_KEYS if name == 'qshift_b_bp' else _QSHIFT_KEYS)
