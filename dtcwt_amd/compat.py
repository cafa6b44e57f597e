"""MATLAB-style function wrappers (interface of dtcwt/compat.py:32-288).

Unlike the reference, whose wrappers are hard-wired to the NumPy backend
(dtcwt/compat.py:17), these follow the backend currently selected through
:func:`dtcwt_amd.push_backend` (default ``'hip'``), so scripts written against
``dtwavexfm2`` / ``dtwaveifm2`` run on the GPU unchanged.
"""
import dtcwt_amd
from dtcwt_amd.defaults import DEFAULT_BIORT, DEFAULT_QSHIFT

__all__ = ['dtwavexfm', 'dtwaveifm', 'dtwavexfm2', 'dtwaveifm2', 'dtwavexfm2b', 'dtwaveifm2b',
           'dtwavexfm3', 'dtwaveifm3']


def _unpack(res, include_scale):
    if include_scale:
        return res.lowpass, res.highpasses, res.scales
    return res.lowpass, res.highpasses


def dtwavexfm(X, nlevels=3, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, include_scale=False):
    """*n*-level 1-D DT-CWT of a column vector or the columns of a matrix
    (dtcwt/compat.py:32-68): returns ``Yl, Yh[, Yscale]``."""
    return _unpack(dtcwt_amd.Transform1d(biort, qshift).forward(X, nlevels, include_scale), include_scale)


def dtwaveifm(Yl, Yh, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, gain_mask=None):
    """1-D reconstruction (dtcwt/compat.py:70-105)."""
    return dtcwt_amd.Transform1d(biort, qshift).inverse(dtcwt_amd.Pyramid(Yl, Yh), gain_mask)


def dtwavexfm2(X, nlevels=3, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, include_scale=False):
    """*n*-level 2-D DT-CWT (dtcwt/compat.py:107-143): returns ``Yl, Yh[, Yscale]``."""
    return _unpack(dtcwt_amd.Transform2d(biort, qshift).forward(X, nlevels, include_scale), include_scale)


def dtwaveifm2(Yl, Yh, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, gain_mask=None):
    """2-D reconstruction (dtcwt/compat.py:145-184)."""
    return dtcwt_amd.Transform2d(biort, qshift).inverse(dtcwt_amd.Pyramid(Yl, Yh), gain_mask)


# the 'b' (band-pass) forms are folded into the plain ones (dtcwt/compat.py:186-187)
dtwavexfm2b = dtwavexfm2
dtwaveifm2b = dtwaveifm2


def dtwavexfm3(X, nlevels=3, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, include_scale=False, ext_mode=4,
               discard_level_1=False):
    """*n*-level 3-D DT-CWT (dtcwt/compat.py:189-246): returns ``Yl, Yh[, Yscale]``."""
    res = dtcwt_amd.Transform3d(biort, qshift, ext_mode).forward(X, nlevels, include_scale, discard_level_1)
    return _unpack(res, include_scale)


def dtwaveifm3(Yl, Yh, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, ext_mode=4):
    """3-D reconstruction (dtcwt/compat.py:248-288)."""
    return dtcwt_amd.Transform3d(biort, qshift, ext_mode).inverse(dtcwt_amd.Pyramid(Yl, Yh))
