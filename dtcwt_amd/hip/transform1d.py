"""1-D DT-CWT on the device: ``Transform1d`` of the ``hip`` backend.

Interface, shapes, exceptions of dtcwt/numpy/transform1d.py:14-196.  A column vector (or
the columns of a matrix) is filtered down axis 0 by the generic device filters; the
real -> complex interleave (``Hi[::2] + 1j*Hi[1::2]``, :88,100) and its inverse ``c2q1d``
(:186-196) are small device kernels.
"""
import numpy as np

from dtcwt_amd.coeffs import biort as _biort, qshift as _qshift
from dtcwt_amd.defaults import DEFAULT_BIORT, DEFAULT_QSHIFT
from dtcwt_amd.utils import asfarray
from dtcwt_amd.hip import _lib
from dtcwt_amd.hip._lib import DeviceArray, check, dtype_code
from dtcwt_amd.hip.common import Pyramid, nlevels_of
from dtcwt_amd.hip import lowlevel as ll

__all__ = ['Transform1d']


def _pack(hi):
    """[2J, k] real -> [J, k] complex on the device."""
    J, k = hi.shape[0] // 2, hi.shape[1]
    cdt = np.complex64 if hi.dtype == np.float32 else np.complex128
    out = DeviceArray(hi.ctx, (J, k), cdt)
    check(_lib.lib().dtcwt_hip_pack1d(hi.ctx.handle, dtype_code(hi.dtype), hi.ptr, J, k, out.ptr))
    return out


def _unpack(yh, gain):
    """[J, k] complex -> [2J, k] real * gain on the device."""
    J, k = yh.shape
    rdt = np.float32 if yh.dtype == np.complex64 else np.float64
    out = DeviceArray(yh.ctx, (2 * J, k), rdt)
    check(_lib.lib().dtcwt_hip_unpack1d(yh.ctx.handle, dtype_code(yh.dtype), yh.ptr, J, k, float(gain), out.ptr))
    return out


class Transform1d(object):
    """An implementation of the 1D DT-CWT on AMD GPUs via HIP.

    :param biort: Level 1 wavelets (name or (h0o, g0o, h1o, g1o)).
    :param qshift: Level >= 2 wavelets (name or 8-tuple).
    """

    def __init__(self, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, ctx=None):
        self.biort = biort
        self.qshift = qshift
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    def _taps(self):
        # resolved per call like the reference (transform1d.py:56-65)
        try:
            b = _biort(self.biort)
        except TypeError:
            b = self.biort
        try:
            q = _qshift(self.qshift)
        except TypeError:
            q = self.qshift
        h0o, g0o, h1o, g1o = b
        h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = q
        return b, q

    def forward(self, X, nlevels=3, include_scale=False):
        """Perform a *n*-level DTCWT decomposition on a 1D column vector *X* (or on the
        columns of a matrix *X*).  Returns a device-resident :class:`Pyramid`."""
        (h0o, g0o, h1o, g1o), (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b) = self._taps()
        if isinstance(X, DeviceArray):
            Xd = X if X.ndim == 2 else X.reshape(X.shape[0], 1)
        else:
            X = np.asanyarray(X)
            if np.issubdtype(X.dtype, np.complexfloating):
                X = asfarray(X)
            if X.ndim == 1:
                X = np.atleast_2d(X).T
            Xd = None
        shape0 = Xd.shape[0] if Xd is not None else X.shape[0]
        if shape0 % 2 != 0:
            raise ValueError('Size of input X must be a multiple of 2')
        if nlevels == 0:
            Xh = Xd.get() if Xd is not None else asfarray(X)
            return Pyramid(Xh, (), ()) if include_scale else Pyramid(Xh, ())
        if Xd is None:
            Xd = self.ctx.to_device_float(X)     # asfarray semantics; integers are widened on the device
        Yh, Ys = [], []
        # a level = one launch with the highpass packing fused (dtcwt_hip_level1d_forward);
        # filters / shapes it declines go through the pair filter + pack kernels
        whole = ll.level1d_forward(Xd, 0, (0, 0), h0o, h1o)
        if whole is not None:
            Lo, y = whole
        else:
            Lo, Hi = ll.axis_colfilter2(Xd, h0o, h1o, axis=0)
            y = _pack(Hi)
        Yh.append(y)
        Ys.append(Lo)
        for level in range(1, nlevels):
            pad = (1, 1) if Lo.shape[0] % 4 != 0 else (0, 0)      # transform1d.py:95-96
            whole = ll.level1d_forward(Lo, 1, pad, (h0b, h0a), (h1b, h1a))
            if whole is not None:
                Lo, y = whole
            else:
                Lo, Hi = ll.axis_coldfilt2(Lo, (h0b, h0a), (h1b, h1a), axis=0, pad=pad)
                y = _pack(Hi)
            Yh.append(y)
            Ys.append(Lo)
        if include_scale:
            return Pyramid(Lo, tuple(Yh), tuple(Ys))
        return Pyramid(Lo, tuple(Yh))

    def inverse(self, pyramid, gain_mask=None, device_output=False):
        """Perform an *n*-level dual-tree complex wavelet (DTCWT) 1D reconstruction.
        ``gain_mask[l]`` is the gain of level *l* (default ones)."""
        (h0o, g0o, h1o, g1o), (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b) = self._taps()
        a = nlevels_of(pyramid)
        if a == 0:
            return pyramid.lowpass
        gain_mask = np.ones(a) if gain_mask is None else np.asarray(gain_mask, dtype=np.float64)
        if hasattr(pyramid, 'device_parts'):
            probe = pyramid.hip_lowpass if pyramid.hip_lowpass is not None else pyramid.lowpass
            rdt = np.float32 if probe.dtype in (np.float32, np.complex64) else np.float64
            Lo, Yh = pyramid.device_parts(self.ctx, rdt)
        else:
            low = asfarray(pyramid.lowpass)
            rdt = np.float32 if low.dtype == np.float32 else np.float64
            cdt = np.complex64 if rdt == np.float32 else np.complex128
            Lo = self.ctx.to_device(low, dtype=rdt)
            Yh = tuple(self.ctx.to_device(np.asarray(y), dtype=cdt) for y in pyramid.highpasses)
        if Lo.ndim == 1:
            Lo = Lo.reshape(Lo.shape[0], 1)
        Yh = [y if y.ndim == 2 else y.reshape(y.shape[0], 1) for y in Yh]
        level = a - 1
        while level >= 1:                                     # transform1d.py:150-160
            want = 2 * Yh[level - 1].shape[0]
            full = 2 * Lo.shape[0]
            crop = (1, 1) if full != want else (0, 0)
            if full - 2 * crop[0] != want or Lo.shape[1] != Yh[level - 1].shape[1] or \
                    (2 * Yh[level].shape[0], Yh[level].shape[1]) != Lo.shape:
                raise ValueError('Yh sizes are not valid for DTWAVEIFM')
            whole = ll.level1d_inverse(Lo, Yh[level], 1, gain_mask[level], crop[0], (g0b, g0a), (g1b, g1a))
            if whole is not None:       # the level in one launch (dtcwt_hip_level1d_inverse)
                Lo = whole
            else:
                Hi = _unpack(Yh[level], gain_mask[level])
                Lo = ll.axis_colifilt_sum2(Lo, Hi, (g0b, g0a), (g1b, g1a), axis=0, crop=crop)
            level -= 1
        if (2 * Yh[0].shape[0], Yh[0].shape[1]) != Lo.shape:
            raise ValueError('Yh sizes are not valid for DTWAVEIFM')
        Z = ll.level1d_inverse(Lo, Yh[0], 0, gain_mask[0], 0, g0o, g1o)
        if Z is None:
            Z = ll.axis_colfilter_sum2(Lo, _unpack(Yh[0], gain_mask[0]), g0o, g1o, axis=0)
        if device_output:
            return Z
        Zh = Z.get()
        return Zh.flatten() if Zh.shape[1] == 1 else Zh
