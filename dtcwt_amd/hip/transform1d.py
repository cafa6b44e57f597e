"""1-D DT-CWT on the device: ``Transform1d`` of the ``hip`` backend.

Interface, shapes, exceptions of dtcwt/numpy/transform1d.py:14-196.  A column vector (or
the columns of a matrix) is filtered down axis 0 by the generic device filters; the
real -> complex interleave (``Hi[::2] + 1j*Hi[1::2]``, :88,100) and its inverse ``c2q1d``
(:186-196) are small device kernels.  Where every level has a one-launch level kernel (odd-length
biort filters, one signal or >= 32 side by side) the whole transform is ONE native call
(``dtcwt_hip_plan1d_*``: geometry, workspaces and level sequencing inside the library).
"""
import collections
import ctypes
import os

import numpy as np

from dtcwt_amd.coeffs import biort as _biort, qshift as _qshift
from dtcwt_amd.defaults import DEFAULT_BIORT, DEFAULT_QSHIFT
from dtcwt_amd.utils import asfarray, flat_taps
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


class _Plan1d(object):
    """RAII wrapper of dtcwt_hip_plan1d: the whole 1-D transform in one native call."""

    def __init__(self, ctx, dtype, n, k, nlevels, biort, qshift):
        L = _lib.lib()
        pd = ctypes.POINTER(ctypes.c_double)
        self.ctx, self.dtype, self.n, self.k, self.nlevels = ctx, np.dtype(dtype), n, k, nlevels
        self._keep = [flat_taps(h) for h in biort[:4]] + [flat_taps(h) for h in qshift[:8]]
        bp = (pd * 4)(*[a.ctypes.data_as(pd) for a in self._keep[:4]])
        bl = (ctypes.c_int * 4)(*[a.shape[0] for a in self._keep[:4]])
        qp = (pd * 8)(*[a.ctypes.data_as(pd) for a in self._keep[4:]])
        ql = (ctypes.c_int * 8)(*[a.shape[0] for a in self._keep[4:]])
        h = ctypes.c_void_p()
        rc = L.dtcwt_hip_plan1d_create(ctx.handle, dtype_code(dtype), n, k, nlevels, bp, bl, qp, ql, ctypes.byref(h))
        if rc == -3:
            raise NotImplementedError(L.dtcwt_hip_last_error().decode())
        check(rc)
        self._h, self._lib = h, L
        s = (ctypes.c_int64 * (1 + 2 * nlevels))()
        check(L.dtcwt_hip_plan1d_shapes(h, s))
        self.low = int(s[0])
        self.high = [int(s[1 + 2 * l]) for l in range(nlevels)]
        self.scale = [int(s[2 + 2 * l]) for l in range(nlevels)]

    def forward(self, Xd, include_scale):
        ctx, nl, k = self.ctx, self.nlevels, self.k
        cdt = np.complex64 if self.dtype == np.float32 else np.complex128
        Yl = DeviceArray(ctx, (self.low, k), self.dtype)
        Yh = [DeviceArray(ctx, (self.high[l], k), cdt) for l in range(nl)]
        Ys = [DeviceArray(ctx, (self.scale[l], k), self.dtype) for l in range(nl)] if include_scale else None
        vp = ctypes.c_void_p
        yh_p = (vp * nl)(*[a.ptr for a in Yh])
        ys_p = (vp * nl)(*[a.ptr for a in Ys]) if Ys else None
        rc = self._lib.dtcwt_hip_plan1d_forward(self._h, Xd.ptr, Yl.ptr, yh_p, ys_p)
        if rc == -3:
            return None
        check(rc)
        return Yl, Yh, Ys

    def inverse(self, Lo, Yh, gain_mask):
        nl = self.nlevels
        Z = DeviceArray(self.ctx, (self.n, self.k), self.dtype)
        vp = ctypes.c_void_p
        yh_p = (vp * nl)(*[a.ptr for a in Yh])
        # the reference indexes gain_mask[level] (transform1d.py:150-176): longer masks are allowed, extra entries unused
        g = np.ascontiguousarray(np.asarray(gain_mask, dtype=np.float64).ravel()[:nl])
        rc = self._lib.dtcwt_hip_plan1d_inverse(self._h, Lo.ptr, yh_p, g.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), Z.ptr)
        if rc == -3:
            return None
        check(rc)
        return Z

    def __del__(self):
        try:
            if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
                self._lib.dtcwt_hip_plan1d_destroy(self._h)
            self._h = None
        except Exception:
            pass


class Transform1d(object):
    """An implementation of the 1D DT-CWT on AMD GPUs via HIP.

    :param biort: Level 1 wavelets (name or (h0o, g0o, h1o, g1o)).
    :param qshift: Level >= 2 wavelets (name or 8-tuple).
    """

    def __init__(self, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, ctx=None):
        self.biort = biort
        self.qshift = qshift
        self._ctx = ctx
        self._plans = collections.OrderedDict()

    MAX_PLANS = 8

    def _plan(self, dtype, n, k, nlevels, b, q):
        """The native whole-transform plan (dtcwt_hip_plan1d_*) or None where some level has no one-launch kernel."""
        if os.environ.get('DTCWT_HIP_PLAN1D', '1') == '0' or len(b) != 4 or len(q) != 8:
            return None
        key = (np.dtype(dtype).str, n, k, nlevels, tuple(np.asarray(h).tobytes() for h in tuple(b) + tuple(q)))
        if key not in self._plans:
            try:
                self._plans[key] = _Plan1d(self.ctx, dtype, n, k, nlevels, b, q)
            except NotImplementedError:         # -3: no one-launch kernel for some level; permanent for this key
                self._plans[key] = None
            except _lib.HipError:               # transient (e.g. out of memory): level-by-level now, retry next time
                return None
            while len(self._plans) > self.MAX_PLANS:
                self._plans.popitem(last=False)
        else:
            self._plans.move_to_end(key)
        return self._plans[key]

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
        plan = self._plan(Xd.dtype, Xd.shape[0], Xd.shape[1], nlevels, (h0o, g0o, h1o, g1o),
                          (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b)) if Xd.dtype in (np.float32, np.float64) else None
        done = plan.forward(Xd, include_scale) if plan is not None else None
        if done is not None:            # the whole transform in one native call
            Lo, Yh, Ys = done
            return Pyramid(Lo, tuple(Yh), tuple(Ys)) if include_scale else Pyramid(Lo, tuple(Yh))
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
        n0, k = 2 * Yh[0].shape[0], Lo.shape[1]
        plan = self._plan(Lo.dtype, n0, k, a, (h0o, g0o, h1o, g1o), (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b)) \
            if Lo.dtype in (np.float32, np.float64) else None
        cdt = np.complex64 if Lo.dtype == np.float32 else np.complex128
        if plan is not None and plan.low == Lo.shape[0] and np.size(gain_mask) >= a and all(
                tuple(Yh[l].shape) == (plan.high[l], k) and Yh[l].dtype == cdt for l in range(a)):
            Z = plan.inverse(Lo, Yh, gain_mask)
            if Z is not None:
                if device_output:
                    return Z
                Zh = Z.get()
                return Zh.flatten() if Zh.shape[1] == 1 else Zh
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
