"""Low-level column filters of the ``hip`` backend: ``colfilter``, ``coldfilt``,
``colifilt`` with the signatures, shape rules and ``ValueError`` behaviour of
dtcwt/numpy/lowlevel.py:47-260, executed by the gfx950 kernels of libdtcwt_hip.so.

NumPy in -> NumPy out (like dtcwt/opencl/lowlevel.py:24-148, which returns host arrays);
:class:`DeviceArray` in -> :class:`DeviceArray` out (stays in HBM).  float32 and float64
are preserved, anything else is promoted to float64 (dtcwt/utils.py:98-105).

The ``axis_*`` functions are the device-side building blocks used by the transforms:
they filter along any axis of a C-contiguous device array, with the logical edge
replication / output cropping the level loops need done by index arithmetic.
"""
import ctypes

import numpy as np

from dtcwt_amd.utils import asfarray
from dtcwt_amd.hip import _lib
from dtcwt_amd.hip._lib import DeviceArray, View, check, dtype_code, taps_arg

__all__ = ['colfilter', 'coldfilt', 'colifilt', 'axis_colfilter', 'axis_coldfilt',
           'axis_colifilt', 'axis_colfilter2', 'axis_colfilter_sum2', 'axis_coldfilt2',
           'axis_colifilt_sum2', 'q2c', 'c2q', 'level2d_forward', 'level2d_inverse', 'level1d_forward', 'level1d_inverse']


def _prod(t):
    r = 1
    for s in t:
        r *= int(s)
    return r


def _view(X, axis, nwrite, pad, crop):
    shape = X.shape
    outer, n, inner = _prod(shape[:axis]), int(shape[axis]), _prod(shape[axis + 1:])
    v = View()
    v.outer, v.n, v.inner = outer, n, inner
    v.xso, v.xsn, v.xsi = n * inner, inner, 1
    v.yso, v.ysn, v.ysi = nwrite * inner, inner, 1
    v.pad_lo, v.pad_hi = int(pad[0]), int(pad[1])
    v.crop_lo, v.crop_hi = int(crop[0]), int(crop[1])
    return v


def _out(X, axis, nwrite, out, accumulate):
    shape = X.shape[:axis] + (nwrite,) + X.shape[axis + 1:]
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs an output array')
        return DeviceArray(X.ctx, shape, X.dtype)
    if out.shape != shape or out.dtype != X.dtype:
        raise ValueError('output array has shape %s dtype %s, need %s %s' % (out.shape, out.dtype, shape, X.dtype))
    return out


def axis_colfilter(X, h, axis=0, out=None, accumulate=False, pad=(0, 0), crop=(0, 0)):
    """Device colfilter along *axis* (dtcwt/numpy/lowlevel.py:47-80)."""
    axis = axis % X.ndim
    keep, hp, m = taps_arg(h)
    L = X.shape[axis] + pad[0] + pad[1]
    nout = L if m % 2 else L + 1
    nwrite = nout - crop[0] - crop[1]
    Y = _out(X, axis, nwrite, out, accumulate)
    v = _view(X, axis, nwrite, pad, crop)
    check(_lib.lib().dtcwt_hip_colfilter(X.ctx.handle, dtype_code(X.dtype), X.ptr, Y.ptr, ctypes.byref(v),
                                         hp, m, _lib.ACCUMULATE if accumulate else 0))
    return Y


def _pair_args(ha, hb):
    ka, pa, ma = taps_arg(ha)
    kb, pb, mb = taps_arg(hb)
    if np.shape(ha) != np.shape(hb) and ma != mb:
        raise ValueError('Shapes of ha and hb must be the same')
    if ma != mb:
        raise ValueError('Shapes of ha and hb must be the same')
    if ma % 2 != 0:
        raise ValueError('Lengths of ha and hb must be even')
    return (ka, kb), pa, pb, ma


def axis_coldfilt(X, ha, hb, axis=0, out=None, accumulate=False, pad=(0, 0), crop=(0, 0)):
    """Device coldfilt along *axis* (dtcwt/numpy/lowlevel.py:82-154)."""
    axis = axis % X.ndim
    keep, pa, pb, m = _pair_args(ha, hb)
    L = X.shape[axis] + pad[0] + pad[1]
    if L % 4 != 0:
        raise ValueError('No. of rows in X must be a multiple of 4')
    nwrite = L // 2 - crop[0] - crop[1]
    Y = _out(X, axis, nwrite, out, accumulate)
    v = _view(X, axis, nwrite, pad, crop)
    check(_lib.lib().dtcwt_hip_coldfilt(X.ctx.handle, dtype_code(X.dtype), X.ptr, Y.ptr, ctypes.byref(v),
                                        pa, pb, m, _lib.ACCUMULATE if accumulate else 0))
    return Y


def axis_colifilt(X, ha, hb, axis=0, out=None, accumulate=False, pad=(0, 0), crop=(0, 0)):
    """Device colifilt along *axis* (dtcwt/numpy/lowlevel.py:156-260)."""
    axis = axis % X.ndim
    keep, pa, pb, m = _pair_args(ha, hb)
    L = X.shape[axis] + pad[0] + pad[1]
    if L % 2 != 0:
        raise ValueError('No. of rows in X must be a multiple of 2')
    nwrite = 2 * L - crop[0] - crop[1]
    Y = _out(X, axis, nwrite, out, accumulate)
    v = _view(X, axis, nwrite, pad, crop)
    check(_lib.lib().dtcwt_hip_colifilt(X.ctx.handle, dtype_code(X.dtype), X.ptr, Y.ptr, ctypes.byref(v),
                                        pa, pb, m, _lib.ACCUMULATE if accumulate else 0))
    return Y


# ---- fused pairs (one pass over the data instead of two) -------------------------------
def axis_colfilter2(X, h0, h1, axis=0, pad=(0, 0), crop=(0, 0)):
    """(colfilter(X, h0), colfilter(X, h1)) along *axis* in one kernel; falls back to two
    calls when the filter lengths differ in parity (different output lengths)."""
    axis = axis % X.ndim
    k0, p0, m0 = taps_arg(h0)
    k1, p1, m1 = taps_arg(h1)
    if (m0 & 1) != (m1 & 1):
        return (axis_colfilter(X, h0, axis=axis, pad=pad, crop=crop),
                axis_colfilter(X, h1, axis=axis, pad=pad, crop=crop))
    L = X.shape[axis] + pad[0] + pad[1]
    nwrite = (L if m0 % 2 else L + 1) - crop[0] - crop[1]
    Y0, Y1 = _out(X, axis, nwrite, None, False), _out(X, axis, nwrite, None, False)
    v = _view(X, axis, nwrite, pad, crop)
    check(_lib.lib().dtcwt_hip_colfilter2(X.ctx.handle, dtype_code(X.dtype), X.ptr, Y0.ptr, Y1.ptr,
                                          ctypes.byref(v), p0, m0, p1, m1))
    return Y0, Y1


def axis_colfilter_sum2(X0, X1, h0, h1, axis=0, crop=(0, 0)):
    """colfilter(X0, h0) + colfilter(X1, h1) along *axis* in one kernel."""
    axis = axis % X0.ndim
    k0, p0, m0 = taps_arg(h0)
    k1, p1, m1 = taps_arg(h1)
    if X0.shape != X1.shape or X0.dtype != X1.dtype:
        raise ValueError('operands of a fused sum must have equal shape and dtype')
    if (m0 & 1) != (m1 & 1):
        Y = axis_colfilter(X0, h0, axis=axis, crop=crop)
        return axis_colfilter(X1, h1, axis=axis, crop=crop, out=Y, accumulate=True)
    L = X0.shape[axis]
    nwrite = (L if m0 % 2 else L + 1) - crop[0] - crop[1]
    Y = _out(X0, axis, nwrite, None, False)
    v = _view(X0, axis, nwrite, (0, 0), crop)
    check(_lib.lib().dtcwt_hip_colfilter_sum2(X0.ctx.handle, dtype_code(X0.dtype), X0.ptr, X1.ptr, Y.ptr,
                                              ctypes.byref(v), p0, m0, p1, m1))
    return Y


def axis_coldfilt2(X, pair0, pair1, axis=0, pad=(0, 0), crop=(0, 0)):
    """(coldfilt(X, *pair0), coldfilt(X, *pair1)) along *axis* in one kernel."""
    axis = axis % X.ndim
    ka, pa0, pb0, m = _pair_args(*pair0)
    kb, pa1, pb1, m1 = _pair_args(*pair1)
    if m != m1:             # pairs of different lengths: two single launches
        return (axis_coldfilt(X, pair0[0], pair0[1], axis=axis, pad=pad, crop=crop),
                axis_coldfilt(X, pair1[0], pair1[1], axis=axis, pad=pad, crop=crop))
    L = X.shape[axis] + pad[0] + pad[1]
    if L % 4 != 0:
        raise ValueError('No. of rows in X must be a multiple of 4')
    nwrite = L // 2 - crop[0] - crop[1]
    Y0, Y1 = _out(X, axis, nwrite, None, False), _out(X, axis, nwrite, None, False)
    v = _view(X, axis, nwrite, pad, crop)
    check(_lib.lib().dtcwt_hip_coldfilt2(X.ctx.handle, dtype_code(X.dtype), X.ptr, Y0.ptr, Y1.ptr,
                                         ctypes.byref(v), pa0, pb0, pa1, pb1, m))
    return Y0, Y1


def axis_colifilt_sum2(X0, X1, pair0, pair1, axis=0, crop=(0, 0)):
    """colifilt(X0, *pair0) + colifilt(X1, *pair1) along *axis* in one kernel."""
    axis = axis % X0.ndim
    ka, pa0, pb0, m = _pair_args(*pair0)
    kb, pa1, pb1, m1 = _pair_args(*pair1)
    if X0.shape != X1.shape or X0.dtype != X1.dtype:
        raise ValueError('operands of a fused sum must have equal shape and dtype')
    if m != m1:             # pairs of different lengths: two single launches
        Y = axis_colifilt(X0, pair0[0], pair0[1], axis=axis, crop=crop)
        return axis_colifilt(X1, pair1[0], pair1[1], axis=axis, crop=crop, out=Y, accumulate=True)
    L = X0.shape[axis]
    if L % 2 != 0:
        raise ValueError('No. of rows in X must be a multiple of 2')
    nwrite = 2 * L - crop[0] - crop[1]
    Y = _out(X0, axis, nwrite, None, False)
    v = _view(X0, axis, nwrite, (0, 0), crop)
    check(_lib.lib().dtcwt_hip_colifilt_sum2(X0.ctx.handle, dtype_code(X0.dtype), X0.ptr, X1.ptr, Y.ptr,
                                             ctypes.byref(v), pa0, pb0, pa1, pb1, m))
    return Y


def q2c(y, Yh, slot0, slot1):
    """Device q2c of plane(s) y [..., R, C] into subbands slot0/slot1 of Yh [..., R/2,
    C/2, 6] (dtcwt/numpy/transform2d.py:301-322 + the slice-assign of :122-127)."""
    R, C = y.shape[-2:]
    batch = _prod(y.shape[:-2])
    check(_lib.lib().dtcwt_hip_q2c(y.ctx.handle, dtype_code(y.dtype), y.ptr, batch, R, C, R * C, C,
                                   Yh.ptr, slot0, slot1))
    return Yh


def c2q(Yh, slot0, slot1, gain0, gain1, out=None):
    """Device c2q of subbands slot0/slot1 of Yh [..., R, C, 6] -> real plane [..., 2R, 2C]
    (dtcwt/numpy/transform2d.py:324-350)."""
    R, C = Yh.shape[-3:-1]
    batch = _prod(Yh.shape[:-3])
    rdt = np.float32 if Yh.dtype == np.complex64 else np.float64
    if out is None:
        out = DeviceArray(Yh.ctx, Yh.shape[:-3] + (2 * R, 2 * C), rdt)
    check(_lib.lib().dtcwt_hip_c2q(Yh.ctx.handle, dtype_code(Yh.dtype), Yh.ptr, batch, R, C, slot0, slot1,
                                   float(gain0), float(gain1), out.ptr, 4 * R * C, 2 * C))
    return out


def _check_na(rc):
    """-3 = "not applicable, use the filter-by-filter kernels"; anything else non-zero raises."""
    if rc == -3:
        return False
    check(rc)
    return True


def level2d_forward(X, kind, pad_r, pad_c, lo, hi):
    """One whole forward level of the 2-D transform on X [B, R, C] in one launch
    (dtcwt_hip_level2d_forward): kind 0 = level 1 with the odd-length pair lo = h0o, hi = h1o
    (dtcwt/numpy/transform2d.py:112-130); kind 1 = a level >= 2 with the q-shift pairs
    lo = (h0b, h0a), hi = (h1b, h1a) in coldfilt's argument order (:132-160).  Returns
    (LoLo, Yh) or None when the library has no such kernel for these filters / sizes."""
    B, R, C = X.shape
    LR, LC = R + pad_r[0] + pad_r[1], C + pad_c[0] + pad_c[1]
    if kind == 0:
        (k0, pa0, m0), (k1, pa1, m1) = taps_arg(lo), taps_arg(hi)
        pb0 = pb1 = None
        R1, C1 = LR, LC
    else:
        k0, pa0, pb0, m0 = _pair_args(*lo)
        k1, pa1, pb1, m1 = _pair_args(*hi)
        R1, C1 = LR // 2, LC // 2
    if R1 % 2 or C1 % 2 or (kind == 1 and (LR % 4 or LC % 4)):
        return None
    cdt = np.complex64 if X.dtype == np.float32 else np.complex128
    Lo = DeviceArray(X.ctx, (B, R1, C), X.dtype)
    Hi = DeviceArray(X.ctx, (B, R1, C), X.dtype)
    LoLo = DeviceArray(X.ctx, (B, R1, C1), X.dtype)
    Yh = DeviceArray(X.ctx, (B, R1 // 2, C1 // 2, 6), cdt)
    rc = _lib.lib().dtcwt_hip_level2d_forward(
        X.ctx.handle, dtype_code(X.dtype), kind, X.ptr, B, R, C, int(pad_r[0]), int(pad_r[1]),
        int(pad_c[0]), int(pad_c[1]), pa0, pb0, pa1, pb1, m0, m1, Lo.ptr, Hi.ptr, LoLo.ptr, Yh.ptr)
    return (LoLo, Yh) if _check_na(rc) else None


def level2d_inverse(Zl, Yh, kind, gains, crop_r, crop_c, lo, hi):
    """One whole inverse level in one launch (dtcwt_hip_level2d_inverse): Zl [B, Rl, Cl]
    lowpass, Yh [B, Rl/2, Cl/2, 6] subbands with per-subband *gains*; kind 0 = level 1 with
    lo = g0o, hi = g1o (dtcwt/numpy/transform2d.py:275-293), kind 1 = a level >= 2 with
    lo = (g0b, g0a), hi = (g1b, g1a) in colifilt's argument order and crop_r / crop_c output
    samples dropped from both ends (:242-273).  Returns Z or None (no kernel for this case)."""
    B, Rl, Cl = Zl.shape
    if Yh.shape != (B, Rl // 2, Cl // 2, 6) or Rl % 2 or Cl % 2:
        return None
    if kind == 0:
        (k0, pa0, m0), (k1, pa1, m1) = taps_arg(lo), taps_arg(hi)
        pb0 = pb1 = None
        Rz, Cz = Rl, Cl
    else:
        k0, pa0, pb0, m0 = _pair_args(*lo)
        k1, pa1, pb1, m1 = _pair_args(*hi)
        Rz, Cz = 2 * Rl - 2 * crop_r, 2 * Cl - 2 * crop_c
    if Rz < 1 or Cz < 1:
        return None
    g = (ctypes.c_double * 6)(*[float(v) for v in gains])
    Y1 = DeviceArray(Zl.ctx, (B, Rz, Cl), Zl.dtype)
    Y2 = DeviceArray(Zl.ctx, (B, Rz, Cl), Zl.dtype)
    Z = DeviceArray(Zl.ctx, (B, Rz, Cz), Zl.dtype)
    rc = _lib.lib().dtcwt_hip_level2d_inverse(
        Zl.ctx.handle, dtype_code(Zl.dtype), kind, Zl.ptr, Yh.ptr, B, Rl, Cl, g, int(crop_r), int(crop_c),
        pa0, pb0, pa1, pb1, m0, m1, Y1.ptr, Y2.ptr, Z.ptr)
    return Z if _check_na(rc) else None


def level1d_forward(X, kind, pad, lo, hi):
    """One forward level of the 1-D transform on X [n, k] in one launch, highpass packing
    included (dtcwt_hip_level1d_forward; dtcwt/numpy/transform1d.py:79-100): kind 0 with
    lo = h0o, hi = h1o, kind 1 with lo = (h0b, h0a), hi = (h1b, h1a).  Returns (Lo, Yh) or None
    when the library has no such kernel for these filters / sizes."""
    n, k = X.shape
    L = n + pad[0] + pad[1]
    if kind == 0:
        (k0, pa0, m0), (k1, pa1, m1) = taps_arg(lo), taps_arg(hi)
        pb0 = pb1 = None
        n1 = L
    else:
        k0, pa0, pb0, m0 = _pair_args(*lo)
        k1, pa1, pb1, m1 = _pair_args(*hi)
        n1 = L // 2
    if n1 % 2 or (kind == 1 and L % 4) or not (k == 1 or k >= 32):
        return None
    cdt = np.complex64 if X.dtype == np.float32 else np.complex128
    Lo = DeviceArray(X.ctx, (n1, k), X.dtype)
    Yh = DeviceArray(X.ctx, (n1 // 2, k), cdt)
    rc = _lib.lib().dtcwt_hip_level1d_forward(X.ctx.handle, dtype_code(X.dtype), kind, X.ptr, n, k, int(pad[0]),
                                              int(pad[1]), pa0, pb0, pa1, pb1, m0, m1, Lo.ptr, Yh.ptr)
    return (Lo, Yh) if _check_na(rc) else None


def level1d_inverse(Lo, Yh, kind, gain, crop, lo, hi):
    """One inverse level of the 1-D transform in one launch (dtcwt_hip_level1d_inverse;
    dtcwt/numpy/transform1d.py:150-176): Lo [n, k], Yh [n/2, k] complex scaled by *gain*; kind 0
    with lo = g0o, hi = g1o, kind 1 with lo = (g0b, g0a), hi = (g1b, g1a) and *crop* samples
    dropped from both ends.  Returns Z or None."""
    n, k = Lo.shape
    if Yh.shape != (n // 2, k) or n % 2 or not (k == 1 or k >= 32):
        return None
    if kind == 0:
        (k0, pa0, m0), (k1, pa1, m1) = taps_arg(lo), taps_arg(hi)
        pb0 = pb1 = None
        nz = n
    else:
        k0, pa0, pb0, m0 = _pair_args(*lo)
        k1, pa1, pb1, m1 = _pair_args(*hi)
        nz = 2 * n - 2 * crop
    if nz < 1:
        return None
    Z = DeviceArray(Lo.ctx, (nz, k), Lo.dtype)
    rc = _lib.lib().dtcwt_hip_level1d_inverse(Lo.ctx.handle, dtype_code(Lo.dtype), kind, Lo.ptr, Yh.ptr, n, k,
                                              float(gain), int(crop), pa0, pb0, pa1, pb1, m0, m1, Z.ptr)
    return Z if _check_na(rc) else None


# ------------------------------------------------------------------ public functions
def _to_dev(X, ctx):
    """-> (DeviceArray 2-D, was_host)"""
    if isinstance(X, DeviceArray):
        if X.ndim != 2:
            raise ValueError('device input must be two-dimensional')
        return X, False
    X = asfarray(X)
    if X.ndim != 2:
        raise ValueError('X must be a two-dimensional array')
    ctx = ctx or _lib.default_context()
    return ctx.to_device(X), True


def colfilter(X, h, ctx=None):
    """Filter the columns of image *X* with *h*, no decimation, symmetric extension.
    Output has the rows of X (odd-length h) or one more (even-length h)."""
    Xd, host = _to_dev(X, ctx)
    Y = axis_colfilter(Xd, h, axis=0)
    return Y.get() if host else Y


def coldfilt(X, ha, hb, ctx=None):
    """Dual-tree decimating column filter: rows % 4 == 0, even-length ha/hb of equal
    shape (``ValueError`` otherwise); output has half the rows."""
    if not isinstance(X, DeviceArray):
        X = asfarray(X)
    r = X.shape[0]
    if r % 4 != 0:
        raise ValueError('No. of rows in X must be a multiple of 4')
    if np.shape(ha) != np.shape(hb):
        raise ValueError('Shapes of ha and hb must be the same')
    if np.asarray(ha).reshape(-1).shape[0] % 2 != 0:
        raise ValueError('Lengths of ha and hb must be even')
    Xd, host = _to_dev(X, ctx)
    Y = axis_coldfilt(Xd, ha, hb, axis=0)
    return Y.get() if host else Y


def colifilt(X, ha, hb, ctx=None):
    """Dual-tree interpolating column filter: rows % 2 == 0, even-length ha/hb of equal
    shape (``ValueError`` otherwise); output has twice the rows."""
    if not isinstance(X, DeviceArray):
        X = asfarray(X)
    r = X.shape[0]
    if r % 2 != 0:
        raise ValueError('No. of rows in X must be a multiple of 2')
    if np.shape(ha) != np.shape(hb):
        raise ValueError('Shapes of ha and hb must be the same')
    if np.asarray(ha).reshape(-1).shape[0] % 2 != 0:
        raise ValueError('Lengths of ha and hb must be even')
    Xd, host = _to_dev(X, ctx)
    Y = axis_colifilt(Xd, ha, hb, axis=0)
    return Y.get() if host else Y
