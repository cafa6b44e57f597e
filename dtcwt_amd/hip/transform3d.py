"""3-D DT-CWT on the device: ``Transform3d`` of the ``hip`` backend.

Interface, shapes, dtypes and exceptions of dtcwt/numpy/transform3d.py:14-619.  The
reference filters slice by slice inside Python loops; here each level filters the whole
volume along axis 2, then 1, then 0 with the generic device filters (the axis-wise order
SURVEY.md section 3.3 shows to be identical) and packs the seven highpass octants with
the ``cube2c`` kernel into the reference's ``[d0/2, d1/2, d2/2, 28]`` complex layout.
Edge padding (``ext_mode`` 4: one plane, 8: two planes per side, :322-335) and the
inverse's cropping (:505-524) are index arithmetic inside the filters, never copies.

float32 volumes whose every level has a fused level kernel run as ONE native call per
transform (``dtcwt_hip_plan3d_*``: the plan owns the level geometry and the lowpass workspaces and
sequences the level kernels in the library); everything else is sequenced here, level by level.

One deliberate difference: the reference's highpass-free level-1 inverse
(``_level1_ifm_no_highpass``, :442-458) forgets a transpose, so it swaps axes 0 and 2 of
cubic volumes and raises on non-cubic ones; this backend returns the intended result unless the
transform is built with ``reference_quirks=True``, which reproduces the reference literally
(exchanged axes for cubic volumes, ``ValueError`` for the others).
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

__all__ = ['Transform3d']

# (hi on axis 0, hi on axis 1, hi on axis 2) of the highpass octants in the order the
# reference concatenates them (transform3d.py:278-289, :372-383)
_OCTANTS = ((0, 1, 0), (1, 0, 0), (1, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 1))


def _cube2c(vol, sub, Yh, octant):
    """cube2c of the leading ``sub`` block of device volume *vol* into Yh[..., 4o:4o+4]."""
    d0, d1, d2 = sub
    s0, s1 = vol.shape[1] * vol.shape[2], vol.shape[2]
    check(_lib.lib().dtcwt_hip_cube2c(vol.ctx.handle, dtype_code(vol.dtype), vol.ptr, d0, d1, d2, s0, s1,
                                      Yh.ptr, octant))


def _fused_level1(Xd, h0o, h1o):
    """Level 1 in one launch (dtcwt_hip_fwd3_level1) -> (LLL, Yh), or None when these taps /
    this volume only have the generic axis passes."""
    if Xd.dtype != np.float32:
        return None
    h0, h1 = flat_taps(h0o), flat_taps(h1o)
    if h0.shape[0] % 2 == 0 or h1.shape[0] % 2 == 0:
        return None
    n0, n1, n2 = Xd.shape
    LLL = DeviceArray(Xd.ctx, (n0, n1, n2), np.float32)
    Yh = DeviceArray(Xd.ctx, (n0 // 2, n1 // 2, n2 // 2, 28), np.complex64)
    pd = ctypes.POINTER(ctypes.c_double)
    rc = _lib.lib().dtcwt_hip_fwd3_level1(Xd.ctx.handle, Xd.ptr, n0, n1, n2, h0.ctypes.data_as(pd), h0.shape[0],
                                          h1.ctypes.data_as(pd), h1.shape[0], LLL.ptr, Yh.ptr)
    if rc == -3:
        return None
    check(rc)
    return LLL, Yh


def _fused_level2(Xd, pads, h0b, h0a, h1b, h1a):
    """One level >= 2 in two launches (dtcwt_hip_fwd3_level2) -> (LLL, Yh), or None."""
    if Xd.dtype != np.float32:
        return None
    taps = [flat_taps(h) for h in (h0b, h0a, h1b, h1a)]
    m = taps[0].shape[0]
    if any(t.shape[0] != m for t in taps) or m % 2:
        return None
    n = Xd.shape
    L = [n[a] + 2 * pads[a][0] for a in range(3)]
    LLL = DeviceArray(Xd.ctx, tuple(l // 2 for l in L), np.float32)
    Yh = DeviceArray(Xd.ctx, tuple(l // 4 for l in L) + (28,), np.complex64)
    pd = ctypes.POINTER(ctypes.c_double)
    rc = _lib.lib().dtcwt_hip_fwd3_level2(Xd.ctx.handle, Xd.ptr, n[0], n[1], n[2], pads[0][0], pads[1][0],
                                          pads[2][0], *[t.ctypes.data_as(pd) for t in taps], m, LLL.ptr, Yh.ptr)
    if rc == -3:
        return None
    check(rc)
    return LLL, Yh


def _fused_inverse_level1(Yl, Yh, g0o, g1o):
    """Level-1 merge in two launches (dtcwt_hip_inv3_level1) -> Z, or None."""
    if Yl.dtype != np.float32 or Yh.dtype != np.complex64:
        return None
    g0, g1 = flat_taps(g0o), flat_taps(g1o)
    if g0.shape[0] % 2 == 0 or g1.shape[0] % 2 == 0:
        return None
    n0, n1, n2 = Yl.shape
    Z = DeviceArray(Yl.ctx, (n0, n1, n2), np.float32)
    pd = ctypes.POINTER(ctypes.c_double)
    rc = _lib.lib().dtcwt_hip_inv3_level1(Yl.ctx.handle, Yl.ptr, Yh.ptr, n0, n1, n2, g0.ctypes.data_as(pd),
                                          g0.shape[0], g1.ctypes.data_as(pd), g1.shape[0], Z.ptr)
    if rc == -3:
        return None
    check(rc)
    return Z


def _fused_inverse_level2(Yl, Yh, crops, g0b, g0a, g1b, g1a):
    """One level >= 2 merge in two launches (dtcwt_hip_inv3_level2) -> Z, or None."""
    if Yl.dtype != np.float32 or Yh.dtype != np.complex64:
        return None
    taps = [flat_taps(h) for h in (g0b, g0a, g1b, g1a)]
    m = taps[0].shape[0]
    if any(t.shape[0] != m for t in taps) or m % 2:
        return None
    n = Yl.shape
    Z = DeviceArray(Yl.ctx, tuple(2 * n[a] - 2 * crops[a][0] for a in range(3)), np.float32)
    pd = ctypes.POINTER(ctypes.c_double)
    rc = _lib.lib().dtcwt_hip_inv3_level2(Yl.ctx.handle, Yl.ptr, Yh.ptr, n[0], n[1], n[2], crops[0][0], crops[1][0],
                                          crops[2][0], *[t.ctypes.data_as(pd) for t in taps], m, Z.ptr)
    if rc == -3:
        return None
    check(rc)
    return Z


class _Plan3d(object):
    """RAII wrapper of dtcwt_hip_plan3d: the whole 3-D transform in one native call."""

    def __init__(self, ctx, shape, nlevels, ext_mode, biort, qshift):
        L = _lib.lib()
        pd = ctypes.POINTER(ctypes.c_double)
        self.ctx = ctx
        self._keep = [flat_taps(h) for h in biort[:4]] + [flat_taps(h) for h in qshift[:8]]
        bp = (pd * 4)(*[a.ctypes.data_as(pd) for a in self._keep[:4]])
        bl = (ctypes.c_int * 4)(*[a.shape[0] for a in self._keep[:4]])
        qp = (pd * 8)(*[a.ctypes.data_as(pd) for a in self._keep[4:]])
        ql = (ctypes.c_int * 8)(*[a.shape[0] for a in self._keep[4:]])
        h = ctypes.c_void_p()
        rc = L.dtcwt_hip_plan3d_create(ctx.handle, shape[0], shape[1], shape[2], nlevels, ext_mode, bp, bl, qp, ql,
                                       ctypes.byref(h))
        if rc == -3:
            raise NotImplementedError(L.dtcwt_hip_last_error().decode())
        check(rc)
        self._h, self._lib, self.nlevels = h, L, nlevels
        s = (ctypes.c_int64 * (3 + 6 * nlevels))()
        check(L.dtcwt_hip_plan3d_shapes(h, s))
        self.low = tuple(s[0:3])
        self.high = [tuple(s[3 + 6 * l:6 + 6 * l]) for l in range(nlevels)]
        self.scale = [tuple(s[6 + 6 * l:9 + 6 * l]) for l in range(nlevels)]

    def forward(self, Xd, include_scale, discard_level_1):
        """-> (Yl, [Yh], [Ys] | None) or None when a level turns out to have no fused kernel (tiny levels)."""
        ctx, nl = self.ctx, self.nlevels
        Yl = DeviceArray(ctx, self.low, np.float32)
        Yh = [None if (l == 0 and discard_level_1) else DeviceArray(ctx, self.high[l] + (28,), np.complex64)
              for l in range(nl)]
        Ys = [DeviceArray(ctx, self.scale[l], np.float32) for l in range(nl)] if include_scale else None
        vp = ctypes.c_void_p
        yh_p = (vp * nl)(*[(a.ptr if a is not None else None) for a in Yh])
        ys_p = (vp * nl)(*[a.ptr for a in Ys]) if Ys else None
        rc = self._lib.dtcwt_hip_plan3d_forward(self._h, Xd.ptr, Yl.ptr, yh_p, ys_p, 1 if discard_level_1 else 0)
        if rc == -3:
            return None
        check(rc)
        return Yl, Yh, Ys

    def inverse(self, Yl, Yh, out_shape):
        nl = self.nlevels
        Z = DeviceArray(self.ctx, out_shape, np.float32)
        vp = ctypes.c_void_p
        yh_p = (vp * nl)(*[(a.ptr if a is not None else None) for a in Yh])
        rc = self._lib.dtcwt_hip_plan3d_inverse(self._h, Yl.ptr, yh_p, Z.ptr, 0)
        if rc == -3:
            return None
        check(rc)
        return Z

    def __del__(self):
        try:
            if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
                self._lib.dtcwt_hip_plan3d_destroy(self._h)
            self._h = None
        except Exception:
            pass


def _c2cube(Yh, octant):
    e0, e1, e2 = Yh.shape[:3]
    rdt = np.float32 if Yh.dtype == np.complex64 else np.float64
    out = DeviceArray(Yh.ctx, (2 * e0, 2 * e1, 2 * e2), rdt)
    check(_lib.lib().dtcwt_hip_c2cube(Yh.ctx.handle, dtype_code(Yh.dtype), Yh.ptr, e0, e1, e2, octant, out.ptr,
                                      4 * e1 * e2, 2 * e2))
    return out


class Transform3d(object):
    """An implementation of the 3D DT-CWT on AMD GPUs via HIP.  *biort*/*qshift* as for
    :class:`Transform2d`; *ext_mode* 4 or 8 as in dtcwt/numpy/transform3d.py:22-35,:86-99."""

    def __init__(self, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, ext_mode=4, ctx=None, reference_quirks=None):
        # None: taken from the environment, so that users of dtcwt.push_backend('hip') -- who never see this
        # constructor -- can ask for the reference's literal behaviour (DTCWT_HIP_REFERENCE_QUIRKS=1)
        if reference_quirks is None:
            reference_quirks = os.environ.get('DTCWT_HIP_REFERENCE_QUIRKS', '0') not in ('', '0')
        self.reference_quirks = bool(reference_quirks)
        self._plans = collections.OrderedDict()
        try:
            self.biort = _biort(biort)
        except TypeError:
            self.biort = biort
        try:
            self.qshift = _qshift(qshift)
        except TypeError:
            self.qshift = qshift
        self.ext_mode = ext_mode
        self._ctx = ctx
        self.fused = os.environ.get('DTCWT_HIP_FUSED3D', '1') != '0'   # False: generic axis passes only

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    def _taps(self):
        if len(self.biort) not in (4, 6):
            raise ValueError('Biort wavelet must have 6 or 4 components.')
        if len(self.qshift) not in (8, 12):
            raise ValueError('Qshift wavelet must have 12 or 8 components.')
        if self.ext_mode != 4 and self.ext_mode != 8:
            raise ValueError('ext_mode must be one of 4 or 8')
        return self.biort[:4], self.qshift[:8]

    MAX_PLANS = 8

    def _plan(self, shape, nlevels):
        """The native whole-transform plan for float32 volumes of *shape*, or None (even-length biort,
        q-shift lengths without a fused kernel, fused kernels switched off ...)."""
        if not self.fused or nlevels < 1:
            return None
        key = (tuple(shape), nlevels, self.ext_mode)
        if key not in self._plans:
            try:
                self._plans[key] = _Plan3d(self.ctx, shape, nlevels, self.ext_mode, self.biort, self.qshift)
            except NotImplementedError:         # -3: permanent for this key
                self._plans[key] = None
            except _lib.HipError:               # transient (e.g. out of memory): level-by-level now, retry next time
                return None
            while len(self._plans) > self.MAX_PLANS:
                self._plans.popitem(last=False)
        else:
            self._plans.move_to_end(key)
        return self._plans[key]

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _split(V, fn2, lo, hi, pads, axes=(2, 1, 0), parts=None):
        """volume -> {(a0, a1, a2): octant}; axis order 2, 1, 0.  *fn2* filters one volume
        with the lo and the hi filter (pair) in a single pass."""
        parts = {(): V} if parts is None else parts
        for axis in axes:
            nxt = {}
            for key, vol in parts.items():
                nxt[(0,) + key], nxt[(1,) + key] = fn2(vol, lo, hi, axis=axis, pad=pads[axis])
            parts = nxt
        return parts

    def _level1_axis_passes(self, V, h0o, h1o, cdt):
        """Level 1 as axis passes (transform3d.py:256-289): axes 2 and 1 with the pair filters,
        then the axis-0 pass of each of the four volumes with cube2c fused into it
        (dtcwt_hip_fwd3_axis0_cube2c).  -> (LLL, Yh)"""
        nopad = ((0, 0),) * 3
        quarter = self._split(V, ll.axis_colfilter2, h0o, h1o, nopad, axes=(2, 1))
        n0, n1, n2 = V.shape
        h0, h1 = flat_taps(h0o), flat_taps(h1o)
        pd = ctypes.POINTER(ctypes.c_double)
        if h0.shape[0] % 2 and h1.shape[0] % 2 and not (n0 % 2 or n1 % 2 or n2 % 2):
            LLL = DeviceArray(V.ctx, V.shape, V.dtype)
            Yh = DeviceArray(V.ctx, (n0 // 2, n1 // 2, n2 // 2, 28), cdt)
            done = True
            for (a1, a2), vol in quarter.items():
                o_lo = -1 if (a1, a2) == (0, 0) else _OCTANTS.index((0, a1, a2))
                rc = _lib.lib().dtcwt_hip_fwd3_axis0_cube2c(
                    V.ctx.handle, dtype_code(V.dtype), vol.ptr, n0, n1, n2, h0.ctypes.data_as(pd), h0.shape[0],
                    h1.ctypes.data_as(pd), h1.shape[0], o_lo, _OCTANTS.index((1, a1, a2)), LLL.ptr, Yh.ptr)
                if rc == -3:
                    done = False
                    break
                check(rc)
            if done:
                return LLL, Yh
        # even-length taps (octants are (N+1)^3, packed from [:N]) or a volume shorter than the filters
        parts = self._split(V, ll.axis_colfilter2, h0o, h1o, nopad, axes=(0,), parts=quarter)
        return parts[(0, 0, 0)], self._pack(parts, V.shape, cdt)

    def _pack(self, parts, sub, cdt):
        Yh = DeviceArray(parts[(0, 0, 0)].ctx, (sub[0] // 2, sub[1] // 2, sub[2] // 2, 28), cdt)
        for n, o in enumerate(_OCTANTS):
            _cube2c(parts[o], sub, Yh, n)
        return Yh

    def forward(self, X, nlevels=3, include_scale=False, discard_level_1=False):
        """Perform a *n*-level DTCWT-3D decomposition on a 3D matrix *X*; each element of
        ``highpasses`` is a 4-D complex array whose last axis has size 28
        (dtcwt/numpy/transform3d.py:37-131)."""
        (h0o, g0o, h1o, g1o), (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b) = self._taps()
        if isinstance(X, DeviceArray):
            Xd = X
            if Xd.ndim != 3:
                raise ValueError('device input must be three-dimensional')
        else:
            X = np.asanyarray(X)
            Xd = self.ctx.to_device_float(np.atleast_3d(asfarray(X) if np.issubdtype(X.dtype, np.complexfloating) else X))
        cdt = np.complex64 if Xd.dtype == np.float32 else np.complex128
        if Xd.dtype == np.float32 and nlevels >= 1:
            mult = 2 if self.ext_mode == 4 else 4
            plan = self._plan(Xd.shape, nlevels) if not any(s % mult for s in Xd.shape) else None
            done = plan.forward(Xd, include_scale, discard_level_1) if plan is not None else None
            if done is not None:            # the whole transform in one native call
                Yl, Yh, Ys = done
                return Pyramid(Yl, tuple(Yh), tuple(Ys)) if include_scale else Pyramid(Yl, tuple(Yh))
        Yl = Xd
        Yh = [None] * nlevels
        Ys = [None] * nlevels
        nopad = ((0, 0),) * 3
        for level in range(nlevels):
            if level == 0:
                mult = 2 if self.ext_mode == 4 else 4
                if any(s % mult for s in Yl.shape):            # :214-217
                    raise ValueError('Input shape should be a multiple of %d in each direction when '
                                     'self.ext_mode == %d' % (mult, self.ext_mode))
                if discard_level_1:                            # :291-315
                    for axis in (2, 1, 0):
                        Yl = ll.axis_colfilter(Yl, h0o, axis=axis)
                else:
                    fused = _fused_level1(Yl, h0o, h1o) if self.fused else None
                    if fused is not None:
                        Yl, Yh[0] = fused
                    else:
                        Yl, Yh[0] = self._level1_axis_passes(Yl, h0o, h1o, cdt)
            else:                                              # :317-383
                mult, npad = (4, 1) if self.ext_mode == 4 else (8, 2)
                pads = tuple((npad, npad) if Yl.shape[a] % mult else (0, 0) for a in range(3))
                fused = _fused_level2(Yl, pads, h0b, h0a, h1b, h1a) if self.fused else None
                if fused is not None:
                    Yl, Yh[level] = fused
                else:
                    parts = self._split(Yl, ll.axis_coldfilt2, (h0b, h0a), (h1b, h1a), pads)
                    Yl = parts[(0, 0, 0)]
                    Yh[level] = self._pack(parts, Yl.shape, cdt)
            Ys[level] = Yl
        if include_scale:
            return Pyramid(Yl, tuple(Yh), tuple(Ys))
        return Pyramid(Yl, tuple(Yh))

    # ------------------------------------------------------------------ inverse
    @staticmethod
    def _merge_axis1_fused(Yl, Yh, g0o, g1o):
        """The axis-1 pass of the level-1 merge with c2cube on load (dtcwt_hip_inv3_axis1_c2cube):
        {(a0, a2): volume}, or None where the library declines (even-length filters, ...)."""
        g0, g1 = flat_taps(g0o), flat_taps(g1o)
        if not (g0.shape[0] % 2 and g1.shape[0] % 2):
            return None
        n0, n1, n2 = Yl.shape
        pd = ctypes.POINTER(ctypes.c_double)
        p1 = {}
        for a0 in (0, 1):
            for a2 in (0, 1):
                out = DeviceArray(Yl.ctx, Yl.shape, Yl.dtype)
                o_lo = -1 if (a0, a2) == (0, 0) else _OCTANTS.index((a0, 0, a2))
                rc = _lib.lib().dtcwt_hip_inv3_axis1_c2cube(
                    Yl.ctx.handle, dtype_code(Yl.dtype), Yl.ptr, Yh.ptr, n0, n1, n2, g0.ctypes.data_as(pd),
                    g0.shape[0], g1.ctypes.data_as(pd), g1.shape[0], o_lo, _OCTANTS.index((a0, 1, a2)), out.ptr)
                if rc == -3:
                    return None
                check(rc)
                p1[(a0, a2)] = out
        return p1

    @staticmethod
    def _merge(Yl, Yh, fsum, lo, hi, crops, p1=None):
        """8 octants -> one volume; axis order 1, 0, 2 (transform3d.py:425-435, :485-495).
        *fsum* computes filter(lo-branch) + filter(hi-branch) in a single pass; *p1* = the
        axis-1 pass already done (:meth:`_merge_axis1_fused`)."""
        if p1 is None:
            parts = {(0, 0, 0): Yl}
            for n, o in enumerate(_OCTANTS):
                parts[o] = _c2cube(Yh, n)
            p1 = {}
            for a0 in (0, 1):
                for a2 in (0, 1):
                    p1[(a0, a2)] = fsum(parts[(a0, 0, a2)], parts[(a0, 1, a2)], lo, hi, axis=1, crop=crops[1])
        p0 = {}
        for a2 in (0, 1):
            p0[a2] = fsum(p1[(0, a2)], p1[(1, a2)], lo, hi, axis=0, crop=crops[0])
        return fsum(p0[0], p0[1], lo, hi, axis=2, crop=crops[2])

    def _inverse_planned(self, Yl, Yh, nlevels):
        """The whole inverse in one native call, or None when the pyramid does not fit a plan."""
        if Yl.dtype != np.float32 or any(y is not None and y.dtype != np.complex64 for y in Yh):
            return None
        if any(y is None for y in Yh[1:]) or (Yh[0] is None and self.reference_quirks):
            return None
        if Yh[0] is not None:
            shape = tuple(2 * s for s in Yh[0].shape[:3])
        elif nlevels >= 2:
            # no level-1 highpasses to read the extents from: the reference then takes level 2 as unpadded
            # (prev_level_size = 2 x the level-2 highpass extents), i.e. the volume is 4 x Yh[1]
            shape = tuple(4 * s for s in Yh[1].shape[:3])
        else:
            shape = tuple(Yl.shape)
        try:
            plan = self._plan(shape, nlevels)
        except Exception:
            plan = None
        if plan is None or plan.low != tuple(Yl.shape):
            return None
        for l in range(nlevels):
            if Yh[l] is not None and tuple(Yh[l].shape) != plan.high[l] + (28,):
                return None
        return plan.inverse(Yl, list(Yh), shape)

    def inverse(self, pyramid, device_output=False):
        """Perform an *n*-level dual-tree complex wavelet (DTCWT) 3D reconstruction
        (dtcwt/numpy/transform3d.py:133-206); ``highpasses[0]`` may be ``None``."""
        (h0o, g0o, h1o, g1o), (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b) = self._taps()
        nlevels = nlevels_of(pyramid)
        if hasattr(pyramid, 'device_parts'):
            probe = pyramid.hip_lowpass if pyramid.hip_lowpass is not None else pyramid.lowpass
            rdt = np.float32 if probe.dtype in (np.float32, np.complex64) else np.float64
            Yl, Yh = pyramid.device_parts(self.ctx, rdt)
        else:
            low = asfarray(pyramid.lowpass)
            rdt = np.float32 if low.dtype == np.float32 else np.float64
            cdt = np.complex64 if rdt == np.float32 else np.complex128
            Yl = self.ctx.to_device(low, dtype=rdt)
            Yh = tuple(None if y is None else self.ctx.to_device(np.asarray(y), dtype=cdt)
                       for y in pyramid.highpasses)
        if nlevels == 0:
            return Yl if device_output else Yl.get()
        Z = self._inverse_planned(Yl, Yh, nlevels)
        if Z is not None:
            return Z if device_output else Z.get()
        nocrop = ((0, 0),) * 3
        for level in range(nlevels):
            cur = Yh[-level - 1]
            if level == nlevels - 1:                           # level 1
                if cur is None:                                # :442-458 (see module docstring)
                    for axis in (1, 0, 2):
                        Yl = ll.axis_colfilter(Yl, g0o, axis=axis)
                    if self.reference_quirks:
                        # the reference filters "rows of each slice" through a transposed view twice and forgets
                        # the third transpose: the result arrives with axes 0 and 2 exchanged, which only fits
                        # the output array of a cubic volume
                        if Yl.shape[0] != Yl.shape[2]:
                            raise ValueError('could not broadcast: the reference cannot invert non-cubic volumes '
                                             'without level-1 highpasses (dtcwt/numpy/transform3d.py:454-456)')
                        Yl = self.ctx.to_device(np.ascontiguousarray(Yl.get().transpose(2, 1, 0)))
                elif flat_taps(g0o).shape[0] % 2 == 0:
                    # even-length taps (:394-398, :437-438): first N samples of the lowpass
                    # block, N -> N+1 per axis, then drop sample 0 of every axis.
                    n0, n1, n2 = (2 * s for s in cur.shape[:3])
                    low = Yl            # leading block on the device: identity filter + crop per axis
                    for axis, n in enumerate((n0, n1, n2)):
                        if low.shape[axis] != n:
                            low = ll.axis_colfilter(low, np.ones(1), axis=axis, crop=(0, low.shape[axis] - n))
                    Yl = self._merge(low, cur, ll.axis_colfilter_sum2, g0o, g1o, ((1, 0),) * 3)
                else:
                    if tuple(Yl.shape) != tuple(2 * s for s in cur.shape[:3]):
                        raise ValueError('Sizes of highpasses are not valid for the 3D inverse')
                    fused = _fused_inverse_level1(Yl, cur, g0o, g1o) if self.fused else None
                    if fused is not None:
                        Yl = fused
                    else:
                        Yl = self._merge(Yl, cur, ll.axis_colfilter_sum2, g0o, g1o, nocrop,
                                         p1=self._merge_axis1_fused(Yl, cur, g0o, g1o))
            else:                                              # :460-526
                if tuple(Yl.shape) != tuple(2 * s for s in cur.shape[:3]):
                    raise ValueError('Sizes of highpasses are not valid for the 3D inverse')
                nxt = Yh[-level - 2]
                prev = tuple(nxt.shape[:3]) if nxt is not None else tuple(2 * s for s in cur.shape[:3])
                c = 1 if self.ext_mode == 4 else 2
                crops = tuple((c, c) if cur.shape[a] * 2 != prev[a] else (0, 0) for a in range(3))
                fused = _fused_inverse_level2(Yl, cur, crops, g0b, g0a, g1b, g1a) if self.fused else None
                Yl = fused if fused is not None else \
                    self._merge(Yl, cur, ll.axis_colifilt_sum2, (g0b, g0a), (g1b, g1a), crops)
        return Yl if device_output else Yl.get()
