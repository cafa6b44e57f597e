"""2-D DT-CWT on MI355X: ``Transform2d`` of the ``hip`` backend.

Same constructor, ``forward`` / ``inverse`` signatures, shapes, dtypes, exceptions and
log messages as dtcwt/numpy/transform2d.py:18-295.  The level loop runs on the device:

* float32 images with the shipped biort / q-shift sets -- the band-pass ``_bp`` ones (6 / 12
  vectors) included -- go through the fused plan (``dtcwt_hip_plan2d_*``): one gfx950 kernel
  per level and direction;
* everything else (float64, unusual tap lengths) goes through the generic device filters
  (``dtcwt_hip_colfilter/coldfilt/colifilt/q2c/c2q``).

There is no host fallback: without the library or a GPU the first use raises
``NoHIPPresentError``.  Batches of equally-sized images are transformed in one go by
``forward_channels`` / ``inverse_channels`` (interface of the reference's TensorFlow
backend, dtcwt/tf/transform2d.py:179-336, :422-588).
"""
import collections
import os
import ctypes
import logging

import numpy as np

from dtcwt_amd.coeffs import biort as _biort, qshift as _qshift
from dtcwt_amd.defaults import DEFAULT_BIORT, DEFAULT_QSHIFT
from dtcwt_amd.utils import asfarray, flat_taps
from dtcwt_amd.hip import _lib
from dtcwt_amd.hip._lib import DeviceArray, check
from dtcwt_amd.hip.common import Pyramid, nlevels_of
from dtcwt_amd.hip import lowlevel as ll

__all__ = ['Transform2d', 'dtwavexfm2', 'dtwaveifm2']

_vp = ctypes.c_void_p
_pd = ctypes.POINTER(ctypes.c_double)


class _Plan2d(object):
    """RAII wrapper of dtcwt_hip_plan2d."""

    def __init__(self, ctx, batch, rows, cols, nlevels, biort, qshift):
        L = _lib.lib()
        self.ctx = ctx
        self._keep = [flat_taps(h) for h in biort[:4]] + [flat_taps(h) for h in qshift[:8]]
        bp = (_pd * 4)(*[a.ctypes.data_as(_pd) for a in self._keep[:4]])
        bl = (ctypes.c_int * 4)(*[a.shape[0] for a in self._keep[:4]])
        qp = (_pd * 8)(*[a.ctypes.data_as(_pd) for a in self._keep[4:]])
        ql = (ctypes.c_int * 8)(*[a.shape[0] for a in self._keep[4:]])
        h = _vp()
        rc = L.dtcwt_hip_plan2d_create(ctx.handle, batch, rows, cols, nlevels, bp, bl, qp, ql, ctypes.byref(h))
        if rc == -3:
            raise NotImplementedError(L.dtcwt_hip_last_error().decode())
        check(rc)
        self._h = h
        self._lib = L
        if len(biort) >= 6 or len(qshift) >= 12:
            # band-pass sets: third filter of the diagonal subbands (transform2d.py:116-129, :145-155)
            b = [flat_taps(h) for h in biort[4:6]] if len(biort) >= 6 else []
            q = [flat_taps(h) for h in qshift[8:12]] if len(qshift) >= 12 else []
            self._keep += b + q
            nb = b[0].shape[0] if b else 0
            nq = q[0].shape[0] if q else 0
            if (b and b[1].shape[0] != nb) or any(v.shape[0] != nq for v in q):
                rc = -3
            else:
                pb = [v.ctypes.data_as(_pd) for v in b] or [None, None]
                pq = [v.ctypes.data_as(_pd) for v in q] or [None] * 4
                rc = L.dtcwt_hip_plan2d_set_bandpass(h, pb[0], pb[1], nb, pq[0], pq[1], pq[2], pq[3], nq)
            if rc:
                self._h = None                      # not ours any more: __del__ must not destroy it again
                L.dtcwt_hip_plan2d_destroy(h)
                if rc == -3:
                    raise NotImplementedError('no fused band-pass kernels for these tap lengths')
                check(rc)
        self.batch, self.rows, self.cols, self.nlevels = batch, rows, cols, nlevels
        s = (ctypes.c_int * (4 + 4 * nlevels))()
        check(L.dtcwt_hip_plan2d_shapes(h, s))
        self.ext = (s[0], s[1])
        self.low = (s[2], s[3])
        self.high = [(s[4 + 4 * l], s[5 + 4 * l]) for l in range(nlevels)]
        self.scale = [(s[6 + 4 * l], s[7 + 4 * l]) for l in range(nlevels)]

    def forward(self, Xd, include_scale):
        ctx, B, nl = self.ctx, self.batch, self.nlevels
        Yl = DeviceArray(ctx, (B,) + self.low, np.float32)
        Yh = [DeviceArray(ctx, (B,) + self.high[l] + (6,), np.complex64) for l in range(nl)]
        Ys = [DeviceArray(ctx, (B,) + self.scale[l], np.float32) for l in range(nl)] if include_scale else None
        yh_p = (_vp * nl)(*[a.ptr for a in Yh])
        ys_p = (_vp * nl)(*[a.ptr for a in Ys]) if Ys else None
        check(self._lib.dtcwt_hip_plan2d_forward(self._h, Xd.ptr, Yl.ptr, yh_p, ys_p))
        return Yl, Yh, Ys

    def forward_into(self, Xd, Yl, Yh, Ys=None):
        """Launch into caller-provided buffers (no allocation: benchmark inner loop)."""
        nl = self.nlevels
        yh_p = (_vp * nl)(*[a.ptr for a in Yh])
        ys_p = (_vp * nl)(*[a.ptr for a in Ys]) if Ys else None
        check(self._lib.dtcwt_hip_plan2d_forward(self._h, Xd.ptr, Yl.ptr, yh_p, ys_p))

    def inverse_into(self, Yl, Yh, gain_mask, Z):
        nl = self.nlevels
        yh_p = (_vp * nl)(*[a.ptr for a in Yh])
        if gain_mask is None:
            gp = None
        else:
            g = np.ascontiguousarray(np.asarray(gain_mask, dtype=np.float64).reshape(6, nl))
            gp = g.ctypes.data_as(_pd)
        check(self._lib.dtcwt_hip_plan2d_inverse(self._h, Yl.ptr, yh_p, gp, Z.ptr))

    def inverse(self, Yl, Yh, gain_mask):
        Z = DeviceArray(self.ctx, (self.batch,) + self.ext, np.float32)
        self.inverse_into(Yl, Yh, gain_mask, Z)
        return Z

    def capture(self, Xd, Yl, Yh, Z=None, gain_mask=None, Ys=None):
        """hipGraph of forward_into (and inverse_into when *Z* is given) on these buffers:
        ``g = plan.capture(X, Yl, Yh, Z); g.launch()`` replays the level loop with one call."""
        nl = self.nlevels
        yh_p = (_vp * nl)(*[a.ptr for a in Yh])
        ys_p = (_vp * nl)(*[a.ptr for a in Ys]) if Ys else None
        gp = None
        if gain_mask is not None:
            gm = np.ascontiguousarray(np.asarray(gain_mask, dtype=np.float64).reshape(6, nl))
            gp = gm.ctypes.data_as(_pd)
        h = _vp()
        check(self._lib.dtcwt_hip_plan2d_capture(self._h, Xd.ptr, Yl.ptr, yh_p, ys_p, gp,
                                                 Z.ptr if Z is not None else None, ctypes.byref(h)))
        return _Graph(self, h, (Xd, Yl, list(Yh), Z, Ys))

    def set_profiling(self, enable=True):
        check(self._lib.dtcwt_hip_plan2d_set_profiling(self._h, 1 if enable else 0))

    def kernel_ms(self):
        """(forward ms per level, inverse ms per level) of the last calls (profiling on)."""
        f = (ctypes.c_float * self.nlevels)()
        i = (ctypes.c_float * self.nlevels)()
        check(self._lib.dtcwt_hip_plan2d_kernel_ms(self._h, f, i))
        return list(f), list(i)

    def set_concurrency(self, transforms_in_flight):
        """Hint: how many independent transforms are kept in flight on this device at a time (other plans on other
        streams included); the marching launches choose their band height by it, and the plan the size from which it uses them
        (:meth:`launches`).  The two programs agree to rounding, not to the bit; :meth:`set_program` pins one."""
        check(self._lib.dtcwt_hip_plan2d_set_concurrency(self._h, int(transforms_in_flight)))

    PROGRAMS = {'auto': -1, 'tiles': 0, 'march': 1}

    def set_program(self, program):
        """``'auto'`` (default): the library picks, per call, between the one-launch marching program for levels 1 + 2
        and the per-level tile programs (by batch x pixels, the concurrency hint, the context's share of the device).
        ``'tiles'`` / ``'march'`` pin one -- the two agree to ~2e-7 relative, NOT to the bit, so pin when a batch and its
        single images (or a forward and the inverse of its pyramid) must be computed by the same arithmetic
        (``dtcwt_hip_plan2d_set_program``).  ``'march'`` still falls back to the tiles where it does not apply."""
        check(self._lib.dtcwt_hip_plan2d_set_program(self._h, self.PROGRAMS[program]))

    def launches(self):
        """(levels 1 + 2 of the forward in one launch?, levels 2 + 1 of the inverse in one launch?)"""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.dtcwt_hip_plan2d_launches(self._h, ctypes.byref(a), ctypes.byref(b)))
        return bool(a.value), bool(b.value)

    def describe(self, scales=False):
        """One line of text: the kernel that runs every level of the forward and of the inverse of this plan (*scales*: a
        forward with ``include_scale``) and the switches the plan was made under (``dtcwt_hip_plan2d_describe``)."""
        buf = ctypes.create_string_buffer(1024)
        check(self._lib.dtcwt_hip_plan2d_describe(self._h, 1 if scales else 0, buf, 1024))
        return buf.value.decode()

    def level1_march(self):
        """(level 1 of the forward as a marching launch of its own?, of the inverse?) -- near_sym_b and antonini, whose
        filters are too long for the fused levels 1 + 2 (``dtcwt_hip_plan2d_level1_march``)."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.dtcwt_hip_plan2d_level1_march(self._h, ctypes.byref(a), ctypes.byref(b)))
        return bool(a.value), bool(b.value)

    def __del__(self):
        # The plan dereferences its context when it is destroyed.  When both die in one garbage
        # cycle (typically at interpreter exit) the finalisers run in arbitrary order: if the
        # context went first, leave the plan's workspaces to the process teardown.
        try:
            if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
                self._lib.dtcwt_hip_plan2d_destroy(self._h)
            self._h = None
        except Exception:
            pass


class _Graph(object):
    """A captured level loop (dtcwt_hip_graph); keeps its plan and buffers alive."""

    def __init__(self, plan, handle, keep):
        self._plan, self._h, self._keep = plan, handle, keep
        self._lib = plan._lib

    def launch(self):
        check(self._lib.dtcwt_hip_graph_launch(self._h))

    def __del__(self):
        try:
            if getattr(self, '_h', None) and getattr(self._plan.ctx, '_h', None):
                self._lib.dtcwt_hip_graph_destroy(self._h)
            self._h = None
        except Exception:
            pass


def _level_geometry(rows, cols, nlevels):
    """Shapes of every level exactly as dtcwt/numpy/transform2d.py:86-160 produces them."""
    R, C = rows + (rows & 1), cols + (cols & 1)
    lv = [dict(inR=rows, inC=cols, padR=rows & 1, padC=cols & 1, LR=R, LC=C, loR=R, loC=C)]
    for _ in range(1, nlevels):
        r, c = lv[-1]['loR'], lv[-1]['loC']
        pr, pc = int(r % 4 != 0), int(c % 4 != 0)
        lv.append(dict(inR=r, inC=c, padR=pr, padC=pc, LR=r + 2 * pr, LC=c + 2 * pc,
                       loR=(r + 2 * pr) // 2, loC=(c + 2 * pc) // 2))
    return (R, C), lv


_MIN_FUSED_DIM = 40     # == DT_MIN_FUSED_DIM (dtcwt_amd/csrc/fused2d_tiles.hpp)


def _fused_levels(rows, cols, nlevels):
    """How many leading levels are large enough for the fused kernels (one-bounce
    reflection needs every level >= 40 samples wide); coarser levels of the same transform
    run through the generic device filters."""
    _, lv = _level_geometry(rows, cols, nlevels)
    k = 0
    for g in lv:
        if min(g['LR'], g['LC']) < _MIN_FUSED_DIM or min(g['loR'], g['loC']) < _MIN_FUSED_DIM // 2:
            break
        k += 1
    return k


_DEGENERATE_DIM = 8
_PLAN_SWITCHES = ('DTCWT_HIP_MARCH', 'DTCWT_HIP_MARCH_BAND', 'DTCWT_HIP_MARCH_PARTS', 'DTCWT_HIP_XCD_ORDER')


def _degenerate(rows, cols, nlevels):
    """A float32 image some level of which is `_DEGENERATE_DIM` samples or fewer along an axis: its filters reflect
    several times and sum the same few samples over and over, and float32 arithmetic leaves up to 1e-6 of a subband's own
    maximum against the float64 oracle (profiles/r04/parity_worst.json: 9.9e-7 for a 2 x 2 image through four levels).
    The coarse tail of such a transform (every level the fused float32 plan does not take) runs in float64 on the device and is
    rounded to float32 once; images too small for any fused level run in float64 whole."""
    if nlevels < 1:
        return False
    _, lv = _level_geometry(rows, cols, nlevels)
    return any(min(g['LR'], g['LC']) <= _DEGENERATE_DIM for g in lv[:nlevels])


class Transform2d(object):
    """An implementation of the 2D DT-CWT on AMD Instinct GPUs via HIP.  *biort* and
    *qshift* are wavelet names (:py:func:`dtcwt_amd.coeffs.biort`/``qshift``) or tuples of
    tap vectors: (h0o, g0o, h1o, g1o[, h2o, g2o]) and (h0a, h0b, g0a, g0b, h1a, h1b, g1a,
    g1b[, h2a, h2b, g2a, g2b]) -- dtcwt/numpy/transform2d.py:18-38.

    *ctx* (optional) is a :class:`dtcwt_amd.hip.Context` (device + stream), the analogue of
    the OpenCL backend's ``queue`` argument (dtcwt/opencl/transform2d.py:108-110).

    *program* (``'auto'`` | ``'tiles'`` | ``'march'``): float32 levels 1 + 2 run either as one marching launch or as one
    tile-program launch per level; ``'auto'`` lets the library choose per call by size (fastest), which means that
    results are reproducible to ~2e-7 relative but NOT bit for bit between a batch and one of its images transformed
    alone, or across context kinds.  Pin ``'tiles'`` (or ``'march'``) where bit-reproducibility across call shapes matters;
    the reference itself is deterministic per call shape only up to its BLAS-free NumPy summation order."""

    def __init__(self, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, ctx=None, program='auto'):
        if program not in _Plan2d.PROGRAMS:
            raise ValueError("program must be 'auto', 'tiles' or 'march'")
        self.program = program
        try:
            self.biort = _biort(biort)
        except TypeError:
            self.biort = biort
        try:
            self.qshift = _qshift(qshift)
        except TypeError:
            self.qshift = qshift
        self._ctx = ctx
        self._plans = collections.OrderedDict()
        # NumPy image in -> the subbands start travelling to the host as soon as their kernels are enqueued
        # (Pyramid.prefetch); DTCWT_HIP_PREFETCH_HOST=0 keeps the purely lazy copies of round 2
        self.prefetch_host = os.environ.get('DTCWT_HIP_PREFETCH_HOST', '1') != '0'

    MAX_PLANS = 8       # fused plans kept per transform object (each owns per-level workspaces in HBM)

    def clear_plans(self):
        """Drop every cached fused plan (and its device workspaces)."""
        self._plans.clear()

    # ------------------------------------------------------------------ helpers
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
        return self.biort, self.qshift

    def _plan(self, batch, rows, cols, nlevels):
        """Fused plan or None when the wavelets have no fused kernels."""
        if len(self.biort) not in (4, 6) or len(self.qshift) not in (8, 12):
            return None
        # a native plan reads its environment switches once, when it is made (include/dtcwt_hip.h: dtcwt_hip_plan2d_describe):
        # a process that changes them (the tests do) gets a new plan, not a stale one
        key = (batch, rows, cols, nlevels) + tuple(os.environ.get(k) for k in _PLAN_SWITCHES)
        if key not in self._plans:
            try:
                self._plans[key] = _Plan2d(self.ctx, batch, rows, cols, nlevels, self.biort, self.qshift)
                if self.program != 'auto':
                    self._plans[key].set_program(self.program)
            except NotImplementedError:
                self._plans[key] = None
            while len(self._plans) > self.MAX_PLANS:         # least recently used first
                self._plans.popitem(last=False)
        else:
            self._plans.move_to_end(key)
        return self._plans[key]

    def plan(self, batch, rows, cols, nlevels):
        """The fused plan for a batch shape (benchmark / advanced use); raises
        NotImplementedError when these wavelets only have the generic path."""
        p = self._plan(batch, rows, cols, nlevels)
        if p is None:
            raise NotImplementedError('no fused plan for these wavelets')
        return p

    # ------------------------------------------------------------------ forward
    def _forward_device(self, Xd, nlevels, include_scale):
        """Xd: DeviceArray [B, r, c] float32/float64.  Returns (Yl, [Yh], [Ys]|None) with
        a leading batch axis."""
        B, r, c = Xd.shape
        if Xd.dtype == np.float32 and _degenerate(r, c, nlevels):
            # Only the coarse TAIL runs in float64: the leading levels that are large enough for the fused float32
            # plan keep it -- a 4096^2 image with ten levels is a fused four... seven-level transform plus a tiny float64 tail,
            # not a float64 transform of the whole image (three convert passes and no fused kernel at all until round 5)
            ctx = Xd.ctx
            k = _fused_levels(r, c, nlevels)
            plan = self._plan(B, r, c, k) if k > 0 else None
            if plan is not None:
                Yl, Yh, Ys = plan.forward(Xd, include_scale)
                Yl, Yh2, Ys2 = self._forward_generic(None, nlevels, include_scale, start_level=k,
                                                     LoLo=ctx.convert(Yl, np.float64), shape=(B, r, c))
                Yh = Yh + [ctx.convert(y, np.complex64) for y in Yh2]
                if include_scale:
                    Ys = Ys + [ctx.convert(y, np.float32) for y in Ys2]
                return ctx.convert(Yl, np.float32), Yh, (Ys if include_scale else None)
            Yl, Yh, Ys = self._forward_generic(ctx.convert(Xd, np.float64), nlevels, include_scale)
            return (ctx.convert(Yl, np.float32), [ctx.convert(y, np.complex64) for y in Yh],
                    [ctx.convert(y, np.float32) for y in Ys] if include_scale else None)
        if Xd.dtype == np.float32:
            k = _fused_levels(r, c, nlevels)
            plan = self._plan(B, r, c, k) if k > 0 else None
            if plan is not None:
                Yl, Yh, Ys = plan.forward(Xd, include_scale)
                if k == nlevels:
                    return Yl, Yh, Ys
                # coarse tail (levels too small for the fused tiles): generic device filters
                Yl, Yh2, Ys2 = self._forward_generic(None, nlevels, include_scale, start_level=k, LoLo=Yl,
                                                     shape=(B, r, c))
                return Yl, Yh + Yh2, (Ys + Ys2) if include_scale else None
        return self._forward_generic(Xd, nlevels, include_scale)

    def _forward_generic(self, Xd, nlevels, include_scale, start_level=0, LoLo=None, shape=None):
        """Levels start_level .. nlevels-1 with the generic device filters.  With
        start_level > 0, *LoLo* is the lowpass of level start_level-1 and *shape* the
        (B, rows, cols) of the original input."""
        biort, qshift = self._taps()
        h0o, g0o, h1o, g1o = biort[:4]
        h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = qshift[:8]
        bp1, bp2 = len(biort) >= 6, len(qshift) >= 12
        B, r, c = Xd.shape if Xd is not None else shape
        src = Xd if Xd is not None else LoLo
        ctx = src.ctx
        cdt = np.complex64 if src.dtype == np.float32 else np.complex128
        (R, C), lv = _level_geometry(r, c, nlevels)
        Yh, Ys = [], []
        if start_level == 0:
            LoLo = self._forward_generic_level1(Xd, lv, cdt, Yh, Ys)
        for level in range(max(start_level, 1), nlevels):        # :132-160
            g = lv[level]
            pr, pc = (g['padR'],) * 2, (g['padC'],) * 2
            # lo and hi filter of one pass in one launch (dtcwt_hip_coldfilt2)
            lo, hi = (h0b, h0a), (h1b, h1a)
            prev = LoLo
            whole = None if bp2 else ll.level2d_forward(prev, 1, pr, pc, lo, hi)
            if whole is not None:       # the whole level in one launch (dtcwt_hip_level2d_forward)
                LoLo, y = whole
                Yh.append(y)
                Ys.append(LoLo)
                continue
            Lo, Hi = ll.axis_coldfilt2(prev, lo, hi, axis=1, pad=pr)
            LoLo, LoHi = ll.axis_coldfilt2(Lo, lo, hi, axis=2, pad=pc)
            y = DeviceArray(ctx, (B, LoLo.shape[1] >> 1, LoLo.shape[2] >> 1, 6), cdt)
            ll.q2c(LoHi, y, 2, 3)
            if bp2:
                ll.q2c(ll.axis_coldfilt(Hi, h0b, h0a, axis=2, pad=pc), y, 0, 5)
                Ba = ll.axis_coldfilt(prev, qshift[9], qshift[8], axis=1, pad=pr)
                ll.q2c(ll.axis_coldfilt(Ba, qshift[9], qshift[8], axis=2, pad=pc), y, 1, 4)
            else:
                HiLo, HiHi = ll.axis_coldfilt2(Hi, lo, hi, axis=2, pad=pc)
                ll.q2c(HiLo, y, 0, 5)
                ll.q2c(HiHi, y, 1, 4)
            Yh.append(y)
            Ys.append(LoLo)
        return LoLo, Yh, (Ys if include_scale else None)

    def _forward_generic_level1(self, Xd, lv, cdt, Yh, Ys):
        biort, qshift = self._taps()
        h0o, g0o, h1o, g1o = biort[:4]
        bp1 = len(biort) >= 6
        B = Xd.shape[0]
        # level 1 (transform2d.py:112-130); odd sizes extended by index math (:86-94)
        pr, pc = (0, lv[0]['padR']), (0, lv[0]['padC'])
        whole = None if bp1 else ll.level2d_forward(Xd, 0, pr, pc, h0o, h1o)
        if whole is not None:           # the whole level in one launch (dtcwt_hip_level2d_forward)
            Yh.append(whole[1])
            Ys.append(whole[0])
            return whole[0]
        Lo, Hi = ll.axis_colfilter2(Xd, h0o, h1o, axis=1, pad=pr)
        LoLo, LoHi = ll.axis_colfilter2(Lo, h0o, h1o, axis=2, pad=pc)
        y = DeviceArray(Xd.ctx, (B, LoLo.shape[1] >> 1, LoLo.shape[2] >> 1, 6), cdt)
        ll.q2c(LoHi, y, 2, 3)
        if bp1:
            ll.q2c(ll.axis_colfilter(Hi, h0o, axis=2, pad=pc), y, 0, 5)
            Ba = ll.axis_colfilter(Xd, biort[4], axis=1, pad=pr)
            ll.q2c(ll.axis_colfilter(Ba, biort[4], axis=2, pad=pc), y, 1, 4)
        else:
            HiLo, HiHi = ll.axis_colfilter2(Hi, h0o, h1o, axis=2, pad=pc)
            ll.q2c(HiLo, y, 0, 5)
            ll.q2c(HiHi, y, 1, 4)
        Yh.append(y)
        Ys.append(LoLo)
        return LoLo

    @staticmethod
    def _log_extension(original, extended):
        """The warnings of dtcwt/numpy/transform2d.py:164-183."""
        er, ec = extended[0] != original[0], extended[1] != original[1]
        if not (er or ec):
            return
        logging.warning('The image entered is now a {0} NOT a {1}.'.format(
            'x'.join(str(s) for s in extended), 'x'.join(str(s) for s in original)))
        if er and ec:
            logging.warning('The bottom row and rightmost column have been duplicated, prior to decomposition.')
        elif er:
            logging.warning('The bottom row has been duplicated, prior to decomposition.')
        else:
            logging.warning('The rightmost column has been duplicated, prior to decomposition.')

    def forward(self, X, nlevels=3, include_scale=False):
        """Perform a *n*-level DTCWT-2D decomposition on a 2D matrix *X*.

        :param X: 2D real array (NumPy-compatible), or a 2-D :class:`DeviceArray` / torch
            GPU tensor already in HBM (cf. dtcwt/opencl/transform2d.py:133-135)
        :param nlevels: Number of levels of wavelet decomposition
        :returns: A :py:class:`Pyramid` whose buffers live on the device.
        """
        self._taps()
        if _lib.is_foreign_device_tensor(X):
            X = DeviceArray.wrap(self.ctx, X)
        if isinstance(X, DeviceArray):
            if X.ndim != 2:
                raise ValueError('Input array must be two-dimensional')
            if X.dtype not in (np.float32, np.float64):
                raise TypeError('device inputs must be float32 or float64, not %s' % X.dtype)
            Xd = X
            from_host = False
        else:
            X = np.atleast_2d(np.asanyarray(X))
            if np.issubdtype(X.dtype, np.complexfloating):
                X = asfarray(X)
            if X.ndim >= 3:
                raise ValueError('The entered image is {0}, which is invalid '.format(
                    'x'.join(str(s) for s in X.shape)) + 'for the 2D transform in a hip backend. ' +
                    'Please enter each image slice separately.')
            Xd = self.ctx.to_device_float(X)         # asfarray semantics; integers are widened on the device
            from_host = self.prefetch_host
        r, c = Xd.shape
        R, C = r + (r & 1), c + (c & 1)
        if nlevels == 0:
            # :98-102 returns the (edge-extended) input
            Xe = Xd.get()
            if R != r:
                Xe = np.vstack((Xe, Xe[[-1], :]))
            if C != c:
                Xe = np.hstack((Xe, Xe[:, [-1]]))
            return Pyramid(Xe, (), ()) if include_scale else Pyramid(Xe, ())
        Yl, Yh, Ys = self._forward_device(Xd.reshape(1, r, c), nlevels, include_scale)
        self._log_extension((r, c), (R, C))
        drop = lambda a: a.reshape(a.shape[1:])
        if include_scale:
            p = Pyramid(drop(Yl), tuple(drop(y) for y in Yh), tuple(drop(s) for s in Ys))
        else:
            p = Pyramid(drop(Yl), tuple(drop(y) for y in Yh))
        # a NumPy image in means NumPy subbands out: start their downloads behind the kernels now (copy stream,
        # page-locked buffers) instead of one blocking copy per subband at first access
        return p.prefetch() if from_host else p

    # ------------------------------------------------------------------ inverse
    @staticmethod
    def _check_shapes(low_shape, high_shapes):
        """Validate pyramid geometry the way the reference's loop would discover it
        (dtcwt/numpy/transform2d.py:263-271); returns per-level crop flags."""
        nl = len(high_shapes)
        crops = [(0, 0)] * nl
        z = tuple(low_shape)
        for level in range(nl - 1, -1, -1):
            h = tuple(high_shapes[level][:2])
            if len(high_shapes[level]) != 3 or high_shapes[level][2] != 6 or z != (2 * h[0], 2 * h[1]):
                raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
            if level >= 1:
                S = (2 * high_shapes[level - 1][0], 2 * high_shapes[level - 1][1])
                z2 = [2 * z[0], 2 * z[1]]
                cr = [0, 0]
                for a in (0, 1):
                    if z2[a] != S[a]:
                        z2[a] -= 2
                        cr[a] = 1
                if tuple(z2) != S:
                    raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
                crops[level] = tuple(cr)
                z = tuple(z2)
        return crops

    def _inverse_device(self, Yl, Yh, gain_mask):
        """Yl [B, r, c], Yh[l] [B, h, w, 6] DeviceArrays -> Z [B, R, C]."""
        nl = len(Yh)
        B = Yl.shape[0]
        crops = self._check_shapes(Yl.shape[1:], [y.shape[1:] for y in Yh])
        R, C = 2 * Yh[0].shape[1], 2 * Yh[0].shape[2]
        if Yl.dtype == np.float32 and _degenerate(R, C, nl):
            ctx = Yl.ctx
            k = _fused_levels(R, C, nl)
            plan = self._plan(B, R, C, k) if k > 0 else None
            if plan is not None and all(plan.high[l] == tuple(Yh[l].shape[1:3]) for l in range(k)):
                # the coarse tail in float64, then the fused float32 plan for the leading levels (see _forward_device)
                Yh64 = [None] * k + [ctx.convert(y, np.complex128) for y in Yh[k:]]
                Z = ctx.convert(self._inverse_generic(ctx.convert(Yl, np.float64), Yh64, gain_mask, crops, stop_level=k), np.float32)
                if plan.low != tuple(Z.shape[1:]):
                    raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
                gm = None if gain_mask is None else np.asarray(gain_mask, dtype=np.float64)[:, :k]
                return plan.inverse(Z, list(Yh[:k]), gm)
            Z = self._inverse_generic(ctx.convert(Yl, np.float64), [ctx.convert(y, np.complex128) for y in Yh], gain_mask, crops)
            return ctx.convert(Z, np.float32)
        if Yl.dtype == np.float32:
            k = _fused_levels(R, C, nl)
            plan = self._plan(B, R, C, k) if k > 0 else None
            if plan is not None:
                if any(plan.high[l] != tuple(Yh[l].shape[1:3]) for l in range(k)):
                    raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
                Z = Yl
                if k < nl:      # coarse tail first, with the generic device filters
                    Z = self._inverse_generic(Yl, Yh, gain_mask, crops, stop_level=k)
                if plan.low != tuple(Z.shape[1:]):
                    raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
                gm = None if gain_mask is None else np.asarray(gain_mask, dtype=np.float64)[:, :k]
                return plan.inverse(Z, list(Yh[:k]), gm)
        return self._inverse_generic(Yl, Yh, gain_mask, crops)

    def _inverse_generic(self, Z, Yh, gain_mask, crops, stop_level=0):
        """Levels len(Yh) .. stop_level+1 with the generic device filters (stop_level = 0:
        the whole inverse)."""
        biort, qshift = self._taps()
        h0o, g0o, h1o, g1o = biort[:4]
        h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = qshift[:8]
        bp1, bp2 = len(biort) >= 6, len(qshift) >= 12
        nl = len(Yh)
        gm = np.ones((6, nl)) if gain_mask is None else np.array(gain_mask, dtype=np.float64)
        level = nl
        while level >= 2 and level > stop_level:             # transform2d.py:242-273
            w, g = Yh[level - 1], gm[:, level - 1]
            cr, cc = crops[level - 1]
            lo, hi = (g0b, g0a), (g1b, g1a)
            whole = None if bp2 else ll.level2d_inverse(Z, w, 1, g, cr, cc, lo, hi)
            if whole is not None:       # the whole level in one launch (dtcwt_hip_level2d_inverse)
                Z = whole
                level -= 1
                continue
            lh = ll.c2q(w, 0, 5, g[0], g[5])
            hl = ll.c2q(w, 2, 3, g[2], g[3])
            hh = ll.c2q(w, 1, 4, g[1], g[4])
            # filter(lo branch) + filter(hi branch) of one pass in one launch (dtcwt_hip_colifilt_sum2)
            lo, hi = (g0b, g0a), (g1b, g1a)
            y1 = ll.axis_colifilt_sum2(Z, lh, lo, hi, axis=1, crop=(cr, cr))
            if bp2:
                y2 = ll.axis_colifilt(hl, g0b, g0a, axis=1, crop=(cr, cr))
                y2bp = ll.axis_colifilt(hh, qshift[11], qshift[10], axis=1, crop=(cr, cr))
            else:
                y2 = ll.axis_colifilt_sum2(hl, hh, lo, hi, axis=1, crop=(cr, cr))
            Z = ll.axis_colifilt_sum2(y1, y2, lo, hi, axis=2, crop=(cc, cc))
            if bp2:
                ll.axis_colifilt(y2bp, qshift[11], qshift[10], axis=2, crop=(cc, cc), out=Z, accumulate=True)
            level -= 1
        if level == 1 and stop_level == 0:                   # :275-293
            w, g = Yh[0], gm[:, 0]
            whole = None if bp1 else ll.level2d_inverse(Z, w, 0, g, 0, 0, g0o, g1o)
            if whole is not None:
                return whole
            lh = ll.c2q(w, 0, 5, g[0], g[5])
            hl = ll.c2q(w, 2, 3, g[2], g[3])
            hh = ll.c2q(w, 1, 4, g[1], g[4])
            y1 = ll.axis_colfilter_sum2(Z, lh, g0o, g1o, axis=1)
            if bp1:
                y2 = ll.axis_colfilter(hl, g0o, axis=1)
                y2bp = ll.axis_colfilter(hh, biort[5], axis=1)
            else:
                y2 = ll.axis_colfilter_sum2(hl, hh, g0o, g1o, axis=1)
            Z = ll.axis_colfilter_sum2(y1, y2, g0o, g1o, axis=2)
            if bp1:
                ll.axis_colfilter(y2bp, biort[5], axis=2, out=Z, accumulate=True)
        return Z

    def inverse(self, pyramid, gain_mask=None, device_output=False):
        """Perform an *n*-level dual-tree complex wavelet (DTCWT) 2D reconstruction.

        :param pyramid: A :py:class:`Pyramid`-like object (device or NumPy buffers).
        :param gain_mask: (6, nlevels) gains per subband/level, default ones
            (dtcwt/numpy/transform2d.py:190-217).
        :param device_output: return the :class:`DeviceArray` instead of a NumPy array.
        :returns: the reconstruction, shape of the (even-extended) input.
        """
        self._taps()
        nl = nlevels_of(pyramid)
        if nl == 0:
            return pyramid.lowpass
        if gain_mask is not None:
            gain_mask = np.array(gain_mask, dtype=np.float64)
            if gain_mask.shape != (6, nl):
                raise ValueError('gain_mask must have shape (6, %d)' % nl)
        if hasattr(pyramid, 'device_parts'):
            probe = pyramid.hip_lowpass if pyramid.hip_lowpass is not None else pyramid.lowpass
            rdt = np.float32 if probe.dtype in (np.float32, np.complex64) else np.float64
            Yl, Yh = pyramid.device_parts(self.ctx, rdt)
        else:
            low = asfarray(pyramid.lowpass)
            rdt = np.float32 if low.dtype == np.float32 else np.float64
            cdt = np.complex64 if rdt == np.float32 else np.complex128
            Yl = self.ctx.to_device(low, dtype=rdt)
            Yh = tuple(self.ctx.to_device(np.asarray(y), dtype=cdt) for y in pyramid.highpasses)
        if Yl.ndim != 2 or any(y.ndim != 3 for y in Yh):
            raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
        Z = self._inverse_device(Yl.reshape((1,) + Yl.shape), [y.reshape((1,) + y.shape) for y in Yh], gain_mask)
        Z = Z.reshape(Z.shape[1:])
        return Z if device_output else Z.get()

    # ------------------------------------------------------------------ batched API
    _FORMATS = ('nhw', 'chw', 'hwn', 'hwc', 'nchw', 'nhwc')

    @staticmethod
    def _to_nhw(X, data_format):
        """Host array in *data_format* -> ([N, h, w] contiguous, restore-info)."""
        if data_format in ('nhw', 'chw'):
            if X.ndim != 3:
                raise ValueError('%s input must be three-dimensional' % data_format)
            return X, None
        if data_format in ('hwn', 'hwc'):
            if X.ndim != 3:
                raise ValueError('%s input must be three-dimensional' % data_format)
            return np.ascontiguousarray(np.moveaxis(X, 2, 0)), None
        if X.ndim != 4:
            raise ValueError('%s input must be four-dimensional' % data_format)
        if data_format == 'nchw':
            n, ch, h, w = X.shape
            return X.reshape(n * ch, h, w), (n, ch)
        n, h, w, ch = X.shape                                  # nhwc
        return np.ascontiguousarray(np.moveaxis(X, 3, 1)).reshape(n * ch, h, w), (n, ch)

    @staticmethod
    def _from_nhw(A, data_format, info):
        """[N, h, w(, 6)] host array back to *data_format* (subband axis stays last)."""
        if data_format in ('nhw', 'chw'):
            return A
        if data_format in ('hwn', 'hwc'):
            return np.moveaxis(A, 0, 2)
        n, ch = info
        A = A.reshape((n, ch) + A.shape[1:])
        if data_format == 'nchw':
            return A
        return np.moveaxis(A, 1, 3)                            # nhwc

    def forward_channels(self, X, data_format='nhw', nlevels=3, include_scale=False):
        """Transform a batch of equally sized images at once.

        ``data_format``: "nhw"/"chw" (native, zero-copy for device arrays), "hwn"/"hwc",
        "nchw", "nhwc".  Returns a Pyramid whose ``lowpass`` has the batch layout of the
        input and whose ``highpasses[l]`` have one extra trailing axis of size 6, as the
        reference's TF backend does (dtcwt/tf/transform2d.py:179-336)."""
        self._taps()
        data_format = data_format.lower()
        if data_format not in self._FORMATS:
            raise ValueError('The data format must be one of: {}'.format(self._FORMATS))
        if _lib.is_foreign_device_tensor(X):
            X = DeviceArray.wrap(self.ctx, X)
        native = data_format in ('nhw', 'chw')
        if isinstance(X, DeviceArray):
            if not native or X.ndim != 3:
                raise ValueError('device inputs must be in nhw/chw layout')
            if X.dtype not in (np.float32, np.float64):
                raise TypeError('device inputs must be float32 or float64, not %s' % X.dtype)
            Xd, info = X, None
        else:
            X = np.asanyarray(X)
            Xh, info = self._to_nhw(asfarray(X) if np.issubdtype(X.dtype, np.complexfloating) else X, data_format)
            Xd = self.ctx.to_device_float(Xh)
        B, r, c = Xd.shape
        if nlevels == 0:
            Xe = Xd.get()
            if r & 1:
                Xe = np.concatenate((Xe, Xe[:, -1:, :]), axis=1)
            if c & 1:
                Xe = np.concatenate((Xe, Xe[:, :, -1:]), axis=2)
            Xe = self._from_nhw(Xe, data_format, info)
            return Pyramid(Xe, (), ()) if include_scale else Pyramid(Xe, ())
        Yl, Yh, Ys = self._forward_device(Xd, nlevels, include_scale)
        self._log_extension((r, c), (r + (r & 1), c + (c & 1)))
        if native:
            return Pyramid(Yl, tuple(Yh), tuple(Ys) if include_scale else None)
        f = lambda a: self._from_nhw(a.get(), data_format, info)
        return Pyramid(f(Yl), tuple(f(y) for y in Yh), tuple(f(s) for s in Ys) if include_scale else None)

    def inverse_channels(self, pyramid, data_format='nhw', gain_mask=None, device_output=False):
        """Inverse of :meth:`forward_channels` (dtcwt/tf/transform2d.py:422-588)."""
        self._taps()
        data_format = data_format.lower()
        if data_format not in self._FORMATS:
            raise ValueError('The data format must be one of: {}'.format(self._FORMATS))
        nl = nlevels_of(pyramid)
        native = data_format in ('nhw', 'chw')
        if nl == 0:
            return pyramid.lowpass
        if native and hasattr(pyramid, 'device_parts'):
            probe = pyramid.hip_lowpass if pyramid.hip_lowpass is not None else pyramid.lowpass
            rdt = np.float32 if probe.dtype in (np.float32, np.complex64) else np.float64
            Yl, Yh = pyramid.device_parts(self.ctx, rdt)
            info = None
        else:
            low = asfarray(pyramid.lowpass)
            rdt = np.float32 if low.dtype == np.float32 else np.float64
            cdt = np.complex64 if rdt == np.float32 else np.complex128
            lowb, info = self._to_nhw(low, data_format)
            Yl = self.ctx.to_device(lowb, dtype=rdt)
            Yh = []
            for y in pyramid.highpasses:
                y = np.asarray(y)
                # move the subband axis out of the way: treat (..., 6) as 6 more channels
                parts = [self._to_nhw(np.ascontiguousarray(y[..., d]), data_format)[0] for d in range(6)]
                Yh.append(self.ctx.to_device(np.stack(parts, axis=-1), dtype=cdt))
        if Yl.ndim != 3 or any(y.ndim != 4 for y in Yh):
            raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
        Z = self._inverse_device(Yl, list(Yh), gain_mask)
        if device_output and native:
            return Z
        return self._from_nhw(Z.get(), data_format, info)


def dtwavexfm2(X, nlevels=3, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, include_scale=False, ctx=None):
    """Function-style forward (cf. dtcwt/opencl/transform2d.py:22-28, dtcwt/compat.py:107)."""
    t = Transform2d(biort=biort, qshift=qshift, ctx=ctx)
    r = t.forward(X, nlevels=nlevels, include_scale=include_scale)
    if include_scale:
        return r.lowpass, r.highpasses, r.scales
    return r.lowpass, r.highpasses


def dtwaveifm2(Yl, Yh, biort=DEFAULT_BIORT, qshift=DEFAULT_QSHIFT, gain_mask=None, ctx=None):
    """Function-style inverse (dtcwt/compat.py:145-184)."""
    t = Transform2d(biort=biort, qshift=qshift, ctx=ctx)
    return t.inverse(Pyramid(Yl, Yh), gain_mask=gain_mask)
