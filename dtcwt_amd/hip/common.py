"""Pyramid container of the ``hip`` backend.

Interface of dtcwt/numpy/common.py:5-32 (``lowpass``, ``highpasses``, ``scales``) with the
device-resident behaviour of the reference's OpenCL pyramid
(dtcwt/opencl/transform2d.py:30-84): the subband buffers stay in HBM
(``hip_lowpass`` / ``hip_highpasses`` / ``hip_scales``, :class:`DeviceArray` instances) and
the NumPy attributes are filled by ONE device-to-host copy on first access and memoised.
Constructed from NumPy arrays it behaves like the NumPy pyramid and uploads lazily when
an inverse transform needs device buffers.

Once an entry has been read through ``lowpass`` / ``highpasses`` / ``scales`` the host array is
the authoritative copy of that entry -- the reference's idiom
``p = t.forward(x); p.highpasses[2][mask] = 0; t.inverse(p)`` edits it in place -- so the inverse
transforms upload it again instead of using the device buffer it came from.  Code that wants
to stay on the device uses the ``hip_*`` handles and never touches the NumPy attributes.
"""
import weakref

import numpy as np

from dtcwt_amd.utils import asfarray
from dtcwt_amd.hip._lib import DeviceArray

__all__ = ['Pyramid', 'nlevels_of']


def _is_dev(x):
    return isinstance(x, DeviceArray)


def _drain(pending):
    """Finaliser of a Pyramid dropped with downloads still in flight: wait for them.  The copies read the pyramid's
    device buffers and write pooled page-locked host buffers; both go back to their pools when the pyramid dies, and
    neither pool knows about the copy stream -- the next blocking download of the same size would be handed a buffer
    a stale DMA is still writing.  Runs before the pyramid's attributes are released."""
    ctxs = {}
    for _arr, ctx in pending.values():
        ctxs[id(ctx)] = ctx
    # wait FIRST, release afterwards: dropping the destination arrays hands their page-locked buffers back to the host
    # pool (or frees them), and another thread could be given one while the DMA is still writing it
    for ctx in ctxs.values():
        try:
            ctx.copy_sync()
        except Exception:
            pass
    pending.clear()


def nlevels_of(pyramid):
    """len(pyramid.highpasses) that does not make a device-resident pyramid copy its levels
    to the host (the ``highpasses`` property of the hip Pyramid materialises them)."""
    n = getattr(pyramid, 'nlevels', None)
    return n if isinstance(n, int) else len(pyramid.highpasses)


class Pyramid(object):
    def __init__(self, lowpass, highpasses, scales=None):
        self._low = lowpass if _is_dev(lowpass) else asfarray(lowpass)
        self._high = tuple(x if (x is None or _is_dev(x)) else asfarray(x) for x in highpasses)
        self._scales = None if scales is None else tuple(x if _is_dev(x) else asfarray(x) for x in scales)
        self._host = {}
        self._pending = {}      # key -> (destination array, context) of downloads started by prefetch()
        weakref.finalize(self, _drain, self._pending)

    # ---- raw device handles (None where the entry was given as a host array) ----
    @property
    def hip_lowpass(self):
        return self._low if _is_dev(self._low) else None

    @property
    def hip_highpasses(self):
        return tuple(x if _is_dev(x) else None for x in self._high)

    @property
    def hip_scales(self):
        if self._scales is None:
            return None
        return tuple(x if _is_dev(x) else None for x in self._scales)

    @property
    def nlevels(self):
        """Number of highpass levels -- without touching (copying to the host) any of them."""
        return len(self._high)

    # ---- NumPy-compatible attributes, memoised --------------------------------
    def prefetch(self):
        """Start downloading every device-resident entry now, without waiting: the copies run on the context's
        copy stream into page-locked pooled buffers, behind the kernels that produce them and beside whatever is
        enqueued next (the next image's upload and transform).  The first access to ``lowpass`` / ``highpasses``
        / ``scales`` waits for them.  Transform2d.forward calls this for NumPy inputs -- the literal drop-in use,
        where the caller is certain to read the result on the host.  A prefetched entry is a snapshot: kernels that
        edit the device buffers afterwards are not reflected by the NumPy attributes (use the ``hip_*`` handles).
        A pyramid dropped unread waits for its downloads before its buffers are recycled."""
        entries = [('l', self._low)] + [(('h', i), x) for i, x in enumerate(self._high)]
        if self._scales is not None:
            entries += [(('s', i), x) for i, x in enumerate(self._scales)]
        for key, x in entries:
            if x is not None and _is_dev(x) and key not in self._host and key not in self._pending:
                self._pending[key] = (x.get_async(), x.ctx)
        return self

    def _get(self, key, x):
        if x is None or not _is_dev(x):
            return x
        if key in self._pending:
            arr, ctx = self._pending.pop(key)
            ctx.copy_sync()
            self._host[key] = arr
        if key not in self._host:
            self._host[key] = x.get()
        return self._host[key]

    @property
    def lowpass(self):
        return self._get('l', self._low)

    @property
    def highpasses(self):
        return tuple(self._get(('h', i), x) for i, x in enumerate(self._high))

    @property
    def scales(self):
        if self._scales is None:
            return None
        return tuple(self._get(('s', i), x) for i, x in enumerate(self._scales))

    # ---- used by the inverse transforms ---------------------------------------
    def device_parts(self, ctx, real_dtype=None):
        """(lowpass, highpasses) as DeviceArrays on *ctx* (uploading host arrays);
        highpass entries may be None.  An entry whose host view has been handed out is uploaded from
        that view: the caller may have edited it."""
        entries = [('l', self._low)] + [(('h', i), x) for i, x in enumerate(self._high)]
        if real_dtype is not None:
            # one precision for the whole pyramid, as NumPy's promotion gives the reference: float64 as soon as any
            # entry is 64-bit (a float32 lowpass next to complex128 subbands must not reach a kernel as if they matched)
            cur = [self._host.get(k, x) for k, x in entries if x is not None]
            if any(np.dtype(x.dtype) in (np.dtype(np.float64), np.dtype(np.complex128)) for x in cur):
                real_dtype = np.float64

        def up(x, cplx, key):
            if x is not None and key in self._host:
                x = self._host[key]
            if x is None:
                return x
            if real_dtype is not None:
                dt = np.dtype((np.complex64 if real_dtype == np.float32 else np.complex128) if cplx else real_dtype)
                if _is_dev(x):
                    return x if x.dtype == dt else ctx.to_device(x.get(), dtype=dt)     # rare: through the host
                return ctx.to_device(x, dtype=dt)
            return x if _is_dev(x) else ctx.to_device(x)
        return up(self._low, False, 'l'), tuple(up(x, True, ('h', i)) for i, x in enumerate(self._high))
