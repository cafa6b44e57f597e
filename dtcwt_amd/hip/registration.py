"""DT-CWT image registration on the device.

Interface and arithmetic of the reference's ``dtcwt.registration`` (dtcwt/registration.py):
``estimatereg``, ``velocityfield``, ``warp``, ``warptransform`` (its ``__all__``, :22-27) and
the building blocks ``qtildematrices``, ``solvetransform``, ``warphighpass``,
``normsample``, ``normsamplehighpass``.  The pyramids of the two images stay in HBM: every
per-pixel loop (confidence, phase gradients, the 27-element Q-tilde accumulation, box
filter, bilinear rescale, the 6x6 solves, warps) is a kernel of
``dtcwt_amd/csrc/registration.hip`` / ``sampling.hip``; this module only sequences them the
way the reference's ``estimatereg`` does (:301-372).  There is no CPU path.

Pyramids may be the ``hip`` backend's (device-resident, used in place) or any object with
the ``dtcwt.Pyramid`` attributes (uploaded).
"""
import ctypes

import numpy as np

from dtcwt_amd.hip import _lib
from dtcwt_amd.hip import sampling as _s
from dtcwt_amd.hip._lib import DeviceArray, check, dtype_code
from dtcwt_amd.hip.common import Pyramid, nlevels_of

__all__ = ['estimatereg', 'velocityfield', 'warp', 'warptransform',
           'qtildematrices', 'solvetransform', 'warphighpass', 'normsample', 'normsamplehighpass']

#: Horizontal and vertical expected phase shifts for each subband (dtcwt/registration.py:29)
EXPECTED_SHIFTS = np.array(((-1, -3), (-3, -3), (-3, -1), (-3, 1), (-3, 3), (-1, 3))) * np.pi / 2.15

_pd = ctypes.POINTER(ctypes.c_double)


def _ctx_of(*objs):
    for o in objs:
        if isinstance(o, DeviceArray):
            return o.ctx
        for h in getattr(o, 'hip_highpasses', None) or ():
            if h is not None:
                return h.ctx
    return _lib.default_context()


def _dev_highpass(pyr, level, ctx):
    h = getattr(pyr, 'hip_highpasses', None)
    d = h[level] if h is not None else None
    if d is None:
        a = np.asarray(pyr.highpasses[level])
        if a.dtype not in (np.complex64, np.complex128):
            a = a.astype(np.complex128)
        d = ctx.to_device(a)
    if d.ndim != 3 or d.shape[2] != 6:
        raise ValueError('highpass level must have shape (rows, cols, 6)')
    return d


def _real_code(d):
    return dtype_code(np.float32 if d.dtype == np.complex64 else np.float64)


def _f64(ctx, a):
    return a if isinstance(a, DeviceArray) else ctx.to_device(np.asarray(a, dtype=np.float64))


def _qtilde_dev(h1, h2, epsilon=1e-6):
    if h1.shape != h2.shape:
        raise ValueError('Subbands should have identical size')
    if h1.dtype != h2.dtype:
        h2 = h1.ctx.to_device(h2.get().astype(h1.dtype))
    H, W = h1.shape[:2]
    out = DeviceArray(h1.ctx, (H, W, 27), np.float64)
    check(_lib.lib().dtcwt_hip_qtilde(h1.ctx.handle, _real_code(h1), h1.ptr, h2.ptr, H, W, float(epsilon), out.ptr))
    return out


def qtildematrices(t_ref, t_target, levels, device_output=False):
    r"""Compute :math:`\tilde{Q}` matrices for given (0-based) *levels* of two transformed
    images: a list of NxMx27 arrays, NxM the shape of the level's subbands
    (dtcwt/registration.py:140-214)."""
    ctx = _ctx_of(t_ref, t_target)
    res = [_qtilde_dev(_dev_highpass(t_ref, l, ctx), _dev_highpass(t_target, l, ctx)) for l in levels]
    return res if device_output else [r.get() for r in res]


def _solve_dev(Qt):
    n = int(np.prod(Qt.shape[:-1], dtype=np.int64))
    out = DeviceArray(Qt.ctx, tuple(Qt.shape[:-1]) + (6,), np.float64)
    check(_lib.lib().dtcwt_hip_solve6(Qt.ctx.handle, Qt.ptr, n, out.ptr))
    return out


def solvetransform(Qtilde_vec, device_output=False):
    r"""Solve for the affine parameter vector :math:`a = -\mathbf{Q}^{-1}\mathbf{q}` of every
    27-element :math:`\tilde{Q}` vector along the last axis (dtcwt/registration.py:216-250)."""
    Qt = _f64(_ctx_of(Qtilde_vec), Qtilde_vec)
    if Qt.shape[-1] != 27:
        raise ValueError('last axis must hold the 27 Q-tilde elements')
    out = _solve_dev(Qt)
    return out if device_output else out.get()


def _velocity_dev(avecs, shape, method):
    """(vx, vy) DeviceArrays of *shape* (dtcwt/registration.py:374-395)."""
    h, w = avecs.shape[:2]
    vx, vy = DeviceArray(avecs.ctx, (h, w), np.float64), DeviceArray(avecs.ctx, (h, w), np.float64)
    check(_lib.lib().dtcwt_hip_affine_velocity(avecs.ctx.handle, avecs.ptr, h, w, vx.ptr, vy.ptr))
    return (_s.rescale(vx, shape[:2], method, device_output=True),
            _s.rescale(vy, shape[:2], method, device_output=True))


def velocityfield(avecs, shape, method=None, device_output=False):
    """x- and y- components, of shape *shape*, of the velocity field implied by the affine
    distortion parameters *avecs* (as returned by :py:func:`estimatereg`), in normalised units
    where the image has width and height of unity (dtcwt/registration.py:374-395)."""
    av = _f64(_ctx_of(avecs), avecs)
    vx, vy = _velocity_dev(av, shape, method)
    return (vx, vy) if device_output else (vx.get(), vy.get())


def _warp_coords(av, H, W, method):
    vx, vy = _velocity_dev(av, (H, W), method)
    xs, ys = DeviceArray(av.ctx, (H, W), np.float64), DeviceArray(av.ctx, (H, W), np.float64)
    check(_lib.lib().dtcwt_hip_warp_coords(av.ctx.handle, vx.ptr, vy.ptr, H, W, xs.ptr, ys.ptr))
    return xs, ys


def warp(I, avecs, method=None, device_output=False):
    """Warp an image according to the velocity field implied by *avecs*
    (dtcwt/registration.py:410-415)."""
    ctx = _ctx_of(I, avecs)
    av = _f64(ctx, avecs)
    img = _s._Image(I, ctx)
    xs, ys = _warp_coords(av, img.H, img.W, method)
    m = _s._method(method)
    out = _s._sample_dev(img, xs, ys, (img.H, img.W), m)
    return _s._finish(out, _s._result_dtype(img, m), device_output)


def _warphighpass_dev(Yh, av, method):
    m = _s._method(method)
    sbs = np.arange(6)
    img = _s._highpass_image(Yh, None)
    xs, ys = _warp_coords(av, img.H, img.W, method)
    un = _s._unwrap(img, sbs)
    smp = _s._sample_dev(un, xs, ys, (img.H, img.W), m)
    nch, src, px, py, keep = _s._roll_args(sbs, np.arange(6))
    out = DeviceArray(img.ctx, (img.H, img.W, 6), un.dtype)
    check(_lib.lib().dtcwt_hip_phase_roll_points(img.ctx.handle, dtype_code(un.real), smp.ptr, img.H * img.W, 6, 6, src,
                                                 px, py, xs.ptr, ys.ptr, 1.0, out.ptr))
    return out


def warphighpass(Yh, avecs, method=None, device_output=False):
    """Warp a highpass subband image according to the velocity field implied by *avecs*,
    'de-rotating' it before sampling and 're-rotating' afterwards (dtcwt/registration.py:397-408)."""
    ctx = _ctx_of(Yh, avecs)
    d = Yh if isinstance(Yh, DeviceArray) else ctx.to_device(np.asarray(Yh))
    out = _warphighpass_dev(d, _f64(ctx, avecs), method)
    return out if device_output else out.get().astype(np.complex128)


def normsamplehighpass(Yh, xs, ys, method=None, **kw):
    """Sample NxMx6 subband responses at *normalised* co-ordinates (image width and height are
    both unity) (dtcwt/registration.py:252-261)."""
    return _s.sample_highpass(Yh, np.asarray(xs) * Yh.shape[1], np.asarray(ys) * Yh.shape[0], method=method, **kw)


def normsample(Yh, xs, ys, method=None, **kw):
    """Sample an NxM image at *normalised* co-ordinates (dtcwt/registration.py:263-272)."""
    return _s.sample(Yh, np.asarray(xs) * Yh.shape[1], np.asarray(ys) * Yh.shape[0], method=method, **kw)


def warptransform(t, avecs, levels, method=None):
    """Return a warped version of a transformed image acting only on the specified (0-based)
    *levels*: a shallow clone of *t* whose warped levels are new, device-resident arrays
    (dtcwt/registration.py:274-299)."""
    ctx = _ctx_of(t, avecs)
    av = _f64(ctx, avecs)
    if isinstance(t, Pyramid):          # raw entries: nothing is copied to the host
        hp, low, scales = list(t._high), t._low, t._scales
    else:
        hp, low, scales = list(t.highpasses), t.lowpass, getattr(t, 'scales', None)
    for l in levels:
        hp[l] = _warphighpass_dev(_dev_highpass(t, l, ctx), av, method)
    return Pyramid(low, tuple(hp), scales)


def _colsum(Q):
    K = Q.shape[-1]
    n = int(np.prod(Q.shape[:-1], dtype=np.int64))
    out = DeviceArray(Q.ctx, (K,), np.float64)
    check(_lib.lib().dtcwt_hip_colsum(Q.ctx.handle, Q.ptr, n, K, out.ptr))
    return out


def _axpy(alpha, x, y):
    check(_lib.lib().dtcwt_hip_axpy(y.ctx.handle, int(np.prod(y.shape, dtype=np.int64)), float(alpha), x.ptr, y.ptr))


def _boxfilter_dev(X, kernel_size):
    if kernel_size % 2 == 0:
        raise ValueError('Kernel size must be odd')
    H, W = X.shape[:2]
    K = int(np.prod(X.shape[2:], dtype=np.int64))
    out = DeviceArray(X.ctx, X.shape, np.float64)
    check(_lib.lib().dtcwt_hip_boxfilter(X.ctx.handle, X.ptr, H, W, K, int(kernel_size), out.ptr))
    return out


def _estimatereg_native(ctx, source, reference, nlevels, shp, levels, device_output):
    used = sorted(set(l for g in levels for l in g))
    if any(l < 0 or l >= nlevels for l in used):
        raise IndexError('level index out of range')
    src = {l: _dev_highpass(source, l, ctx) for l in used}
    ref = {l: _dev_highpass(reference, l, ctx) for l in used}
    dt = src[used[0]].dtype
    for l in used:
        if src[l].shape != ref[l].shape:
            raise ValueError('Subbands should have identical size')
        if src[l].dtype != dt:
            src[l] = ctx.to_device(src[l].get().astype(dt))
        if ref[l].dtype != dt:
            ref[l] = ctx.to_device(ref[l].get().astype(dt))
    vp = ctypes.c_void_p
    ps = (vp * nlevels)(*[src[l].ptr if l in src else None for l in range(nlevels)])
    pr = (vp * nlevels)(*[ref[l].ptr if l in ref else None for l in range(nlevels)])
    shapes = (ctypes.c_int64 * (2 * nlevels))()
    for l in used:
        shapes[2 * l], shapes[2 * l + 1] = src[l].shape[0], src[l].shape[1]
    sizes = (ctypes.c_int * len(levels))(*[len(g) for g in levels])
    flat = [l for g in levels for l in g]
    lv = (ctypes.c_int * max(len(flat), 1))(*flat)
    avecs = DeviceArray(ctx, shp + (6,), np.float64)
    check(_lib.lib().dtcwt_hip_estimatereg(ctx.handle, _real_code(src[used[0]]), nlevels, ps, pr, shapes, shp[0], shp[1],
                                           len(levels), sizes, lv, avecs.ptr))
    return avecs if device_output else avecs.get()


def estimatereg(source, reference, regshape=None, levels=None, device_output=False, native=True):
    """Estimate the registration which will map *source* to *reference* (transformed images
    with the ``dtcwt.Pyramid`` API).  The local affine distortion is estimated at 8x8 pixel
    scales: returns a NxMx6 array whose 6-vector at (N, M) holds the affine distortion
    parameters of that block (default shape: that of the level-4 subbands, or *regshape*).
    *levels*, if not None, is a sequence of sequences of 0-based level indices to use
    (dtcwt/registration.py:301-372).  Use :py:func:`velocityfield` to turn the result into a
    velocity field.

    *native* (default): the whole kernel sequence is issued by one library call
    (``dtcwt_hip_estimatereg``); ``native=False`` issues the same kernels one by one from here."""
    ctx = _ctx_of(source, reference)
    nlevels = nlevels_of(source)
    if regshape is None:
        h3 = getattr(source, 'hip_highpasses', None)
        shp = (h3[3].shape if h3 is not None and h3[3] is not None else source.highpasses[3].shape)[:2]
    else:
        shp = tuple(regshape[:2])
    shp = (int(shp[0]), int(shp[1]))
    if levels is None:                                          # :330-337
        levels = [list(x for x in range(nlevels - 1, nlevels - 3, -1) if x >= 0)]
        for s in np.arange(nlevels - 1, 0, -0.5):
            refine = list(int(np.floor(s)) - x for x in range(2) if s - x >= 2)
            if len(refine) >= 2:
                levels.append(refine)

    if native:
        return _estimatereg_native(ctx, source, reference, nlevels, shp, levels, device_output)

    # the same sequence launch by launch (kept for inspection of the intermediate results)
    # initial global transform: Q-tilde summed over every pixel of the coarsest levels
    Qt = ctx.zeros((27,), np.float64)
    for l in levels[0]:
        _axpy(1.0, _colsum(_qtilde_dev(_dev_highpass(source, l, ctx), _dev_highpass(reference, l, ctx))), Qt)
    a = _solve_dev(Qt.reshape(1, 27)).get().reshape(6)
    avecs = DeviceArray(ctx, shp + (6,), np.float64)
    row = np.ascontiguousarray(a, np.float64)
    check(_lib.lib().dtcwt_hip_fill_rows(ctx.handle, shp[0] * shp[1], 6, row.ctypes.data_as(_pd), avecs.ptr))

    # refinement: warp the levels with the current estimate, re-estimate per block
    for est in levels[1:]:
        qts = None
        for l in est:
            warped = _warphighpass_dev(_dev_highpass(source, l, ctx), avecs, 'bilinear')
            q = _qtilde_dev(warped, _dev_highpass(reference, l, ctx))
            q = _s.rescale(_boxfilter_dev(q, 3), shp, 'bilinear', device_output=True)
            if qts is None:
                qts = q
            else:
                _axpy(1.0, q, qts)
        if qts is None:
            continue
        _axpy(1.0, _solve_dev(qts), avecs)
    return avecs if device_output else avecs.get()
