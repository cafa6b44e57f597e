"""Re-sampling of lowpass images and complex highpass subbands on the device.

Interface, argument meaning, result shapes and dtypes of the reference's ``dtcwt.sampling``
(dtcwt/sampling.py:105-393): ``sample``, ``rescale``, ``upsample`` and their ``*_highpass``
forms, which demodulate each subband by its expected phase advance before interpolating
(`_phase_image`, :167-190).  Inputs may be NumPy arrays or :class:`DeviceArray` (e.g.
``pyramid.hip_highpasses[l]``): a pyramid is re-sampled without leaving HBM, and
``device_output=True`` keeps the result there as well.

All arithmetic runs in the kernels of ``dtcwt_amd/csrc/sampling.hip``; there is no CPU path.
Differences from the reference are confined to precision: a float32 / complex64 input is
interpolated in float32 on the device (the reference promotes Lanczos and highpass results
to float64 / complex128; the returned host array has that promoted dtype, its values are
float32-accurate).
"""
import ctypes

import numpy as np

from dtcwt_amd.hip import _lib
from dtcwt_amd.hip._lib import DeviceArray, check, dtype_code

__all__ = ('sample', 'sample_highpass', 'rescale', 'rescale_highpass', 'upsample', 'upsample_highpass',
           'DTHETA_DX_2D', 'DTHETA_DY_2D')

_W0 = -3 * np.pi / 2.15
_W1 = -np.pi / 2.15
#: The expected phase advances in the x-direction for each subband of the 2D transform
DTHETA_DX_2D = np.array((_W1, _W0, _W0, _W0, _W0, _W1))
#: The expected phase advances in the y-direction for each subband of the 2D transform
DTHETA_DY_2D = np.array((_W0, _W0, _W1, -_W1, -_W0, -_W0))

_METHODS = {'nearest': 0, 'bilinear': 1, 'lanczos': 2}
_pd = ctypes.POINTER(ctypes.c_double)
_pi = ctypes.POINTER(ctypes.c_int)


def _method(method):
    method = 'lanczos' if method is None else method
    if method not in _METHODS:
        raise NotImplementedError('Sampling method "{0}" is not implemented.'.format(method))
    return method


class _Image(object):
    """An image on the device viewed as [H][W][ncomp] of its real dtype."""

    def __init__(self, im, ctx):
        if isinstance(im, DeviceArray):
            self.src_dtype = np.dtype(im.dtype)
            d = im
        else:
            im = np.atleast_2d(np.asanyarray(im))
            self.src_dtype = im.dtype
            if not np.issubdtype(im.dtype, np.inexact):
                im = im.astype(np.float64)          # integer images interpolate in float64
            ctx = ctx if ctx is not None else _lib.default_context()
            d = ctx.to_device(im)
        if d.ndim < 2:
            raise ValueError('image must have at least two dimensions')
        self.dev = d
        self.ctx = d.ctx
        self.dtype = np.dtype(d.dtype)
        self.complex = self.dtype.kind == 'c'
        self.real = np.dtype(np.float32 if self.dtype in (np.float32, np.complex64) else np.float64)
        if self.dtype not in (np.float32, np.float64, np.complex64, np.complex128):
            raise ValueError('unsupported image dtype %s' % self.dtype)
        self.H, self.W = d.shape[:2]
        self.trail = tuple(d.shape[2:])
        self.ncomp = int(np.prod(self.trail, dtype=np.int64)) * (2 if self.complex else 1)


def _coords(ctx, xs, ys):
    xs, ys = np.asanyarray(xs, dtype=np.float64), np.asanyarray(ys, dtype=np.float64)
    if xs.shape != ys.shape:
        raise ValueError('Shape of xs and ys must match')
    return ctx.to_device(xs), ctx.to_device(ys), xs.shape


def _finish(out, host_dtype, device_output):
    if device_output:
        return out
    res = out.get()
    return res if host_dtype is None or res.dtype == host_dtype else res.astype(host_dtype)


def _result_dtype(img, method):
    """dtype of the reference's result (sampling.py:66 casts bilinear back; :97 promotes)."""
    if method == 'lanczos':
        return np.result_type(img.src_dtype, np.float64)
    return img.src_dtype


def _sample_dev(img, dx, dy, shape, method):
    out = DeviceArray(img.ctx, tuple(shape) + img.trail, img.dtype)
    n = int(np.prod(shape, dtype=np.int64))
    check(_lib.lib().dtcwt_hip_sample(img.ctx.handle, dtype_code(img.real), img.dev.ptr, img.H, img.W, img.ncomp,
                                      dx.ptr, dy.ptr, n, _METHODS[method], out.ptr))
    return out


def sample(im, xs, ys, method=None, ctx=None, device_output=False):
    """Sample image at (x,y) given by elements of *xs* and *ys*.  Both *xs* and *ys* must have
    identical shape and output will have this same shape (plus the trailing axes of *im*).
    The location (x,y) refers to the *centre* of ``im[y,x]``.  *method* is one of 'bilinear',
    'lanczos' (default) or 'nearest' (dtcwt/sampling.py:105-129).

    :raise ValueError: if ``xs`` and ``ys`` have differing shapes
    """
    method = _method(method)
    img = _Image(im, ctx)
    dx, dy, shape = _coords(img.ctx, xs, ys)
    return _finish(_sample_dev(img, dx, dy, shape, method), _result_dtype(img, method), device_output)


def _rescale_dev(img, shape, method):
    oh, ow = int(shape[0]), int(shape[1])
    out = DeviceArray(img.ctx, (oh, ow) + img.trail, img.dtype)
    check(_lib.lib().dtcwt_hip_rescale(img.ctx.handle, dtype_code(img.real), img.dev.ptr, img.H, img.W, img.ncomp,
                                       oh, ow, _METHODS[method], out.ptr))
    return out


def rescale(im, shape, method=None, ctx=None, device_output=False):
    """Return a resampled version of *im* scaled to *shape*: the result has the same extent
    (-0.5, w-0.5] x (-0.5, h-0.5] as *im* (dtcwt/sampling.py:131-165)."""
    method = _method(method)
    img = _Image(im, ctx)
    return _finish(_rescale_dev(img, shape, method), _result_dtype(img, method), device_output)


def _upsample_taps(method):
    """Per-axis taps of the factor-two interpolation (dtcwt/sampling.py:280-336)."""
    method = 'lanczos' if method is None else method
    if method == 'lanczos':
        a = 3.0
        offs = np.arange(-a, a + 1)
        return (offs.astype(np.int32), np.sinc(-0.25 - offs) * np.sinc((-0.25 - offs) / a),
                np.sinc(0.25 - offs) * np.sinc((0.25 - offs) / a))
    if method == 'nearest':
        return np.array([0], np.int32), np.array([1.0]), np.array([1.0])
    if method == 'bilinear':
        return np.array([-1, 0, 1], np.int32), np.array([0.25, 0.75, 0.0]), np.array([0.0, 0.75, 0.25])
    raise ValueError('Unknown interpolation mode: {0}'.format(method))


def _upsample_dev(img, method):
    offs, wa, wb = _upsample_taps(method)
    offs = np.ascontiguousarray(offs, np.int32)
    wa, wb = np.ascontiguousarray(wa, np.float64), np.ascontiguousarray(wb, np.float64)
    out = DeviceArray(img.ctx, (2 * img.H, 2 * img.W) + img.trail, img.dtype)
    check(_lib.lib().dtcwt_hip_upsample2(img.ctx.handle, dtype_code(img.real), img.dev.ptr, img.H, img.W, img.ncomp,
                                         len(offs), offs.ctypes.data_as(_pi), wa.ctypes.data_as(_pd),
                                         wb.ctypes.data_as(_pd), out.ptr))
    return out


def upsample(image, method=None, ctx=None, device_output=False):
    """Upsample an image by a factor of two along rows and columns with the given sampling
    method ('nearest', 'bilinear', 'lanczos'; default 'lanczos'): shape (N, M, ...) ->
    (2N, 2M, ...) (dtcwt/sampling.py:338-367)."""
    _upsample_taps(method)                          # the reference's ValueError for unknown modes
    img = _Image(image, ctx)
    host = img.src_dtype if np.issubdtype(img.src_dtype, np.inexact) else np.dtype(np.float64)
    return _finish(_upsample_dev(img, method), host, device_output)


# ------------------------------------------------------------------------------ highpass
def _roll_args(sbs, src):
    sbs = np.asarray(sbs, dtype=np.int64)
    src = np.ascontiguousarray(src, np.int32)
    dx = np.ascontiguousarray(DTHETA_DX_2D[sbs], np.float64)
    dy = np.ascontiguousarray(DTHETA_DY_2D[sbs], np.float64)
    return len(sbs), src.ctypes.data_as(_pi), dx.ctypes.data_as(_pd), dy.ctypes.data_as(_pd), (src, dx, dy)


def _highpass_image(im, ctx):
    img = _Image(im, ctx)
    if not img.complex:                             # real data: the rolled result is complex
        c = img.dev.get().astype(np.complex64 if img.dtype == np.float32 else np.complex128)
        img = _Image(img.ctx.to_device(c), None)
    if len(img.trail) != 1:
        raise ValueError('highpass image must have shape (rows, cols, subbands)')
    return img


def _unwrap(img, sbs):
    """im[:, :, sbs] * exp(-j phase(X, Y)) on the pixel grid (sampling.py:209-213)."""
    if sbs.size < 1 or sbs.size > 6 or sbs.min() < 0 or sbs.max() >= img.trail[0] or sbs.max() > 5:
        raise IndexError('subband index out of range')
    nch, src, dx, dy, keep = _roll_args(sbs, sbs)
    out = DeviceArray(img.ctx, (img.H, img.W, nch), img.dtype)
    check(_lib.lib().dtcwt_hip_phase_roll_grid(img.ctx.handle, dtype_code(img.real), img.dev.ptr, img.H, img.W,
                                               img.trail[0], nch, src, dx, dy, 1.0, 1.0, -1.0, out.ptr))
    return _Image(out, None)


def _rewrap_grid(img, sbs, xscale, yscale):
    nch, src, dx, dy, keep = _roll_args(sbs, np.arange(len(sbs)))
    out = DeviceArray(img.ctx, (img.H, img.W, nch), img.dtype)
    check(_lib.lib().dtcwt_hip_phase_roll_grid(img.ctx.handle, dtype_code(img.real), img.dev.ptr, img.H, img.W,
                                               nch, nch, src, dx, dy, float(xscale), float(yscale), 1.0, out.ptr))
    return out


def _complex_result(img):
    """The reference's phase images are complex128, so its highpass results always are."""
    return np.dtype(np.complex128)


def sample_highpass(im, xs, ys, method=None, sbs=None, ctx=None, device_output=False):
    """As :py:func:`sample` except that the highpass image is first phase shifted to be
    centred on approximately DC; *sbs* selects (and orders) the subbands that are returned
    (dtcwt/sampling.py:192-222)."""
    method = _method(method)
    sbs = np.arange(6) if sbs is None else np.asarray(sbs)
    img = _highpass_image(im, ctx)
    un = _unwrap(img, sbs)
    dx, dy, shape = _coords(un.ctx, xs, ys)
    smp = _sample_dev(un, dx, dy, shape, method)
    nch, src, px, py, keep = _roll_args(sbs, np.arange(len(sbs)))
    out = DeviceArray(un.ctx, tuple(shape) + (nch,), un.dtype)
    n = int(np.prod(shape, dtype=np.int64))
    check(_lib.lib().dtcwt_hip_phase_roll_points(un.ctx.handle, dtype_code(un.real), smp.ptr, n, nch, nch, src,
                                                 px, py, dx.ptr, dy.ptr, 1.0, out.ptr))
    return _finish(out, _complex_result(img), device_output)


def rescale_highpass(im, shape, method=None, sbs=None, ctx=None, device_output=False):
    """As :py:func:`rescale` except that the highpass image is first phase shifted to be
    centred on approximately DC; *sbs* as for :py:func:`sample_highpass`
    (dtcwt/sampling.py:224-278)."""
    method = _method(method)
    sbs = np.arange(6) if sbs is None else np.asarray(sbs)
    img = _highpass_image(im, ctx)
    un = _unwrap(img, sbs)
    res = _Image(_rescale_dev(un, shape, method), None)
    out = _rewrap_grid(res, sbs, float(img.W) / float(shape[1]), float(img.H) / float(shape[0]))
    return _finish(out, _complex_result(img), device_output)


def upsample_highpass(im, method=None, ctx=None, device_output=False):
    """As :py:func:`upsample` except that the highpass image is first phase rolled so that
    the filter has approximate DC centre frequency: the function to use when re-sampling
    complex subband images (dtcwt/sampling.py:369-393)."""
    _upsample_taps(method)
    sbs = np.arange(6)
    img = _highpass_image(im, ctx)
    if img.trail[0] != 6:
        raise ValueError('upsample_highpass needs all six subbands')
    un = _unwrap(img, sbs)
    res = _Image(_upsample_dev(un, method), None)
    out = _rewrap_grid(res, sbs, 0.5, 0.5)
    return _finish(out, _complex_result(img), device_output)
