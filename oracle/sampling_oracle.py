"""CPU restatement of the reference's subband re-sampling (rjw57/dtcwt, dtcwt/sampling.py).

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by dtcwt_amd (the product has no CPU
path).  Parity PINNED: oracle/check_sampling_against_reference.py compares every function
here with the imported reference on random and adversarial inputs, and
tests/golden/sampling.npz holds outputs of the reference itself.

Everything is written as one gather-and-weight form:

    out[p] = sum_{a, b}  wy_b(fy_p) * wx_a(fx_p) * im[rho(floor(y_p) + b), rho(floor(x_p) + a)]

with rho the half-sample symmetric reflection of integer indices (what
``reflect(., -0.5, n - 0.5).astype(int)`` computes for integer-valued input,
sampling.py:36-40, utils.py:136-153) and (taps, weights) per method:

    nearest   (sampling.py:42-43)   one tap at round-half-even(x)
    bilinear  (:45-66)              a in {0, 1},       w = (1 - f, f)
    lanczos   (:68-103)             a in {-2 .. 3},    w = L(f - a),  L(t) = sinc(t) sinc(t / 3)

Highpass variants (:167-278, :369-393) demodulate each subband by its expected phase
advance before interpolating and re-modulate at the sample positions.
"""
import numpy as np

from oracle.dtcwt_oracle import reflect_index

_W0 = -3 * np.pi / 2.15
_W1 = -np.pi / 2.15
#: expected phase advance per sample along x / y of the six 2-D subbands (sampling.py:26-33)
DTHETA_DX_2D = np.array((_W1, _W0, _W0, _W0, _W0, _W1))
DTHETA_DY_2D = np.array((_W0, _W0, _W1, -_W1, -_W0, -_W0))

METHODS = ('nearest', 'bilinear', 'lanczos')


def _lanczos(t, a=3.0):
    return np.sinc(t) * np.sinc(t / a)


def _kernel(method):
    """(tap offsets, weight function of the fractional part) of one axis."""
    if method == 'lanczos':
        offs = np.arange(-2, 4)
        return offs, lambda f: np.stack([_lanczos(f - d) for d in offs])
    raise NotImplementedError('Sampling method "{0}" is not implemented.'.format(method))


def _gather(im, yi, xi):
    h, w = im.shape[:2]
    return im[reflect_index(yi.astype(np.int64), h), reflect_index(xi.astype(np.int64), w), ...]


def sample(im, xs, ys, method=None):
    """dtcwt/sampling.py:105-129.  Result dtype as the reference's: the image's for
    'nearest' and 'bilinear' (:66), float64-promoted for 'lanczos' (:97)."""
    method = 'lanczos' if method is None else method
    im = np.atleast_2d(np.asanyarray(im))
    xs, ys = np.asanyarray(xs, dtype=np.float64), np.asanyarray(ys, dtype=np.float64)
    if method == 'nearest':
        return _gather(im, np.round(ys), np.round(xs))
    if xs.shape != ys.shape:
        raise ValueError('Shape of xs and ys must match')
    x0, y0 = np.floor(xs), np.floor(ys)
    if method == 'bilinear':
        # the reference's association (x first, :63-66), so that the cast back to an integer
        # image dtype truncates the very same float64 values
        fx = (xs - x0).reshape(xs.shape + (1,) * (im.ndim - 2))
        fy = (ys - y0).reshape(xs.shape + (1,) * (im.ndim - 2))
        lower = (1.0 - fx) * _gather(im, y0, x0) + fx * _gather(im, y0, x0 + 1)
        upper = (1.0 - fx) * _gather(im, y0 + 1, x0) + fx * _gather(im, y0 + 1, x0 + 1)
        return ((1.0 - fy) * lower + fy * upper).astype(im.dtype)
    offs, wfun = _kernel(method)
    wx, wy = wfun(xs - x0), wfun(ys - y0)
    extra = (1,) * (im.ndim - 2)
    acc = None
    for a, da in enumerate(offs):
        for b, db in enumerate(offs):
            w = (wx[a] * wy[b]).reshape(xs.shape + extra)
            term = w * _gather(im, y0 + db, x0 + da)
            acc = term if acc is None else acc + term
    return acc


def _grid(src_shape, dst_shape):
    """Centres of the pixels of a dst_shape array in the coordinates of a src_shape one
    (:141-163): x(dx) = (dx + 1/2) sw / dw - 1/2."""
    sh, sw = src_shape[:2]
    dh, dw = dst_shape[:2]
    dxs, dys = np.meshgrid(np.arange(dw), np.arange(dh))
    return (float(sw) / float(dw)) * (dxs + 0.5) - 0.5, (float(sh) / float(dh)) * (dys + 0.5) - 0.5


def rescale(im, shape, method=None):
    """dtcwt/sampling.py:131-165."""
    im = np.asanyarray(im)
    sxs, sys_ = _grid(im.shape, shape)
    return sample(im, sxs, sys_, method)


def _phase(xs, ys, sign, sbs):
    """exp(sign * j * (dtheta_dx x + dtheta_dy y)) per selected subband, stacked last (:167-190)."""
    xs, ys = np.asanyarray(xs, dtype=np.float64), np.asanyarray(ys, dtype=np.float64)
    return np.stack([np.exp(sign * 1j * (DTHETA_DX_2D[s] * xs + DTHETA_DY_2D[s] * ys)) for s in sbs], axis=-1)


def _unwrapped(im, sbs):
    X, Y = np.meshgrid(np.arange(im.shape[1]), np.arange(im.shape[0]))
    return im[:, :, sbs] * _phase(X, Y, -1, sbs)


def sample_highpass(im, xs, ys, method=None, sbs=None):
    """dtcwt/sampling.py:192-222."""
    sbs = np.arange(6) if sbs is None else np.asarray(sbs)
    im = np.asanyarray(im)
    return _phase(xs, ys, +1, sbs) * sample(_unwrapped(im, sbs), xs, ys, method)


def rescale_highpass(im, shape, method=None, sbs=None):
    """dtcwt/sampling.py:224-278."""
    sbs = np.arange(6) if sbs is None else np.asarray(sbs)
    im = np.asanyarray(im)
    sxs, sys_ = _grid(im.shape, shape)
    return sample(_unwrapped(im, sbs), sxs, sys_, method) * _phase(sxs, sys_, +1, sbs)


def upsample_taps(method=None):
    """(offsets, weights of the even outputs, weights of the odd outputs) of the factor-two
    interpolation of one axis (:280-336): outputs 2i and 2i+1 sit at i - 1/4 and i + 1/4."""
    method = 'lanczos' if method is None else method
    if method == 'lanczos':
        offs = np.arange(-3, 4)
        return offs, _lanczos(-0.25 - offs), _lanczos(0.25 - offs)
    if method == 'nearest':
        return np.array([0]), np.array([1.0]), np.array([1.0])
    if method == 'bilinear':
        return np.array([-1, 0, 1]), np.array([0.25, 0.75, 0.0]), np.array([0.0, 0.75, 0.25])
    raise ValueError('Unknown interpolation mode: {0}'.format(method))


def _as_float(a):
    a = np.asanyarray(a)
    return a if np.issubdtype(a.dtype, np.inexact) else a.astype(np.float64)


def _upsample_axis(X, axis, method):
    offs, wa, wb = upsample_taps(method)
    n = X.shape[axis]
    shape = list(X.shape)
    shape[axis] *= 2
    out = np.zeros(shape, dtype=X.dtype)
    ev = [slice(None)] * X.ndim
    od = [slice(None)] * X.ndim
    ev[axis], od[axis] = slice(0, None, 2), slice(1, None, 2)
    for d, a, b in zip(offs, wa, wb):
        g = np.take(X, reflect_index(np.arange(n) + d, n), axis=axis)
        out[tuple(ev)] += a * g
        out[tuple(od)] += b * g
    return out


def upsample(image, method=None):
    """dtcwt/sampling.py:338-367: columns (axis 1) of the transposed image first, i.e.
    axis 0, then axis 1."""
    image = np.atleast_2d(_as_float(image))
    return _upsample_axis(_upsample_axis(image, 0, method), 1, method)


def upsample_highpass(im, method=None):
    """dtcwt/sampling.py:369-393."""
    im = np.atleast_2d(_as_float(im))
    sbs = np.arange(6)
    dxs, dys = np.meshgrid(np.arange(im.shape[1] * 2), np.arange(im.shape[0] * 2))
    sxs, sys_ = 0.5 * (dxs + 0.5) - 0.5, 0.5 * (dys + 0.5) - 0.5
    return upsample(_unwrapped(im, sbs), method) * _phase(sxs, sys_, +1, sbs)
