"""Shared bits of the GPU suite."""
import numpy as np

from tests._golden import rel_err

LOW_TOL = 1e-6      # low-level filters, float32 (the reference's tests/util.py:11 uses 1e-6 max-abs)
XFM_TOL = 1e-6      # transforms, float32: max|a-b| / max|b| per subband (north star: 1e-6 relative)
INV_TOL = 3e-6      # float32 reconstructions accumulate two more filter stages per level
F64_TOL = 1e-12


def assert_close(a, b, tol, what=''):
    e = rel_err(a, b)
    assert e <= tol, '%s rel err %.3e > %g' % (what, e, tol)


def as_f64(X):
    return np.asarray(X, dtype=np.float64)


def cast_pyramid(p, real):
    """Pyramid-like with arrays cast to real / matching complex dtype."""
    from oracle import dtcwt_oracle as o
    cplx = np.complex64 if real == np.float32 else np.complex128
    return o.Pyramid(p.lowpass.astype(real), tuple(None if y is None else y.astype(cplx) for y in p.highpasses))


def assert_pyramids_close(p, q, tol, same_dtype=True):
    assert p.lowpass.shape == q.lowpass.shape
    assert_close(p.lowpass, q.lowpass, tol, 'Yl')
    assert len(p.highpasses) == len(q.highpasses)
    for l, (a, b) in enumerate(zip(p.highpasses, q.highpasses)):
        if a is None or b is None:
            assert a is None and b is None
            continue
        if same_dtype:
            assert a.dtype == b.dtype, (a.dtype, b.dtype)
        assert_close(a, b, tol, 'Yh[%d]' % l)
    if q.scales is not None:
        assert p.scales is not None
        for l, (a, b) in enumerate(zip(p.scales, q.scales)):
            assert_close(a, b, tol, 'Yscale[%d]' % l)
