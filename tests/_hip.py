"""Shared bits of the GPU suite."""
import os

import numpy as np

from tests._golden import rel_err

LOW_TOL = 1e-6      # low-level filters, float32 (the reference's tests/util.py:11 uses 1e-6 max-abs)
XFM_TOL = 1e-6      # transforms, float32: max|a-b| / max|b| per subband (north star: 1e-6 relative)
INV_TOL = 1e-6      # float32 reconstructions: the north-star bound as well (gpurun_out/parity_worst.json lists the five worst tests of a session)
F64_TOL = 1e-12


# worst relative error seen per tolerance class in this session; tests/conftest.py writes it to
# gpurun_out/parity_worst.json so that a drift from 2e-7 to 2.9e-6 under a 3e-6 bound does not pass unseen
WORST = {}


def assert_close(a, b, tol, what=''):
    e = rel_err(a, b)
    key = '%g' % tol
    w = WORST.setdefault(key, {'worst': 0.0, 'what': '', 'n': 0})
    w['n'] += 1
    test = os.environ.get('PYTEST_CURRENT_TEST', '').replace(' (call)', '')
    if e > w['worst']:
        w['worst'], w['what'], w['test'] = float(e), what, test
    # the five worst TESTS of the class (one entry per test), so that one outlier does not hide the rest
    top = w.setdefault('top', [])
    mine = [t for t in top if t[2] == test]
    if mine:
        if e > mine[0][0]:
            mine[0][0], mine[0][1] = float(e), what
    else:
        top.append([float(e), what, test])
    top.sort(key=lambda t: -t[0])
    del top[5:]
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
