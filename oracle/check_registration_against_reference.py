"""Pin oracle/registration_oracle.py: compare it with the reference's dtcwt.registration.

BUILD-CONTAINER ONLY (needs /root/reference):

    /opt/conda/bin/python3.9 oracle/check_registration_against_reference.py

Interpreter of record only (NumPy 1.26): under NumPy 2 the reference's batched
`np.linalg.solve(Q, -q)` (registration.py:242) is read with the new "b is a stack of matrices"
rule and raises before anything can be compared.
"""
import os
import sys
import logging
import warnings

import numpy as np

logging.disable(logging.WARNING)
warnings.filterwarnings('ignore')
if not hasattr(np, 'int'):
    np.int = int
if not hasattr(np, 'asfarray'):
    def _asfarray(a, dtype=None):
        a = np.asanyarray(a)
        if dtype is None or not np.issubdtype(np.dtype(dtype), np.inexact):
            dtype = np.float64
        return np.asarray(a, dtype=dtype)
    np.asfarray = _asfarray
if not hasattr(np, 'issubsctype'):
    np.issubsctype = lambda a, b: np.issubdtype(a if isinstance(a, type) else np.dtype(a).type, b)

REF = os.environ.get('DTCWT_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True

import dtcwt                                    # noqa: E402  (the reference)
import dtcwt.registration as R                  # noqa: E402
from oracle import registration_oracle as O     # noqa: E402



class _ListIndexable(np.ndarray):
    """The reference's `_boxfilter` indexes with a *list* of slices/arrays (registration.py:436-442),
    which NumPy < 1.23 read as a tuple and modern NumPy rejects.  Like the np.int alias above this
    restores the old reading -- for arrays of this type only, in this process only."""

    def __getitem__(self, key):
        return super().__getitem__(tuple(key) if isinstance(key, list) else key)


_ref_boxfilter = R._boxfilter
R._boxfilter = lambda X, kernel_size: np.asarray(_ref_boxfilter(np.asarray(X).view(_ListIndexable), kernel_size))

if int(np.__version__.split('.')[0]) >= 2:
    print('skipped: the reference registration module does not run on NumPy %s; use /opt/conda/bin/python3.9' % np.__version__)
    sys.exit(0)

checks = fails = 0


def same(a, b, what, tol=1e-10):
    global checks, fails
    checks += 1
    a, b = np.asarray(a), np.asarray(b)
    ok = a.shape == b.shape
    if ok and a.size:
        ok = np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() <= tol * max(np.abs(b).max(), 1e-30)
    if not ok:
        fails += 1
        print('MISMATCH', what, a.shape, b.shape,
              (np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() / max(np.abs(b).max(), 1e-30)) if a.shape == b.shape else '')


def scene(n, shift, seed):
    """Smooth random texture and a slightly shifted / scaled copy of it."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:n, 0:n] / float(n)
    im = np.zeros((n, n))
    for _ in range(24):
        fx, fy, ph = rs.uniform(1, 9), rs.uniform(1, 9), rs.uniform(0, 6.28)
        im += rs.uniform(0.2, 1) * np.cos(6.283 * (fx * xx + fy * yy) + ph)
    x2, y2 = xx * (1 + shift[2]) + shift[0], yy * (1 + shift[3]) + shift[1]
    im2 = np.zeros((n, n))
    rs = np.random.RandomState(seed)
    for _ in range(24):
        fx, fy, ph = rs.uniform(1, 9), rs.uniform(1, 9), rs.uniform(0, 6.28)
        im2 += rs.uniform(0.2, 1) * np.cos(6.283 * (fx * x2 + fy * y2) + ph)
    return im, im2


rs = np.random.RandomState(9)
for shape in ((12, 16), (9, 7)):
    a = rs.standard_normal(shape) + 1j * rs.standard_normal(shape)
    b = rs.standard_normal(shape) + 1j * rs.standard_normal(shape)
    for w in (None, (0.3, -1.1), tuple(R.EXPECTED_SHIFTS[2])):
        for x, y, nm in zip(O.phasegradient(a, b, w), R.phasegradient(a, b, w), ('dy', 'dx', 'dt')):
            same(x, y, 'phasegradient %s %s' % (shape, nm))
    same(O.confidence(a, b), R.confidence(a, b), 'confidence %s' % (shape,))
    X = rs.standard_normal(shape + (5,))
    for k in (1, 3, 5):
        same(O.boxfilter(X, k), R._boxfilter(X, k), 'boxfilter %d' % k)
Q = rs.standard_normal((4, 5, 27))
same(O.solvetransform(Q), R.solvetransform(Q), 'solvetransform')
same(O.solvetransform(Q[0, 0]), R.solvetransform(Q[0, 0]), 'solvetransform vec')

for n, dt in ((128, np.float64), (128, np.float32), (96, np.float64)):
    im1, im2 = scene(n, (0.01, -0.007, 0.01, -0.005), 3)
    t = dtcwt.Transform2d()
    nl = 5
    p1, p2 = t.forward(im1.astype(dt), nlevels=nl), t.forward(im2.astype(dt), nlevels=nl)
    tol = 1e-9 if dt == np.float64 else 2e-3     # float32 pyramids: angles and products in single precision
    for l, (x, y) in enumerate(zip(O.qtildematrices(p1, p2, [1, 2, 4]), R.qtildematrices(p1, p2, [1, 2, 4]))):
        same(x, y, 'qtildematrices n=%d %s level %d' % (n, dt.__name__, l), tol)
    av = rs.standard_normal(p1.highpasses[3].shape[:2] + (6,)) * 0.01
    for x, y, nm in zip(O.velocityfield(av, (n // 4, n // 4), 'bilinear'), R.velocityfield(av, (n // 4, n // 4), 'bilinear'), 'xy'):
        same(x, y, 'velocityfield ' + nm)
    same(O.warphighpass(p1.highpasses[2], av, 'bilinear'), R.warphighpass(p1.highpasses[2], av, 'bilinear'), 'warphighpass', 1e-9 if dt == np.float64 else 1e-5)
    same(O.warp(im1.astype(dt), av, 'lanczos'), R.warp(im1.astype(dt), av, 'lanczos'), 'warp', 1e-9 if dt == np.float64 else 1e-5)
    same(O.estimatereg(p1, p2), R.estimatereg(p1, p2), 'estimatereg n=%d %s' % (n, dt.__name__), 1e-6 if dt == np.float64 else 5e-2)
    same(O.estimatereg(p1, p2, regshape=(5, 7), levels=[[4, 3], [3, 2]]), R.estimatereg(p1, p2, regshape=(5, 7), levels=[[4, 3], [3, 2]]),
         'estimatereg custom', 1e-6 if dt == np.float64 else 5e-2)
print('%d checks, %d failures (NumPy %s)' % (checks, fails, np.__version__))
sys.exit(1 if fails else 0)
