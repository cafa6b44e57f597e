"""Pin oracle/sampling_oracle.py: compare it with the reference's dtcwt.sampling.

BUILD-CONTAINER ONLY (needs /root/reference).  Either interpreter works:

    /opt/conda/bin/python3.9 oracle/check_sampling_against_reference.py
    python oracle/check_sampling_against_reference.py
"""
import os
import sys
import warnings

import numpy as np

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

import dtcwt.sampling as R                      # noqa: E402  (the reference)
from oracle import sampling_oracle as O        # noqa: E402

checks = fails = 0


def same(a, b, what, tol=1e-12):
    global checks, fails
    checks += 1
    a, b = np.asarray(a), np.asarray(b)
    ok = a.shape == b.shape and a.dtype == b.dtype
    if a.dtype in (np.float32, np.complex64):
        tol = max(tol, 4e-7)            # single-precision results: summation order / NumPy-2 promotion
    if ok and a.size:
        ok = np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() <= tol * max(np.abs(b).max(), 1e-30)
    if not ok:
        fails += 1
        print('MISMATCH', what, a.shape, a.dtype, b.shape, b.dtype)


rs = np.random.RandomState(5)
for dt in (np.float32, np.float64):
    for shape in ((17, 23), (40, 31, 3), (8, 9, 6)):
        im = rs.standard_normal(shape).astype(dt)
        for pts in ((1, 50), (7, 11)):       # the reference's weight broadcasting needs 2-D coordinate arrays
            # well outside the image as well: several reflections
            xs = rs.uniform(-2.5 * shape[1], 3.5 * shape[1], pts)
            ys = rs.uniform(-2.5 * shape[0], 3.5 * shape[0], pts)
            xs.flat[:4] = (-0.5, shape[1] - 0.5, 0.0, shape[1] - 1.0)     # edges and exact pixel centres
            ys.flat[:4] = (shape[0] - 0.5, -0.5, 0.0, 2.5)
            for m in O.METHODS + (None,):
                same(O.sample(im, xs, ys, m), R.sample(im, xs, ys, m), 'sample %s %s %s' % (dt.__name__, shape, m))
        for out in ((30, 50), (9, 5), shape[:2]):
            for m in O.METHODS + (None,):
                same(O.rescale(im, out, m), R.rescale(im, out, m), 'rescale %s %s->%s %s' % (dt.__name__, shape, out, m))
        for m in O.METHODS + (None,):
            same(O.upsample(im, m), R.upsample(im, m), 'upsample %s %s %s' % (dt.__name__, shape, m))
for cdt in (np.complex64, np.complex128):
    for shape in ((12, 15, 6), (33, 20, 6)):
        im = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(cdt)
        xs = rs.uniform(-1.5 * shape[1], 2.5 * shape[1], (9, 13))
        ys = rs.uniform(-1.5 * shape[0], 2.5 * shape[0], (9, 13))
        for m in O.METHODS + (None,):
            for sbs in (None, np.array([0, 2, 3, 5]), np.array([5, 1])):
                same(O.sample_highpass(im, xs, ys, m, sbs), R.sample_highpass(im, xs, ys, m, sbs),
                     'sample_highpass %s %s %s %s' % (cdt.__name__, shape, m, sbs))
                same(O.rescale_highpass(im, (25, 40), m, sbs), R.rescale_highpass(im, (25, 40), m, sbs),
                     'rescale_highpass %s %s %s %s' % (cdt.__name__, shape, m, sbs))
            same(O.upsample_highpass(im, m), R.upsample_highpass(im, m), 'upsample_highpass %s %s %s' % (cdt.__name__, shape, m))
            same(O.sample(im, xs, ys, m), R.sample(im, xs, ys, m), 'sample complex %s %s' % (cdt.__name__, m))
# integer image
im = rs.randint(0, 255, (10, 12))
for m in O.METHODS:
    same(O.upsample(im, m), R.upsample(im, m), 'upsample int %s' % m)
    same(O.rescale(im, (20, 30), m), R.rescale(im, (20, 30), m), 'rescale int %s' % m)
print('%d checks, %d failures (NumPy %s)' % (checks, fails, np.__version__))
sys.exit(1 if fails else 0)
