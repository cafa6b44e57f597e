"""Standalone low-level filters on one MI355X: the public
colfilter / coldfilt / colifilt of the hip backend on a device-resident 4096^2 image."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Context
from dtcwt_amd.hip import lowlevel as ll

ctx = Context(0)
h0o, g0o, h1o, g1o = biort('near_sym_a')
h0a, h0b = qshift('qshift_a')[:2]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.device_sync()
    return (time.perf_counter() - t0) / reps


for dt in (np.float32, np.float64):
    X = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)).astype(dt))
    e = X.dtype.itemsize
    n = 4096 * 4096
    for axis in (0, 1):
        for name, fn, byt in (
                ('colfilter 7 taps', lambda: ll.axis_colfilter(X, h1o, axis=axis), 2 * n * e),
                ('coldfilt 10 taps', lambda: ll.axis_coldfilt(X, h0b, h0a, axis=axis), 1.5 * n * e),
                ('colifilt 10 taps', lambda: ll.axis_colifilt(X, h0b, h0a, axis=axis), 3 * n * e)):
            t = timeit(fn)
            print('%-8s axis %d %-18s %8.1f us  %5.2f TB/s algorithmic' % (dt.__name__, axis, name, t * 1e6, byt / t / 1e12))
