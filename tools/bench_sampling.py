"""Timings of the re-sampling kernels on one MI355X, with the NumPy oracle
timed beside them on a bounded sample.  Run on the GPU box: python tools/bench_sampling.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd import sampling
from dtcwt_amd.hip import Context
from oracle import sampling_oracle as so

ctx = Context(0)
rs = np.random.RandomState(4)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.device_sync()
    return (time.perf_counter() - t0) / reps


def cpu(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


lo = rs.standard_normal((2048, 2048)).astype(np.float32)
hi = (rs.standard_normal((1024, 1024, 6)) + 1j * rs.standard_normal((1024, 1024, 6))).astype(np.complex64)
dlo, dhi = ctx.to_device(lo), ctx.to_device(hi)
for m in ('nearest', 'bilinear', 'lanczos'):
    t = timeit(lambda: sampling.rescale(dlo, (4096, 4096), m, device_output=True))
    byt = lo.nbytes + 4096 * 4096 * 4
    tc = cpu(lambda: so.rescale(lo[:512, :512], (1024, 1024), m)) * 16
    print('rescale 2048^2 -> 4096^2 f32 %-8s %8.1f us  %7.0f Mpix/s out  %5.2f TB/s algorithmic | oracle (1/16 sample, scaled) %6.2f s  x%.0f'
          % (m, t * 1e6, 4096 * 4096 / t / 1e6, byt / t / 1e12, tc, tc / t))
    t = timeit(lambda: sampling.upsample(dlo, m, device_output=True))
    print('upsample 2048^2 x2 f32       %-8s %8.1f us  %7.0f Mpix/s out  %5.2f TB/s algorithmic' % (m, t * 1e6, 4096 * 4096 / t / 1e6, byt / t / 1e12))
    t = timeit(lambda: sampling.rescale_highpass(dhi, (2048, 2048), m, device_output=True))
    byt = hi.nbytes + 2048 * 2048 * 48
    tc = cpu(lambda: so.rescale_highpass(hi[:256, :256], (512, 512), m)) * 16
    print('rescale_highpass 1024^2x6 -> 2048^2x6 c64 %-8s %8.1f us  %5.2f TB/s algorithmic | oracle (scaled) %6.2f s  x%.0f'
          % (m, t * 1e6, byt / t / 1e12, tc, tc / t))
