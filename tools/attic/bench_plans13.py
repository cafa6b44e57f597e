"""Latency of small 3-D / 1-D transforms with the native whole-transform plans (one library call per
transform) and with the level loop sequenced from Python (one call per level), GPU box:
python tools/bench_plans13.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd.hip import Context, Transform1d, Transform3d


def timeit(fn, ctx, reps=200, warm=20):
    for _ in range(warm):
        fn()
    ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.device_sync()
    return (time.perf_counter() - t0) / reps * 1e6


ctx = Context(0)
rs = np.random.RandomState(0)
for n in (32, 64, 128):
    V = ctx.to_device(rs.standard_normal((n, n, n)).astype(np.float32))
    row = []
    for planned in (True, False):
        t3 = Transform3d(ctx=ctx)
        if not planned:
            t3._plan = lambda shape, nlevels: None
        p = t3.forward(V, nlevels=3)
        tf = timeit(lambda: t3.forward(V, nlevels=3), ctx)
        ti = timeit(lambda: t3.inverse(p, device_output=True), ctx)
        row.append((tf, ti))
    print('3-D %3d^3 nlevels=3 f32: plan fwd %7.1f us inv %7.1f us | per-level calls fwd %7.1f us inv %7.1f us' % (
        n, row[0][0], row[0][1], row[1][0], row[1][1]))
for n, k in ((4096, 1), (65536, 1), (4096, 64)):
    x = ctx.to_device(rs.standard_normal((n, k)).astype(np.float32))
    row = []
    for planned in (True, False):
        os.environ['DTCWT_HIP_PLAN1D'] = '1' if planned else '0'
        t1 = Transform1d(ctx=ctx)
        p = t1.forward(x, nlevels=5)
        tf = timeit(lambda: t1.forward(x, nlevels=5), ctx)
        ti = timeit(lambda: t1.inverse(p, device_output=True), ctx)
        row.append((tf, ti))
    print('1-D %6d x %2d nlevels=5 f32: plan fwd %7.1f us inv %7.1f us | per-level calls fwd %7.1f us inv %7.1f us' % (
        n, k, row[0][0], row[0][1], row[1][0], row[1][1]))
