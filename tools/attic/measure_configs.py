"""Timings of the other BASELINE configs on one MI355X (development tool; the bench metric is
configs[1] only).  Run on the GPU box: python tools/measure_configs.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dtcwt_amd
from dtcwt_amd.hip import Context, DeviceArray, Transform2d, Transform3d


def timeit(fn, ctx, reps=10, warm=2):
    for _ in range(warm):
        fn()
    ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.device_sync()
    return (time.perf_counter() - t0) / reps


def plan_case(ctx, B, R, C, nl, label):
    t = Transform2d(ctx=ctx)
    plan = t.plan(B, R, C, nl)
    rs = np.random.RandomState(1)
    X = ctx.to_device(rs.standard_normal((B, R, C)).astype(np.float32))
    Yl = DeviceArray(ctx, (B,) + plan.low, np.float32)
    Yh = [DeviceArray(ctx, (B,) + plan.high[l] + (6,), np.complex64) for l in range(nl)]
    Z = DeviceArray(ctx, (B,) + plan.ext, np.float32)
    tf = timeit(lambda: plan.forward_into(X, Yl, Yh), ctx)
    tb = timeit(lambda: (plan.forward_into(X, Yl, Yh), plan.inverse_into(Yl, Yh, None, Z)), ctx)
    px = B * R * C
    print('%-34s fwd %8.3f ms %9.0f Mpix/s (%.2f of 8TB/s @20B/px) | fwd+inv %8.3f ms %9.0f Mpix/s (%.2f @40B/px)' % (
        label, tf * 1e3, px / tf / 1e6, 20 * px / tf / 8e12, tb * 1e3, px / tb / 1e6, 40 * px / tb / 8e12))


ctx = Context(0)
plan_case(ctx, 1, 4096, 4096, 4, 'C2  1 x 4096^2 nl=4')
plan_case(ctx, 64, 1024, 1024, 5, 'C3 64 x 1024^2 nl=5')
plan_case(ctx, 64, 2048, 2048, 4, 'C5 64 x 2048^2 nl=4 (1/8 of 512)')
plan_case(ctx, 8, 4096, 4096, 4, '    8 x 4096^2 nl=4')
# C4: 3-D 256^3 nlevels=3 (fused level kernels; the generic axis passes for comparison)
rs = np.random.RandomState(2)
V = ctx.to_device(rs.standard_normal((256, 256, 256)).astype(np.float32))
for fused in (True, False):
    t3 = Transform3d(ctx=ctx)
    t3.fused = fused
    tf = timeit(lambda: t3.forward(V, nlevels=3), ctx, reps=10, warm=3)
    p3 = t3.forward(V, nlevels=3)
    ti = timeit(lambda: t3.inverse(p3, device_output=True), ctx, reps=10, warm=3)
    print('%-34s fwd %8.3f ms %9.0f Mvox/s (%.3f of 8TB/s @36B/vox) | inv %8.3f ms %9.0f Mvox/s' % (
        'C4 256^3 nl=3 (%s)' % ('fused levels' if fused else 'generic axis passes'), tf * 1e3, 256 ** 3 / tf / 1e6,
        36 * 256 ** 3 / tf / 8e12, ti * 1e3, 256 ** 3 / ti / 1e6))
