"""One small transform at a time: stream launches against hipGraph replay, and with a sync after every transform
(256^2 / 512^2 nlevels=3: 30 us, one ~5 us launch per level and direction whatever the size; replay does not shorten it).

    python tools/kbench/latency_small.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import dtcwt_amd.hip
from dtcwt_amd.hip import Context, DeviceArray
ctx = Context(0); t = dtcwt_amd.hip.Transform2d(ctx=ctx); rs = np.random.RandomState(1)
print('one transform at a time, us per forward + inverse: stream launches / hipGraph replay')
for n, nl in ((256, 3), (512, 3), (512, 4), (1024, 4), (2048, 4)):
    pl = t.plan(1, n, n, nl)
    X = ctx.to_device(rs.standard_normal((1, n, n)).astype(np.float32))
    Yl = DeviceArray(ctx, (1,) + pl.low, np.float32); Yh = [DeviceArray(ctx, (1,) + pl.high[l] + (6,), np.complex64) for l in range(nl)]
    Z = DeviceArray(ctx, (1,) + pl.ext, np.float32)
    g = pl.capture(X, Yl, Yh, Z)
    def a():
        pl.forward_into(X, Yl, Yh); pl.inverse_into(Yl, Yh, None, Z)
    res = []
    for f in (a, g.launch):
        best = 1e9
        for rep in range(3):
            for _ in range(50): f()
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(500): f()
            ctx.sync(); best = min(best, (time.perf_counter() - t0) / 500 * 1e6)
        res.append(best)
    # latency of ONE call incl. sync
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(200): a(); ctx.sync()
    lat = (time.perf_counter() - t0) / 200 * 1e6
    print('%4d^2 nl=%d  %7.1f %7.1f   with a sync after each: %7.1f' % (n, nl, res[0], res[1], lat))
