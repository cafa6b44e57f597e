"""Transform1d of the hip backend on device-resident signals: one long vector and a batch of columns."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform1d
ctx = Context(0)
def timeit(fn, reps=10):
    fn(); ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.device_sync()
    return (time.perf_counter() - t0) / reps
for dt in (np.float32, np.float64):
    for n, k, nl in ((1 << 24, 1, 6), (1 << 16, 256, 6), (1 << 12, 4096, 4)):
        X = ctx.to_device(np.random.RandomState(0).standard_normal((n, k)).astype(dt))
        t = Transform1d(ctx=ctx)
        p = t.forward(X, nlevels=nl)
        tf = timeit(lambda: t.forward(X, nlevels=nl))
        ti = timeit(lambda: t.inverse(p, device_output=True))
        b = n * k * X.dtype.itemsize
        print('%-8s n=%-9d k=%-5d nl=%d  fwd %8.1f us (%.2f TB/s @3x)  inv %8.1f us' % (dt.__name__, n, k, nl, tf * 1e6, 3 * b / tf / 1e12, ti * 1e6))
