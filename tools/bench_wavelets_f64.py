import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
X64 = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)))
X32 = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)).astype(np.float32))
def timeit(fn, reps=5):
    fn(); ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.device_sync()
    return (time.perf_counter() - t0) / reps
for b, q in (('near_sym_a', 'qshift_a'), ('near_sym_b', 'qshift_b'), ('near_sym_b', 'qshift_d'), ('antonini', 'qshift_06'), ('legall', 'qshift_c'), ('near_sym_a', 'qshift_32')):
    for name, X in (('f64', X64),):
        t = Transform2d(b, q, ctx=ctx)
        p = t.forward(X, nlevels=4)
        tf = timeit(lambda: t.forward(X, nlevels=4)); ti = timeit(lambda: t.inverse(p, device_output=True))
        print('%s %-11s %-10s fwd %7.1f us  inv %7.1f us' % (name, b, q, tf * 1e6, ti * 1e6))
