import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform3d
for b, q in (('near_sym_b', 'qshift_b'), ('near_sym_a', 'qshift_a')):
    ctx = Context(0)
    t = Transform3d(b, q, ctx=ctx)
    X = ctx.to_device(np.random.RandomState(0).standard_normal((256, 256, 256)).astype(np.float32))
    for _ in range(5):
        p = t.forward(X, nlevels=3)
    ctx.device_sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            p = t.forward(X, nlevels=3)
        t1 = time.perf_counter()
        ctx.device_sync()
        t2 = time.perf_counter()
        print(b, q, 'forward: host issue %.1f us per call, total %.1f us per call' % ((t1 - t0) / 40 * 1e6, (t2 - t0) / 40 * 1e6))
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(40):
            z = t.inverse(p, device_output=True)
        t1 = time.perf_counter()
        ctx.device_sync()
        t2 = time.perf_counter()
        print(b, q, 'inverse: host issue %.1f us per call, total %.1f us per call' % ((t1 - t0) / 40 * 1e6, (t2 - t0) / 40 * 1e6))
