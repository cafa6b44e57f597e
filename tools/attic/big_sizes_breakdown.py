import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d, Transform3d
ctx = Context(0)
for N in (4096, 8192, 16384):
    Xd = ctx.to_device(np.random.RandomState(0).standard_normal((N, N)).astype(np.float32))
    t = Transform2d(ctx=ctx)
    for it in range(3):
        t0 = time.perf_counter(); p = t.forward(Xd, nlevels=5); ctx.device_sync(); tf = time.perf_counter() - t0
        t0 = time.perf_counter(); z = t.inverse(p, device_output=True); ctx.device_sync(); ti = time.perf_counter() - t0
        t0 = time.perf_counter(); del p, z; td = time.perf_counter() - t0
        print(N, it, 'fwd %.2f ms inv %.2f ms del %.2f ms' % (tf * 1e3, ti * 1e3, td * 1e3))
    del Xd
V = ctx.to_device(np.random.RandomState(0).standard_normal((640, 640, 640)).astype(np.float32))
t3 = Transform3d(ctx=ctx)
for it in range(3):
    t0 = time.perf_counter(); p = t3.forward(V, nlevels=3); ctx.device_sync(); tf = time.perf_counter() - t0
    t0 = time.perf_counter(); z = t3.inverse(p, device_output=True); ctx.device_sync(); ti = time.perf_counter() - t0
    del p, z
    print('3d', it, 'fwd %.2f ms inv %.2f ms' % (tf * 1e3, ti * 1e3))
