"""C4 (256^3 float32, nlevels=3 forward) wall-clock, fused vs generic level 1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform3d
ctx = Context(0)
N = int(os.environ.get('N3D', '256'))
V = ctx.to_device(np.random.RandomState(2).standard_normal((N, N, N)).astype(np.float32))


def wall(fn, reps=10):
    for _ in range(3):
        fn()
    ctx.device_sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.device_sync(); ts.append(time.perf_counter() - t0)
    return min(ts), float(np.median(ts))


for fused in (True, False):
    t3 = Transform3d(ctx=ctx)
    t3.fused = fused
    for nl in (1, 2, 3):
        mn, md = wall(lambda: t3.forward(V, nlevels=nl))
        print('fused=%d nlevels=%d fwd min %.1f us median %.1f us  (%.0f Mvox/s)' % (fused, nl, mn * 1e6, md * 1e6, N ** 3 / mn / 1e6))
t3 = Transform3d(ctx=ctx)
p = t3.forward(V, nlevels=3)
mn, md = wall(lambda: t3.inverse(p, device_output=True))
print('inverse nlevels=3 min %.1f us median %.1f us' % (mn * 1e6, md * 1e6))
z = t3.inverse(p, device_output=True)
print('PR err', float(np.abs(z.get() - V.get()).max()))
