import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
X = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)).astype(np.float32))
t = Transform2d(ctx=ctx)
for _ in range(10):
    p = t.forward(X, nlevels=4); z = t.inverse(p, device_output=True)
ctx.device_sync()
hf = hi = 0.0
t0 = time.perf_counter()
N = 200
for _ in range(N):
    a = time.perf_counter(); p = t.forward(X, nlevels=4); b = time.perf_counter(); z = t.inverse(p, device_output=True); c = time.perf_counter()
    hf += b - a; hi += c - b
ctx.device_sync(); t1 = time.perf_counter()
print('wall per fwd+inv %.1f us; host time in forward() %.1f us, in inverse() %.1f us' % ((t1 - t0) / N * 1e6, hf / N * 1e6, hi / N * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    p = t.forward(X, nlevels=4); z = t.inverse(p, device_output=True)
pr.disable(); ctx.device_sync()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
