"""forward + inverse of small single images through the 2-D plan (us per pair of calls): the regime where every
level is a latency chain of a few workgroups."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, DeviceArray, Transform2d
ctx = Context(0)
for n, nl in ((256, 3), (512, 4), (1024, 4), (2048, 4)):
    t = Transform2d(ctx=ctx)
    plan = t.plan(1, n, n, nl)
    X = ctx.to_device(np.random.RandomState(1).standard_normal((1, n, n)).astype(np.float32))
    Yl = DeviceArray(ctx, (1,) + plan.low, np.float32)
    Yh = [DeviceArray(ctx, (1,) + plan.high[l] + (6,), np.complex64) for l in range(nl)]
    Z = DeviceArray(ctx, (1,) + plan.ext, np.float32)
    def step():
        plan.forward_into(X, Yl, Yh); plan.inverse_into(Yl, Yh, None, Z)
    for _ in range(50): step()
    ctx.device_sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200): step()
        ctx.device_sync()
        best = min(best, (time.perf_counter() - t0) / 200)
    print('%4d^2 nlevels=%d: %.1f us fwd+inv' % (n, nl, best * 1e6), flush=True)
