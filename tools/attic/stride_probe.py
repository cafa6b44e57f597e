import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
t = Transform2d(ctx=ctx)
for shape in ((4096, 4096), (4096, 4160), (4096, 4224), (4160, 4096), (4032, 4032), (8192, 2048), (2048, 8192)):
    X = ctx.to_device(np.random.RandomState(0).standard_normal(shape).astype(np.float32))
    p = t.forward(X, nlevels=4)
    def step():
        q = t.forward(X, nlevels=4); t.inverse(q, device_output=True)
    for _ in range(5): step()
    ctx.device_sync(); t0 = time.perf_counter()
    for _ in range(50): step()
    ctx.device_sync(); dt = (time.perf_counter() - t0) / 50
    print('%s fwd+inv %.1f us  %.0f Mpix/s' % (shape, dt * 1e6, shape[0] * shape[1] / dt / 1e6))
