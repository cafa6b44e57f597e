import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform3d
ctx = Context(0)
V = ctx.to_device(np.random.RandomState(2).standard_normal((256, 256, 256)).astype(np.float32))
t3 = Transform3d(ctx=ctx)
for _ in range(3):
    p = t3.forward(V, nlevels=3)
ctx.device_sync()
t0 = time.perf_counter(); p = t3.forward(V, nlevels=3); ctx.device_sync(); print('fwd wall', time.perf_counter() - t0)
t0 = time.perf_counter(); z = t3.inverse(p, device_output=True); ctx.device_sync(); print('inv wall', time.perf_counter() - t0)
t0 = time.perf_counter(); z = t3.inverse(p, device_output=True); ctx.device_sync(); print('inv wall (warm)', time.perf_counter() - t0)
import numpy as np
print('PR err', float(np.abs(z.get() - V.get()).max()))
