"""256^3 float64 Transform3d (filter-by-filter levels) for a rocprofv3 --kernel-trace --stats run."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform3d
ctx = Context(0)
V = ctx.to_device(np.random.RandomState(2).standard_normal((256, 256, 256)))
t3 = Transform3d(ctx=ctx)
p = t3.forward(V, nlevels=3)
for _ in range(2):
    t3.forward(V, nlevels=3); t3.inverse(p, device_output=True)
ctx.device_sync()
t0 = time.perf_counter()
for _ in range(10):
    t3.forward(V, nlevels=3)
ctx.device_sync(); t1 = time.perf_counter()
for _ in range(10):
    t3.inverse(p, device_output=True)
ctx.device_sync(); t2 = time.perf_counter()
print('float64 256^3 nl=3: fwd %.1f us  inv %.1f us' % ((t1 - t0) * 1e5, (t2 - t1) * 1e5))
