"""Inverse-only C4 loop for rocprofv3 (kernel-trace) runs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform3d
ctx = Context(0)
N = int(os.environ.get('N3D', '256'))
V = ctx.to_device(np.random.RandomState(2).standard_normal((N, N, N)).astype(np.float32))
t3 = Transform3d(ctx=ctx)
p = t3.forward(V, nlevels=3)
ts = []
for _ in range(int(os.environ.get('REPS', '10'))):
    t0 = time.perf_counter(); z = t3.inverse(p, device_output=True); ctx.device_sync(); ts.append(time.perf_counter() - t0)
print('inv min %.1f us' % (min(ts) * 1e6))
print('PR err', float(np.abs(z.get() - V.get()).max()))
