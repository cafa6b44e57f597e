"""float64 2-D transform (generic kernels) for a rocprofv3 --kernel-trace --stats run."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
X = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)))
t = Transform2d(ctx=ctx)
for _ in range(10):
    p = t.forward(X, nlevels=4)
    Z = t.inverse(p, device_output=True)
ctx.device_sync()
