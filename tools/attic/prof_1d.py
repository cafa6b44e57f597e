import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform1d
ctx = Context(0)
X = ctx.to_device(np.random.RandomState(0).standard_normal((1 << 24, 1)).astype(np.float32))
t = Transform1d(ctx=ctx)
for _ in range(5):
    p = t.forward(X, nlevels=6)
    z = t.inverse(p, device_output=True)
ctx.device_sync()
