import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
X = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)))
for b, q in (('near_sym_b', 'qshift_b'), ('antonini', 'qshift_06'), ('legall', 'qshift_c')):
    t = Transform2d(b, q, ctx=ctx)
    for _ in range(3):
        p = t.forward(X, nlevels=2)
        t.inverse(p, device_output=True)
ctx.device_sync()
