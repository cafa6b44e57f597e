"""Do small latency-bound transforms on one stream hide behind bandwidth-bound ones on another?
big = 4096^2 nlevels=2 (levels 1-2 of the bench image), small = 1024^2 nlevels=2 (its levels 3-4)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, DeviceArray, Transform2d

def setup(ctx, n, nl):
    t = Transform2d(ctx=ctx)
    plan = t.plan(1, n, n, nl)
    X = ctx.to_device(np.random.RandomState(1).standard_normal((1, n, n)).astype(np.float32))
    Yl = DeviceArray(ctx, (1,) + plan.low, np.float32)
    Yh = [DeviceArray(ctx, (1,) + plan.high[l] + (6,), np.complex64) for l in range(nl)]
    Z = DeviceArray(ctx, (1,) + plan.ext, np.float32)
    def step():
        plan.forward_into(X, Yl, Yh); plan.inverse_into(Yl, Yh, None, Z)
    return step

c0 = Context(0)
if os.environ.get('PRIO'):
    import torch                      # the process shares torch's HIP runtime (dtcwt_amd/hip/_lib.py)
    ts = torch.cuda.Stream(priority=-1)
    c1 = Context(0, stream=ts.cuda_stream)
    print('small work on a high-priority stream')
else:
    c1 = Context(0)
big = setup(c0, 4096, 2); small = setup(c1, 1024, 2)
def run(fs, n=300):
    for f in fs:
        for _ in range(20): f()
    c0.device_sync()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fs: f()
    c0.sync(); c1.sync(); c0.device_sync()
    return (time.perf_counter() - t0) / n * 1e6
for rep in range(2):
    print('big alone   %.1f us/step' % run([big]))
    print('small alone %.1f us/step' % run([small]))
    print('both        %.1f us/step' % run([big, small]), flush=True)
