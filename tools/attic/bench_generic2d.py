import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
X32 = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)).astype(np.float32))
X64 = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)))
def timeit(fn, reps=5):
    fn(); ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.device_sync()
    return (time.perf_counter() - t0) / reps
for name, X, b, q in (('f32 fused', X32, 'near_sym_a', 'qshift_a'), ('f64 generic', X64, 'near_sym_a', 'qshift_a'),
                      ('f32 _bp generic', X32, 'near_sym_b_bp', 'qshift_b_bp'), ('f32 near_sym_b/qshift_d fused', X32, 'near_sym_b', 'qshift_d')):
    t = Transform2d(b, q, ctx=ctx)
    p = t.forward(X, nlevels=4)
    tf = timeit(lambda: t.forward(X, nlevels=4))
    ti = timeit(lambda: t.inverse(p, device_output=True)) if hasattr(t, 'inverse') else 0
    print('%-32s fwd %8.1f us  inv %8.1f us' % (name, tf * 1e6, ti * 1e6))
