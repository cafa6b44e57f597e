"""Large-size sanity and throughput (development tool, GPU box): 16384^2 2-D and 640^3 3-D."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d, Transform3d, DeviceArray
ctx = Context(0)
N = 16384
rs = np.random.RandomState(0)
X = rs.standard_normal((N, N)).astype(np.float32)
Xd = ctx.to_device(X)
t = Transform2d(ctx=ctx)
for _ in range(2):                      # warm the buffer cache; release before re-allocating
    p = t.forward(Xd, nlevels=5)
    z = t.inverse(p, device_output=True)
    ctx.device_sync()
    del p, z
t0 = time.perf_counter(); p = t.forward(Xd, nlevels=5); z = t.inverse(p, device_output=True); ctx.device_sync(); dt = time.perf_counter() - t0
zz = z.get()
print('16384^2 nl=5 fwd+inv %.2f ms  %.0f Mpix/s  PR max err %.2e' % (dt * 1e3, N * N / dt / 1e6, np.abs(zz - X).max()))
# spot-check a corner block of the finest level against the oracle on a crop (interior of the crop)
from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
crop = X[-600:, -700:].astype(np.float64)
want = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(crop, nlevels=1)
got = p.hip_highpasses[0].get()[-200:, -250:]
print('far-corner Yh[0] block rel err %.2e' % (np.abs(got - want.highpasses[0][-200:, -250:]).max() / np.abs(want.highpasses[0]).max()))
del p, z, Xd
V = rs.standard_normal((640, 640, 640)).astype(np.float32)
Vd = ctx.to_device(V)
t3 = Transform3d(ctx=ctx)
for _ in range(2):
    p3 = t3.forward(Vd, nlevels=3)
    z3 = t3.inverse(p3, device_output=True)
    ctx.device_sync()
    del p3, z3
t0 = time.perf_counter(); p3 = t3.forward(Vd, nlevels=3); ctx.device_sync(); tf = time.perf_counter() - t0
t0 = time.perf_counter(); z3 = t3.inverse(p3, device_output=True); ctx.device_sync(); ti = time.perf_counter() - t0
print('640^3 nl=3 fwd %.2f ms (%.0f Mvox/s) inv %.2f ms  PR max err %.2e' % (tf * 1e3, 640 ** 3 / tf / 1e6, ti * 1e3, np.abs(z3.get() - V).max()))
