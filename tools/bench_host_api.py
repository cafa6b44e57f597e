"""The drop-in (NumPy in, NumPy out) path: PCIe-inclusive timings of Transform2d at 4096^2 and the raw
host<->device copy rates behind them."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
def best(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)
for mb in (64, 256):
    h = np.random.RandomState(0).standard_normal(mb * (1 << 20) // 4).astype(np.float32)
    d = ctx.to_device(h)
    t = best(lambda: d.set(h)); print('H2D pageable %4d MB: %6.2f ms  %5.1f GB/s' % (mb, t * 1e3, h.nbytes / t / 1e9))
    t = best(lambda: d.get()); print('D2H pageable %4d MB: %6.2f ms  %5.1f GB/s (incl. np.empty)' % (mb, t * 1e3, h.nbytes / t / 1e9))
    out = np.empty_like(h)
    from dtcwt_amd.hip._lib import check, _vp
    t = best(lambda: check(ctx._lib.dtcwt_hip_memcpy_d2h(ctx.handle, out.ctypes.data_as(_vp), d.ptr, d.nbytes)))
    print('D2H pageable %4d MB: %6.2f ms  %5.1f GB/s (into a touched buffer)' % (mb, t * 1e3, h.nbytes / t / 1e9))
X = np.random.RandomState(1).standard_normal((4096, 4096)).astype(np.float32)
tr = Transform2d(ctx=ctx)
def fwd():
    p = tr.forward(X, nlevels=4)
    return p.lowpass, p.highpasses
t = best(fwd, 3); print('Transform2d.forward host->host 4096^2 nl=4: %.1f ms (%.0f Mpix/s)' % (t * 1e3, 16.78 / t))
p = tr.forward(X, nlevels=4); lo, hi = p.lowpass, p.highpasses
from dtcwt_amd.hip import Pyramid
t = best(lambda: tr.inverse(Pyramid(lo, hi)), 3); print('Transform2d.inverse host->host 4096^2 nl=4: %.1f ms (%.0f Mpix/s)' % (t * 1e3, 16.78 / t))
X8 = (np.random.RandomState(2).uniform(size=(4096, 4096)) * 255).astype(np.uint8)
def fwd8(widen_on_host):
    p = tr.forward(X8.astype(np.float64) if widen_on_host else X8, nlevels=4)
    ctx.device_sync()
    return p
t = best(lambda: fwd8(True), 3); print('uint8 4096^2 -> device pyramid, widened to float64 on the host:   %.1f ms' % (t * 1e3))
t = best(lambda: fwd8(False), 3); print('uint8 4096^2 -> device pyramid, widened on the device:            %.1f ms' % (t * 1e3))
