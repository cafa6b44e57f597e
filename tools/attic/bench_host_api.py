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
from dtcwt_amd.hip._lib import host_pool
for mb in (64, 256):        # page-locked destination (the pool's buffers): one DMA transfer, no staging pass
    d = ctx.to_device(np.random.RandomState(0).standard_normal(mb * (1 << 20) // 4).astype(np.float32))
    t = best(lambda: d.get()); print('D2H pinned   %4d MB: %6.2f ms  %5.1f GB/s (pooled page-locked buffer, pinned %d MB)' % (mb, t * 1e3, d.nbytes / t / 1e9, host_pool.pinned_bytes >> 20))
X = np.random.RandomState(1).standard_normal((4096, 4096)).astype(np.float32)
tr = Transform2d(ctx=ctx)
def fwd():
    p = tr.forward(X, nlevels=4)
    return p.lowpass, p.highpasses
t = best(fwd, 5); print('Transform2d.forward host->host 4096^2 nl=4: %.2f ms (%.0f Mpix/s)  [64 MiB up + 256 MiB down: %.2f ms at 57 GB/s]' % (t * 1e3, 16.78 / t, (64 + 256) * 1.048576 / 57))
tr.prefetch_host = False
t = best(fwd, 5); print('   the same with lazy blocking copies (DTCWT_HIP_PREFETCH_HOST=0): %.2f ms' % (t * 1e3))
tr.prefetch_host = True
# a stream of images (the reference's examples/register_video.py:125-156 shape): image k+1 goes up and is transformed
# while the subbands of image k come down (full-duplex link, copy stream)
Xs = [np.random.RandomState(10 + i).standard_normal((4096, 4096)).astype(np.float32) for i in range(4)]
def stream(n=8):
    prev = None
    for i in range(n):
        p = tr.forward(Xs[i % 4], nlevels=4)
        if prev is not None:
            _ = prev.lowpass, prev.highpasses
        prev = p
    _ = prev.lowpass, prev.highpasses
stream(4)
t0 = time.perf_counter(); stream(8); t = (time.perf_counter() - t0) / 8
print('stream of 4096^2 images, forward host->host, downloads overlapped with the next image: %.2f ms per image (%.0f Mpix/s)' % (t * 1e3, 16.78 / t))
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
