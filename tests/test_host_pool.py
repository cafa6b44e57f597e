"""The host-buffer pool behind DeviceArray.get(): a buffer is recycled only when no view of the
array that was handed out survives it."""
import gc

import numpy as np

from dtcwt_amd.hip import _lib


def test_recycles_only_unreferenced_buffers():
    hp = _lib._HostPool()
    assert hp._quiet_refs > 0
    a = hp.empty((512, 1024), np.float32)
    addr = a.ctypes.data
    a[:] = 3
    del a
    gc.collect()
    assert hp._idle == 512 * 1024 * 4
    b = hp.empty((1024, 512), np.float32)              # same byte size, other shape: reused
    assert b.ctypes.data == addr and hp._idle == 0
    for keep in (lambda x: x[2:5], lambda x: x.T, lambda x: x.reshape(-1), lambda x: x.view(np.uint8),
                 lambda x: memoryview(x)):
        arr = hp.empty((1024, 512), np.float32)
        alias = keep(arr)
        del arr
        gc.collect()
        assert hp._idle == 0, 'a buffer with a live alias was recycled'
        del alias
        gc.collect()
        hp.trim()
    small = hp.empty((16,), np.float64)                # below the pooling threshold: plain arrays
    assert small.base is None


def test_pool_limit_and_disable(monkeypatch):
    monkeypatch.setenv('DTCWT_HIP_HOST_POOL_MB', '1')
    hp = _lib._HostPool()
    a = hp.empty((1 << 20,), np.float32)               # 4 MiB > 1 MiB limit: never kept
    del a
    gc.collect()
    assert hp._idle == 0
    monkeypatch.setenv('DTCWT_HIP_HOST_POOL_MB', '0')
    hp = _lib._HostPool()
    assert hp.empty((1 << 20,), np.float32).base is None
