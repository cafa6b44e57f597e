"""Time dtcwt_hip_fwd3_level1 alone (256^3 float32 by default; N3D=<n> for n^3).  KNOCKS=a,b,... repeats the
measurement with DTCWT_HIP_F3_KNOCK set to each value -- only meaningful with an experiment build that reads it
(the phase knock-outs quoted in DESIGN.md section 4 were such builds, made with tools/build_variant.sh)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.coeffs import biort
from dtcwt_amd.hip import Context
from dtcwt_amd.hip import _lib
N = int(os.environ.get('N3D', '256'))
ctx = Context(0)
lib = _lib.lib()
pd = ctypes.POINTER(ctypes.c_double)
X = ctx.to_device(np.random.RandomState(2).standard_normal((N, N, N)).astype(np.float32))
LLL = ctx.empty((N, N, N), np.float32)
Yh = ctx.empty((N // 2, N // 2, N // 2, 28), np.complex64)
h0o, g0o, h1o, g1o = biort('near_sym_a')
h0 = np.ascontiguousarray(h0o, np.float64).ravel(); h1 = np.ascontiguousarray(h1o, np.float64).ravel()
def run(n):
    for _ in range(n):
        rc = lib.dtcwt_hip_fwd3_level1(ctx.handle, X.ptr, N, N, N, h0.ctypes.data_as(pd), h0.size, h1.ctypes.data_as(pd), h1.size, LLL.ptr, Yh.ptr)
        assert rc == 0, rc
for kn in os.environ.get('KNOCKS', '0').split(','):
    os.environ['DTCWT_HIP_F3_KNOCK'] = kn
    run(5); ctx.device_sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(20); ctx.device_sync(); best = min(best, (time.perf_counter() - t0) / 20)
    print('knock=%-3s fwd3_level1 %d^3: %.1f us' % (kn, N, best * 1e6), flush=True)
