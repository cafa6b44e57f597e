"""Single two-launch levels (dtcwt_hip_level2d_*) at a few sizes, for a rocprofv3 --kernel-trace
run: prints nothing, the per-dispatch durations are read from the trace in dispatch order."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Context
from dtcwt_amd.hip import lowlevel as ll
ctx = Context(0)
h0o, g0o, h1o, g1o = biort('near_sym_a')[:4]
h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = qshift('qshift_a')[:8]
CASES = [(np.float64, 2048, 2048), (np.float64, 2048, 2304), (np.float64, 4096, 4096),
         (np.float32, 2048, 2048), (np.float32, 4096, 4096)]
if __name__ == '__main__':
    if len(sys.argv) > 1:       # one case: 'f64 4096'
        CASES = [(np.float64 if sys.argv[1] == 'f64' else np.float32, int(sys.argv[2]), int(sys.argv[2]))]
    for dt, R, C in CASES:
        X = ctx.to_device(np.random.RandomState(0).standard_normal((1, R, C)).astype(dt))
        for _ in range(3):
            L0, Y0 = ll.level2d_forward(X, 0, (0, 0), (0, 0), h0o, h1o)
            L1, Y1 = ll.level2d_forward(X, 1, (0, 0), (0, 0), (h0b, h0a), (h1b, h1a))
            ll.level2d_inverse(L1, Y1, 1, np.ones(6), 0, 0, (g0b, g0a), (g1b, g1a))
            ll.level2d_inverse(L0, Y0, 0, np.ones(6), 0, 0, g0o, g1o)
        ctx.device_sync()
