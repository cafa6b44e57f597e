"""Experiment: shares of the compute units x streams per share (profiles/r04/ab_partition_mix.txt).

    python tools/kbench/mix_partitions.py 4q|2hx2|4qx2|2h|2hx3|4qx2h1
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import dtcwt_amd.hip
from dtcwt_amd.hip import Context, DeviceArray

def rig(parts, per, hint):
    ctxs = [Context(0, partition=(p, parts)) for p in range(parts) for _ in range(per)]
    t2s = [dtcwt_amd.hip.Transform2d(ctx=c) for c in ctxs]
    plans = [t.plan(1, 4096, 4096, 4) for t in t2s]
    for pl in plans: pl.set_concurrency(hint)
    rs = np.random.RandomState(5)
    S = len(ctxs); nsets = 8 if 8 % S == 0 else S * 2
    sets = []
    for k in range(nsets):
        c = ctxs[k % S]; pl = plans[0]
        sets.append((c.to_device(rs.standard_normal((1, 4096, 4096)).astype(np.float32)), DeviceArray(c, (1,) + pl.low, np.float32),
                     [DeviceArray(c, (1,) + pl.high[l] + (6,), np.complex64) for l in range(4)], DeviceArray(c, (1,) + pl.ext, np.float32)))
    def run(n):
        for k in range(n):
            X, Yl, Yh, Z = sets[k % nsets]; pl = plans[k % S]
            pl.forward_into(X, Yl, Yh); pl.inverse_into(Yl, Yh, None, Z)
    return ctxs, run

which = sys.argv[1]
cfg = {'4q': (4, 1, 1), '2hx2': (2, 2, 2), '4qx2': (4, 2, 2), '2h': (2, 1, 1), '2hx3': (2, 3, 3), '4qx2h1': (4, 2, 1)}[which]
ctxs, run = rig(*cfg)
t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end: run(16)
ctxs[0].device_sync()
res = []
for rep in range(4):
    run(16); ctxs[0].device_sync()
    t0 = time.perf_counter(); run(200); ctxs[0].device_sync()
    res.append((time.perf_counter() - t0) / 200 * 1e3)
print(which, cfg, ' '.join('%.4f' % r for r in res))
