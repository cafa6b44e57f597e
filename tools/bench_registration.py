"""Timing of estimatereg on one MI355X in the set-up of the reference's
scripts/benchmark_registration.py (two frames, Transform2d nlevels=6, estimatereg of the two
pyramids), with the NumPy oracle timed beside it.  Synthetic frames (the reference's tennis
frames are 288 x 352; sizes here: that and 1024^2).  Run on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd import registration as reg
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Context, Transform2d
from oracle import dtcwt_oracle as o
from oracle import registration_oracle as ro

ctx = Context(0)


def frames(h, w, seed=7):
    yy, xx = np.mgrid[0:h, 0:w]
    yy, xx = yy / float(h), xx / float(w)
    x2, y2 = xx * 1.01 + 0.012, yy * 0.994 - 0.008
    out = []
    for gx, gy in ((xx, yy), (x2, y2)):
        rs = np.random.RandomState(seed)
        im = np.zeros((h, w))
        for _ in range(24):
            fx, fy, ph = rs.uniform(1, 9), rs.uniform(1, 9), rs.uniform(0, 6.28)
            im += rs.uniform(0.2, 1) * np.cos(6.283 * (fx * gx + fy * gy) + ph)
        out.append(im.astype(np.float32))
    return out


for (h, w) in ((288, 352), (1024, 1024)):
    f1, f2 = frames(h, w)
    t = Transform2d(ctx=ctx)
    t1, t2 = t.forward(f1, nlevels=6), t.forward(f2, nlevels=6)
    for _ in range(3):
        reg.estimatereg(t1, t2, device_output=True)
    ctx.device_sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        av = reg.estimatereg(t1, t2, device_output=True)
    ctx.device_sync()
    tg = (time.perf_counter() - t0) / reps
    to = o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    o1, o2 = to.forward(f1, nlevels=6), to.forward(f2, nlevels=6)
    c0 = time.perf_counter()
    want = ro.estimatereg(o1, o2)
    tc = time.perf_counter() - c0
    err = np.abs(av.get() - want).max() / np.abs(want).max()
    print('estimatereg %4dx%-4d f32 nlevels=6: hip %8.1f us/iteration | oracle (1 core) %7.3f s  x%.0f | rel diff %.1e'
          % (h, w, tg * 1e6, tc, tc / tg, err))
