"""Randomised soak of the public low-level filters on the device (filters.hip: the marching kernels down a strided axis, the LDS-row kernels
along the contiguous one, the one-group-per-thread fall-backs) against the oracle: random shapes (2-D and 3-D arrays, any axis), tap counts
inside and outside the buckets, both filter phases, edge padding, crops, accumulation, float32 / float64.

    python tools/soak_lowlevel.py [seconds=60] [seed=0]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd.hip import default_context           # noqa: E402
from dtcwt_amd.hip import lowlevel as ll             # noqa: E402
from oracle import dtcwt_oracle as o                 # noqa: E402


def ref(fn, X, axis, pad, crop, *taps):
    Xm = np.moveaxis(X, axis, 0)
    shp = Xm.shape
    X2 = Xm.reshape(shp[0], -1)
    X2 = np.pad(X2, (tuple(pad), (0, 0)), mode='edge')
    Y = fn(np.ascontiguousarray(X2), *taps)
    Y = Y[crop[0]:Y.shape[0] - crop[1]]
    return np.moveaxis(Y.reshape((Y.shape[0],) + shp[1:]), 0, axis)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = default_context()
    t_end = time.time() + secs
    n, worst = 0, {np.float32: 0.0, np.float64: 0.0}
    while time.time() < t_end:
        nd = int(rs.choice([2, 2, 3]))
        axis = int(rs.randint(nd))
        shape = [int(rs.choice([1, 3, 8, 17, 64, 100, 130])) for _ in range(nd)]
        kind = int(rs.randint(3))
        L = int(rs.choice([8, 12, 40, 100, 256, 260, 516, 1024, 1100, 2052]))
        dt = rs.choice([np.float32, np.float64])
        pad = (int(rs.randint(3)), int(rs.randint(3))) if rs.rand() < 0.4 else (0, 0)
        if kind == 1:                     # coldfilt: padded length in fours
            while (L + pad[0] + pad[1]) % 4:
                L += 1
        elif kind == 2 and (L + pad[0] + pad[1]) % 2:
            L += 1
        shape[axis] = L
        X = rs.standard_normal(shape).astype(dt)
        d = ctx.to_device(X)
        crop = (int(rs.randint(5)), int(rs.randint(5))) if rs.rand() < 0.4 else (0, 0)
        if rs.rand() < 0.3:
            crop = (4 * (crop[0] // 2), 4 * (crop[1] // 2))
        if kind == 0:
            m = int(rs.choice([1, 2, 3, 5, 7, 8, 9, 13, 19, 20, 21, 24]))
            taps = (rs.standard_normal(m),)
            fd, fo = ll.axis_colfilter, o.colfilter
        else:
            m = int(rs.choice([2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 32]))
            sgn = rs.choice([-1.0, 1.0])
            taps = (rs.standard_normal(m), sgn * rs.standard_normal(m))
            fd, fo = (ll.axis_coldfilt, o.coldfilt) if kind == 1 else (ll.axis_colifilt, o.colifilt)
        Lp = L + pad[0] + pad[1]
        nout = (Lp if m % 2 else Lp + 1) if kind == 0 else (Lp // 2 if kind == 1 else 2 * Lp)
        if nout - crop[0] - crop[1] <= 0:
            continue
        want = ref(fo, X.astype(np.float64), axis, pad, crop, *taps)
        if rs.rand() < 0.25:
            base = rs.standard_normal(want.shape).astype(dt)
            out = ctx.to_device(base)
            fd(d, *taps, axis=axis, out=out, accumulate=True, pad=pad, crop=crop)
            got, want = out.get(), want + base
        else:
            got = fd(d, *taps, axis=axis, pad=pad, crop=crop).get()
        assert got.shape == want.shape, (got.shape, want.shape, kind, shape, axis, pad, crop, m)
        e = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
        worst[dt] = max(worst[dt], e)
        assert e < (2e-6 if dt == np.float32 else 1e-12), (e, kind, shape, axis, pad, crop, m, dt)
        n += 1
    print('soak_lowlevel: %d random filter calls in %.0f s; worst relative error float32 %.3g, float64 %.3g' % (n, secs, worst[np.float32], worst[np.float64]))


if __name__ == '__main__':
    main()
