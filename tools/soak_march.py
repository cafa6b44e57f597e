"""Randomised comparison of the one-launch levels (marching kernels, march2d.hpp) with the per-level tile programs:
random sizes (multiples of 4), batches, level counts, band heights, gain masks; every subband and the reconstruction.
The two paths sum in different orders, so the bound is 2e-6 of the subband's maximum, not equality.

    python tools/soak_march.py [seconds=120] [seed=0]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd.hip import Transform2d, Pyramid      # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run(X, nl, gm, batch):
    t = Transform2d()
    if batch:
        p = t.forward_channels(X, 'nhw', nlevels=nl)
        yl, ys = np.array(p.lowpass), [np.array(y) for y in p.highpasses]
        z = np.array(t.inverse_channels(p, 'nhw', gain_mask=gm))
    else:
        p = t.forward(X, nlevels=nl)
        yl, ys = np.array(p.lowpass), [np.array(y) for y in p.highpasses]
        z = np.array(t.inverse(Pyramid(yl, tuple(ys)), gm))
    return yl, ys, z


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < secs:
        nl = int(rs.randint(2, 5))
        lo = 40 * 2 ** (nl - 1)            # every level at least 40 samples (the fused plan's floor)
        R = 4 * int(rs.randint((lo + 3) // 4, 400)); C = 4 * int(rs.randint((lo + 3) // 4, 400))
        if rs.uniform() < 0.3:
            C = 232 * int(rs.randint(1, 5)) + 4 * int(rs.randint(-2, 3))     # around the strip boundaries
            C = max(C, lo + (-lo) % 4)
        B = int(rs.choice([0, 0, 2, 5]))
        X = rs.standard_normal(((B, R, C) if B else (R, C))).astype(np.float32)
        gm = None if rs.uniform() < 0.5 else rs.uniform(0.2, 1.5, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
        os.environ['DTCWT_HIP_MARCH'] = '0'
        a = run(X, nl, gm, B)
        os.environ['DTCWT_HIP_MARCH'] = '1'
        band = int(rs.choice([0, 8, 12, 24, 40, 100]))
        if band:
            os.environ['DTCWT_HIP_MARCH_BAND'] = str(band)
        else:
            os.environ.pop('DTCWT_HIP_MARCH_BAND', None)
        b = run(X, nl, gm, B)
        errs = [rel(b[0], a[0])] + [rel(y, w) for y, w in zip(b[1], a[1])] + [rel(b[2], a[2])]
        worst = max(worst, max(errs))
        assert max(errs) < 2e-6, (R, C, B, nl, band, errs)
        n += 1
    print('%d random transforms in %.0f s, marching launches vs tile programs: worst relative difference %.3g' % (n, time.time() - t0, worst))


if __name__ == '__main__':
    main()
