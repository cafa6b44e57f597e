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


WAVES = [('near_sym_a', 'qshift_a'), ('near_sym_a', 'qshift_a'), ('near_sym_b', 'qshift_b'), ('near_sym_b', 'qshift_d'),
         ('antonini', 'qshift_c'), ('legall', 'qshift_06'),
         # the marching PAIRS (march2d_pair.hpp / march2d_ipair.hpp): both directions for near_sym_a and legall
         ('near_sym_a', 'qshift_b'), ('near_sym_a', 'qshift_d'), ('legall', 'qshift_b'), ('legall', 'qshift_d')]
if os.environ.get('SOAK_PAIRS_ONLY'):
    WAVES = WAVES[-4:]


def run(X, nl, gm, batch, wave=WAVES[0]):
    t = Transform2d(*wave)
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
        wave = WAVES[int(rs.randint(len(WAVES)))]
        R = 4 * int(rs.randint((lo + 3) // 4, 400)); C = 4 * int(rs.randint((lo + 3) // 4, 400))
        if wave[0] in ('near_sym_b', 'antonini') and rs.uniform() < 0.5:
            R += 2                          # level 1 alone as a march (march2d_l1.hpp) needs even rows only
        if rs.uniform() < 0.3:
            C = 232 * int(rs.randint(1, 5)) + 4 * int(rs.randint(-2, 3))     # around the strip boundaries
            C = max(C, lo + (-lo) % 4)
        B = int(rs.choice([0, 0, 2, 5]))
        X = rs.standard_normal(((B, R, C) if B else (R, C))).astype(np.float32)
        gm = None if rs.uniform() < 0.5 else rs.uniform(0.2, 1.5, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
        os.environ['DTCWT_HIP_MARCH'] = '0'
        a = run(X, nl, gm, B, wave)
        os.environ['DTCWT_HIP_MARCH'] = '1'
        band = int(rs.choice([0, 8, 12, 24, 40, 100])) if wave[0] in ('near_sym_a', 'legall') else int(rs.choice([0, 20, 40, 60, 100]))
        if band:
            os.environ['DTCWT_HIP_MARCH_BAND'] = str(band)
        else:
            os.environ.pop('DTCWT_HIP_MARCH_BAND', None)
        os.environ['DTCWT_HIP_MARCH_PARTS'] = str(rs.choice([255 & ~128, 255 | 256]))       # k_inv21m or the same macro-steps as a pair (bit-identical)
        b = run(X, nl, gm, B, wave)
        os.environ.pop('DTCWT_HIP_MARCH_PARTS')
        errs = [rel(b[0], a[0])] + [rel(y, w) for y, w in zip(b[1], a[1])] + [rel(b[2], a[2])]
        worst = max(worst, max(errs))
        assert max(errs) < 2e-6, (wave, R, C, B, nl, band, errs)
        n += 1
    print('%d random 2-D transforms in %.0f s (near_sym_a / legall: fused levels 1 + 2, marching pairs with the 14- / 18-tap q-shift sets; near_sym_b / antonini: level 1 alone), marching launches vs '
          'tile programs: worst relative difference %.3g' % (n, time.time() - t0, worst))
    # ---- 3-D: level 1 as a marching pair of wavefronts (fused3d_march.hpp) against the tile program
    from dtcwt_amd.hip import Transform3d
    t1, n3, worst3 = time.time(), 0, 0.0
    while time.time() - t1 < secs / 3:
        n0, n1 = 2 * int(rs.randint(4, 40)), 2 * int(rs.randint(4, 40))
        n2 = 4 * int(rs.choice([rs.randint(4, 66), rs.randint(62, 68), rs.randint(120, 135)]))
        V = rs.standard_normal((n0, n1, n2)).astype(np.float32)
        t = Transform3d(biort=str(rs.choice(['near_sym_a', 'legall'])))
        os.environ['DTCWT_HIP_FWD3_MARCH'] = '0'
        a = t.forward(V, nlevels=1)
        os.environ['DTCWT_HIP_FWD3_MARCH'] = '1'
        os.environ['DTCWT_HIP_FWD3_CHUNK'] = str(rs.choice([8, 16, 64]))
        b = t.forward(V, nlevels=1)
        e = max(rel(np.asarray(b.lowpass), np.asarray(a.lowpass)), rel(np.asarray(b.highpasses[0]), np.asarray(a.highpasses[0])))
        worst3 = max(worst3, e)
        assert e < 2e-6, (n0, n1, n2, e)
        n3 += 1
    print('%d random 3-D level-1 transforms in %.0f s, marching pair vs tile program: worst relative difference %.3g' % (n3, time.time() - t1, worst3))


if __name__ == '__main__':
    main()
