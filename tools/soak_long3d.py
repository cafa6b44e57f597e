"""Randomised comparison of the two-launch level 1 of the 3-D transforms for the 13 / 19-tap filters (near_sym_b:
fused3d_long.hpp -- k_fwd1m / k_inv1m with plane volumes, k_fwd3l_axis0, k_inv3l_axis0) with the axis-by-axis generic
kernels: random volumes (20 .. 120 slices, rows of 40 .. 140, 40 .. 320 columns in fours: one strip with halo lanes, one
without, two strips), one to three levels, both ext_modes, the three q-shift lengths with a tile program, random chunk
lengths of both axis-0 marches; every subband and the reconstruction.  The two paths sum in different orders, so the bound
is 2e-6 of the subband's maximum, not equality.

    python tools/soak_long3d.py [seconds=120] [seed=0]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd.hip import Transform3d      # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t_end = time.time() + secs
    n, worst_f, worst_i, worst_pr = 0, 0.0, 0.0, 0.0
    while time.time() < t_end:
        ext = int(rs.choice([4, 8]))
        mult = 2 if ext == 4 else 4
        n0 = int(mult * rs.randint(20 // mult, 120 // mult + 1))
        n1 = int(mult * rs.randint(40 // mult, 140 // mult + 1))
        n2 = int(4 * rs.randint(10, 81))
        qn = ['qshift_b', 'qshift_a', 'qshift_d'][rs.randint(3)]
        nl = int(rs.randint(1, 4))
        for k, lo, hi in (('DTCWT_HIP_LONG3D_CHUNK', 10, 60), ('DTCWT_HIP_LONG3D_ICHUNK', 4, 40)):
            if rs.rand() < 0.5:
                os.environ[k] = str(int(rs.randint(lo, hi)))
            else:
                os.environ.pop(k, None)
        # round 6: the forward's default cut is axis 0 first + k_fwd3l_slices; a quarter of the cases keep the round-5 cut ('2'),
        # and the band height of the new launch is random in half of the others
        os.environ['DTCWT_HIP_LONG3D'] = '2' if rs.rand() < 0.25 else '1'
        if rs.rand() < 0.5:
            os.environ['DTCWT_HIP_LONG3D_BAND'] = str(int(20 * rs.randint(1, 6)))
        else:
            os.environ.pop('DTCWT_HIP_LONG3D_BAND', None)
        if rs.rand() < 0.3:
            os.environ['DTCWT_HIP_LONG3D_EDGE'] = str(int(rs.randint(2)))
        else:
            os.environ.pop('DTCWT_HIP_LONG3D_EDGE', None)
        X = rs.standard_normal((n0, n1, n2)).astype(np.float32)
        t, g = Transform3d('near_sym_b', qn, ext_mode=ext), Transform3d('near_sym_b', qn, ext_mode=ext)
        g.fused = False
        p, q = t.forward(X, nlevels=nl), g.forward(X, nlevels=nl)
        ef = max([rel(p.lowpass, q.lowpass)] + [rel(a, b) for a, b in zip(p.highpasses, q.highpasses)])
        z, zg = t.inverse(p), g.inverse(p)
        ei, epr = rel(z, zg), rel(z, X)
        assert ef < 2e-6 and ei < 5e-6 and epr < 2e-5, ((n0, n1, n2), qn, ext, nl, ef, ei, epr, dict((k, v) for k, v in os.environ.items() if 'LONG3D' in k))
        worst_f, worst_i, worst_pr = max(worst_f, ef), max(worst_i, ei), max(worst_pr, epr)
        n += 1
    print('soak_long3d: %d random volumes in %.0f s; worst forward %.2e, inverse vs generic %.2e, PR %.2e' % (n, secs, worst_f, worst_i, worst_pr))


if __name__ == '__main__':
    main()
