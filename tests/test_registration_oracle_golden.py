"""The registration oracle (oracle/registration_oracle.py) against vectors generated from the
reference's dtcwt.registration (tests/golden/registration.npz, make_golden_registration.py)."""
import os

import numpy as np

from oracle import dtcwt_oracle as o
from oracle import registration_oracle as ro
from dtcwt_amd.coeffs import biort, qshift

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    return np.load(os.path.join(HERE, 'golden', 'registration.npz'))


def pyramids(g, dtype=np.float64, nlevels=5):
    t = o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    return t.forward(g['im1'].astype(dtype), nlevels=nlevels), t.forward(g['im2'].astype(dtype), nlevels=nlevels)


def rel(a, b):
    return np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)).max() / max(np.abs(b).max(), 1e-30)


def test_oracle_matches_reference_vectors():
    g = golden()
    p1, p2 = pyramids(g)
    for l, q in zip((2, 4), ro.qtildematrices(p1, p2, [2, 4])):
        assert q.shape == g['qtilde/%d' % l].shape and rel(q, g['qtilde/%d' % l]) < 1e-9
    assert rel(ro.solvetransform(g['solve_in']), g['solve_out']) < 1e-12
    av = ro.estimatereg(p1, p2)
    assert av.shape == g['estimatereg'].shape and rel(av, g['estimatereg']) < 1e-6
    assert rel(ro.estimatereg(p1, p2, regshape=(5, 7), levels=[[4, 3], [3, 2]]), g['estimatereg_custom']) < 1e-6
    vx, vy = ro.velocityfield(g['estimatereg'], (32, 32), 'bilinear')
    assert rel(vx, g['velocity_x']) < 1e-12 and rel(vy, g['velocity_y']) < 1e-12
    assert rel(ro.warp(g['im1'], g['estimatereg'], 'bilinear'), g['warp']) < 1e-12
    assert rel(ro.warphighpass(p1.highpasses[2], g['estimatereg'], 'bilinear'), g['warphighpass']) < 1e-9


def test_registration_recovers_a_known_shift():
    """Sanity of the algorithm itself: the estimated field points the right way."""
    g = golden()
    p1, p2 = pyramids(g)
    av = ro.estimatereg(p1, p2)
    vx, vy = ro.velocityfield(av, (16, 16), 'bilinear')
    # im2(x) = im1(1.01 x + 0.012, 0.994 y - 0.008): source -> reference needs a positive x / negative y drift
    assert np.median(vx) > 0.005 and np.median(vy) < -0.003
