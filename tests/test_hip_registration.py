"""dtcwt_amd.registration (hip kernels) against the reference's golden vectors and the oracle."""
import numpy as np
import pytest

import dtcwt_amd
from dtcwt_amd import registration as reg
from dtcwt_amd.hip import Transform2d
from dtcwt_amd.hip._lib import DeviceArray
from oracle import registration_oracle as ro
from tests.test_registration_oracle_golden import golden, pyramids, rel

pytestmark = pytest.mark.gpu


def hip_pyramids(g, dtype=np.float64, nlevels=5):
    t = Transform2d()
    return t.forward(g['im1'].astype(dtype), nlevels=nlevels), t.forward(g['im2'].astype(dtype), nlevels=nlevels)


def test_building_blocks_vs_reference_vectors():
    g = golden()
    p1, p2 = hip_pyramids(g)
    for l, q in zip((2, 4), reg.qtildematrices(p1, p2, [2, 4])):
        assert q.shape == g['qtilde/%d' % l].shape and q.dtype == np.float64
        assert rel(q, g['qtilde/%d' % l]) < 1e-9
    assert rel(reg.solvetransform(g['solve_in']), g['solve_out']) < 1e-12
    vx, vy = reg.velocityfield(g['estimatereg'], (32, 32), 'bilinear')
    assert rel(vx, g['velocity_x']) < 1e-12 and rel(vy, g['velocity_y']) < 1e-12
    assert rel(reg.warp(g['im1'], g['estimatereg'], 'bilinear'), g['warp']) < 1e-11
    w = reg.warphighpass(p1.highpasses[2], g['estimatereg'], 'bilinear')
    assert w.dtype == np.complex128 and rel(w, g['warphighpass']) < 1e-9


def test_estimatereg_vs_reference_vectors():
    g = golden()
    p1, p2 = hip_pyramids(g)
    av = reg.estimatereg(p1, p2)
    assert av.shape == g['estimatereg'].shape and av.dtype == np.float64
    assert rel(av, g['estimatereg']) < 1e-6
    assert rel(reg.estimatereg(p1, p2, regshape=(5, 7), levels=[[4, 3], [3, 2]]), g['estimatereg_custom']) < 1e-6
    # one native call and the launch-by-launch sequence are the same kernels
    assert rel(reg.estimatereg(p1, p2, native=False), av) < 1e-12
    assert rel(reg.estimatereg(p1, p2, regshape=(5, 7), levels=[[4, 3], [3, 2]], native=False), g['estimatereg_custom']) < 1e-6
    # host pyramids (uploaded) give the same answer as device-resident ones
    o1, o2 = pyramids(g)
    assert rel(reg.estimatereg(o1, o2), av) < 1e-9


def test_float32_pyramids_and_device_outputs():
    """The bench configuration of the reference's scripts/benchmark_registration.py: float32
    pyramids with six levels; everything stays on the device."""
    g = golden()
    p1, p2 = hip_pyramids(g, np.float32, 6)
    assert isinstance(p1.hip_highpasses[3], DeviceArray) and p1.hip_highpasses[3].dtype == np.complex64
    av = reg.estimatereg(p1, p2, device_output=True)
    assert isinstance(av, DeviceArray) and av.shape == p1.hip_highpasses[3].shape[:2] + (6,)
    o1, o2 = pyramids(g, np.float32, 6)
    want = ro.estimatereg(o1, o2)
    assert rel(av.get(), want) < 5e-3          # single-precision pyramids; the solves amplify
    wt = reg.warptransform(p1, av, [2, 3], 'bilinear')
    assert isinstance(wt.hip_highpasses[2], DeviceArray) and wt.hip_highpasses[0] is p1.hip_highpasses[0]
    want_w = ro.warphighpass(o1.highpasses[2], want, 'bilinear')
    assert rel(wt.highpasses[2], want_w) < 1e-3
    assert dtcwt_amd.registration.estimatereg is reg.estimatereg


def test_estimatereg_ragged_levels_vs_oracle():
    """Levels whose sides are not multiples of the 8 x 4 tile of the batched kernels (41 x 53 ... 6 x 7; level indices are 0-based), a
    group of three levels, and a grid larger than one workgroup's fold (regshape 60 x 60)."""
    from oracle import dtcwt_oracle as o
    from dtcwt_amd.coeffs import biort, qshift
    yy, xx = np.mgrid[0:164, 0:212]
    rs = np.random.RandomState(5)
    ims = []
    for sx, sy in ((0.0, 0.0), (0.9, -0.6)):
        im = np.zeros(yy.shape)
        for _ in range(16):
            fx, fy, ph = rs.uniform(0.01, 0.12, 2).tolist() + [rs.uniform(0, 6.28)]
            im += np.cos(6.283 * (fx * (xx + sx) + fy * (yy + sy)) + ph)
        ims.append(im)
        rs = np.random.RandomState(5)
    t, to = Transform2d(), o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    p1, p2 = t.forward(ims[0], nlevels=5), t.forward(ims[1], nlevels=5)
    o1, o2 = to.forward(ims[0], nlevels=5), to.forward(ims[1], nlevels=5)
    for kw in (dict(), dict(levels=[[4, 3], [4, 3, 2], [3, 2, 1]]), dict(regshape=(60, 60), levels=[[4, 3], [3, 2]])):
        assert rel(reg.estimatereg(p1, p2, **kw), ro.estimatereg(o1, o2, **kw)) < 1e-6, kw


def test_argument_errors():
    g = golden()
    p1, p2 = hip_pyramids(g)
    with pytest.raises(ValueError):
        reg.solvetransform(np.zeros((3, 26)))
    with pytest.raises(ValueError):
        reg.qtildematrices(p1, Transform2d().forward(g['im1'][:64, :64], nlevels=5), [1])
