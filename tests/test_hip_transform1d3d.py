"""Transform1d / Transform3d of the hip backend against the oracle and the golden vectors
(patterns: the reference's tests/test_xfm1.py, test_ifm1.py, test_xfm3.py)."""
import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Transform1d, Transform3d, Pyramid
from tests import _golden as G
from tests._hip import assert_close, assert_pyramids_close, as_f64, XFM_TOL, INV_TOL, F64_TOL

pytestmark = pytest.mark.gpu


def _haar():
    h0 = np.array((1.0, 1.0)) / 2
    g0 = h0.copy()
    h1 = g0 * np.cumprod(-np.ones_like(g0))
    g1 = -h0 * np.cumprod(-np.ones_like(h0))
    return (h0, g0, h1, g1)


# ------------------------------------------------------------------------------ 1-D
def test_1d_golden():
    s = G.load('transform1d')
    for case in s['cases']:
        xn, bn, qn, nl, dt = G.parse_case(str(case))
        nl = int(nl[2:])
        tol = 1e-12 if dt == 'float64' else 2e-6
        t = Transform1d(bn, qn)
        p = t.forward(s[xn].astype(dt), nlevels=nl, include_scale=True)
        G.check_pyramid(s, case + '/fwd', p, tol)
        G.check_stored(s, case + '/inv', t.inverse(p), tol * 10)
        if nl:
            G.check_stored(s, case + '/inv_gain', t.inverse(p, s[case + '/gain_mask']), tol * 10)


@pytest.mark.parametrize('shape', [(630,), (630, 20), (64, 3), (16,), (100, 1)])
def test_1d_vs_oracle(shape):
    rs = np.random.RandomState(2)
    for dt, tol, itol in ((np.float32, XFM_TOL, INV_TOL), (np.float64, F64_TOL, 1e-11)):
        X = rs.standard_normal(shape).astype(dt)
        t, to = Transform1d(), o.Transform1d(biort('near_sym_a'), qshift('qshift_a'))
        for nl in (1, 2, 4):
            p = t.forward(X, nlevels=nl, include_scale=True)
            want = to.forward(X, nlevels=nl, include_scale=True)
            assert_pyramids_close(p, want, tol)
            z = t.inverse(p)
            zo = to.inverse(want)
            assert z.shape == zo.shape and z.dtype == X.dtype      # (n, 1) flattens like the reference
            assert_close(z, zo, itol * 3, 'inverse')
            assert_close(z.reshape(X.shape), X, itol * 3, 'PR')


@pytest.mark.parametrize('bn,qn', [('near_sym_a', 'qshift_a'), ('near_sym_b', 'qshift_d'), ('antonini', 'qshift_06'),
                                   ('legall', 'qshift_c'), ('near_sym_b', 'qshift_b')])
@pytest.mark.parametrize('shape', [(4096,), (1500, 1), (630, 40), (322, 64), (200, 130)])
def test_1d_one_launch_levels_vs_oracle(bn, qn, shape):
    """The one-launch levels (dtcwt_hip_level1d_*: k = 1 through LDS rows, k >= 32 marching down
    the rows, highpass packing fused) against the oracle, lengths that need the (1, 1) padding
    and the matching crop, gains on the way back."""
    from dtcwt_amd.hip import lowlevel as ll, default_context
    rs = np.random.RandomState(31)
    to = o.Transform1d(biort(bn), qshift(qn))
    for dt, tol, itol in ((np.float32, XFM_TOL, INV_TOL), (np.float64, F64_TOL, 1e-11)):
        X = rs.standard_normal(shape).astype(dt)
        t = Transform1d(bn, qn)
        Xd = default_context().to_device(X.reshape(shape[0], -1))
        assert ll.level1d_forward(Xd, 0, (0, 0), *biort(bn)[0:3:2]) is not None      # this path is what runs
        for nl in (1, 3, 5):
            want = to.forward(X, nlevels=nl, include_scale=True)
            p = t.forward(X, nlevels=nl, include_scale=True)
            assert_pyramids_close(p, want, tol)
            gm = rs.uniform(0.3, 1.4, size=nl)
            for g in (None, gm):
                assert_close(t.inverse(p, g), to.inverse(want, g), itol, 'inverse nl=%d' % nl)


def test_1d_errors_and_zero_levels():
    t = Transform1d()
    with pytest.raises(ValueError):
        t.forward(np.zeros(15))                         # transform1d.py:70-71
    X = np.arange(16.0)
    p = t.forward(X, nlevels=0)
    assert np.array_equal(np.ravel(p.lowpass), X)
    p = t.forward(np.random.RandomState(0).standard_normal(64), nlevels=3)
    bad = Pyramid(p.lowpass, (p.highpasses[0], p.highpasses[0], p.highpasses[2]))
    with pytest.raises(ValueError):
        t.inverse(bad)


def test_1d_inverse_mixed_dtypes_and_long_gain_mask():
    """A device pyramid whose lowpass and highpasses disagree in precision must not reach the native plan as if they
    matched (it takes the level-by-level path, like a host pyramid with mixed dtypes does), and a gain mask longer
    than nlevels is indexed per level as the reference does (dtcwt/numpy/transform1d.py:150-176), not reshaped."""
    from dtcwt_amd.hip import default_context
    rs = np.random.RandomState(12)
    X = rs.standard_normal(256).astype(np.float32)
    t = Transform1d()
    to = o.Transform1d(biort('near_sym_a'), qshift('qshift_a'))
    p = t.forward(X, nlevels=3)
    want = to.forward(X.astype(np.float64), nlevels=3)
    gm = np.array([1.0, 0.5, 2.0, 7.0, 9.0])             # two entries more than levels
    z = t.inverse(p, gm)
    assert_close(z, to.inverse(want, gm), INV_TOL)
    # float32 lowpass with complex128 highpasses, both resident on the device
    ctx = default_context()
    mixed = Pyramid(ctx.to_device(np.asarray(p.lowpass, np.float32).reshape(-1, 1)),
                    tuple(ctx.to_device(np.asarray(y, np.complex128).reshape(-1, 1)) for y in p.highpasses))
    zm = np.ravel(t.inverse(mixed))
    assert_close(zm, to.inverse(want), INV_TOL * 4)


# ------------------------------------------------------------------------------ 3-D
def _volume(s, xn):
    if xn == 'e32':
        g = slice(-16, 16)
        X, Y, Z = np.mgrid[g, g, g]
        r = np.sqrt(X * X + (Y * 1.2) ** 2 + (Z * 1.4) ** 2)
        return np.where(r <= 0.4 * 32, 1.0, 0.0)
    return s[xn]


def test_3d_golden():
    s = G.load('transform3d')
    for case in s['cases']:
        xn, bn, qn, nl, ext, d, dt = G.parse_case(str(case))
        nl, ext, d = int(nl[2:]), int(ext[3:]), bool(int(d[1:]))
        tol = 1e-12 if dt == 'float64' else 2e-6
        b = _haar() if bn == 'haar' else bn
        t = Transform3d(b, qn, ext_mode=ext)
        p = t.forward(_volume(s, xn).astype(dt), nlevels=nl, include_scale=True, discard_level_1=d)
        G.check_pyramid(s, case + '/fwd', p, tol)
        z = t.inverse(p)
        if d:
            # the fixture is the reference's literal output (axes 0 / 2 exchanged, transform3d.py:454-456):
            # reference_quirks=True reproduces it, the default returns the intended volume = its transpose
            tq = Transform3d(b, qn, ext_mode=ext, reference_quirks=True)
            G.check_stored(s, case + '/inv', tq.inverse(p), tol * 10)
            z = z.transpose(2, 1, 0)
        G.check_stored(s, case + '/inv', z, tol * 10)


@pytest.mark.parametrize('shape,ext', [((16, 24, 32), 4), ((30, 26, 22), 4), ((36, 28, 20), 8), ((8, 8, 8), 4)])
def test_3d_vs_oracle_and_pr(shape, ext):
    rs = np.random.RandomState(6)
    for dt, tol, ptol in ((np.float32, XFM_TOL, INV_TOL), (np.float64, F64_TOL, 1e-11)):
        X = rs.standard_normal(shape).astype(dt)
        t = Transform3d(ext_mode=ext)
        to = o.Transform3d(biort('near_sym_a'), qshift('qshift_a'), ext_mode=ext)
        for nl in (1, 2, 3):
            p = t.forward(X, nlevels=nl, include_scale=True)
            want = to.forward(X, nlevels=nl, include_scale=True)
            assert p.highpasses[0].dtype == (np.complex64 if dt == np.float32 else np.complex128)
            assert p.highpasses[0].shape[-1] == 28
            assert_pyramids_close(p, want, tol)
            z = t.inverse(p)
            assert z.shape == X.shape
            assert_close(z, X, ptol, 'PR nl=%d' % nl)


def test_3d_discard_level_1_and_errors():
    rs = np.random.RandomState(8)
    X = rs.standard_normal((16, 20, 12))
    t = Transform3d()
    p = t.forward(X, nlevels=3, discard_level_1=True)
    q = t.forward(X, nlevels=3)
    assert p.highpasses[0] is None
    assert_close(p.lowpass, q.lowpass, F64_TOL)
    for a, b in zip(p.highpasses[1:], q.highpasses[1:]):
        assert_close(a, b, F64_TOL)
    to = o.Transform3d(biort('near_sym_a'), qshift('qshift_a'))
    assert_close(t.inverse(p), to.inverse(to.forward(X, nlevels=3, discard_level_1=True)), 1e-11)
    with pytest.raises(ValueError):
        Transform3d(ext_mode=5).forward(X)
    with pytest.raises(ValueError):
        Transform3d().forward(np.zeros((5, 4, 4)))
    with pytest.raises(ValueError):
        Transform3d(ext_mode=8).forward(np.zeros((6, 4, 4)))


def test_3d_medium_volume_config_c4_shape():
    """BASELINE config[3] (256^3, nlevels=3) at a reduced 64^3 the oracle finishes quickly,
    plus shape/PR checks; the full 256^3 volume is covered by linearity + PR."""
    rs = np.random.RandomState(2)
    X = rs.standard_normal((64, 64, 64)).astype(np.float32)
    t = Transform3d()
    p = t.forward(X, nlevels=3)
    want = o.Transform3d(biort('near_sym_a'), qshift('qshift_a')).forward(X, nlevels=3)
    assert_pyramids_close(p, want, XFM_TOL)
    V = rs.standard_normal((256, 256, 256)).astype(np.float32)
    p = t.forward(V, nlevels=3)
    assert p.lowpass.shape == (64, 64, 64)
    assert [y.shape for y in p.highpasses] == [(128, 128, 128, 28), (64, 64, 64, 28), (32, 32, 32, 28)]
    z = t.inverse(p)
    assert np.abs(z - V).max() < INV_TOL * np.abs(V).max()


@pytest.mark.parametrize('shape', [(8, 8, 8), (12, 20, 70), (34, 18, 130), (64, 48, 80), (96, 160, 72), (14, 18, 262),
                                   (12, 16, 520)])
@pytest.mark.parametrize('bname', ['near_sym_a', 'antonini', 'legall'])
def test_3d_fused_level1_matches_generic_and_oracle(shape, bname):
    """The single-launch level 1 (dtcwt_hip_fwd3_level1) against the generic axis passes and,
    on the small shapes, the oracle."""
    X = np.random.RandomState(12).standard_normal(shape).astype(np.float32)
    t = Transform3d(biort=bname)
    assert t.fused
    g = Transform3d(biort=bname)
    g.fused = False
    for nl in (1, 2):
        p, q = t.forward(X, nlevels=nl), g.forward(X, nlevels=nl)
        assert_pyramids_close(p, q, XFM_TOL)
        if X.size <= 70000:
            want = o.Transform3d(biort(bname), qshift('qshift_a')).forward(X, nlevels=nl)
            assert_pyramids_close(p, want, XFM_TOL)
        z = t.inverse(p)
        assert_close(z, X, INV_TOL, 'PR')
        assert_close(z, g.inverse(p), INV_TOL, 'fused vs generic inverse')


@pytest.mark.parametrize('shape', [(8, 8, 16), (14, 10, 72), (40, 12, 252), (26, 34, 256), (10, 8, 260), (18, 6, 520), (72, 20, 128),
                                   (16, 130, 64)])
@pytest.mark.parametrize('bname', ['near_sym_a', 'legall'])
def test_3d_level1_march_matches_tile_program_and_oracle(shape, bname, monkeypatch):
    """k_fwd3m_l1 (fused3d_march.hpp: level 1 as a marching pair of wavefronts) against the tile program k_fwd3_l1 and the
    oracle: one strip with and without idle lanes (16 .. 256 columns), two and three strips (260, 520: halo lanes at the
    interior boundaries, the mirror columns taken in-lane at the faces), slice counts that are not whole chunks or ring
    periods, whole-volume and 8-slice chunks."""
    X = np.random.RandomState(31).standard_normal(shape).astype(np.float32)
    t = Transform3d(biort=bname)
    monkeypatch.setenv('DTCWT_HIP_FWD3_MARCH', '0')
    p0 = t.forward(X, nlevels=1)
    monkeypatch.setenv('DTCWT_HIP_FWD3_MARCH', '1')
    for chunk in (None, '8'):
        if chunk:
            monkeypatch.setenv('DTCWT_HIP_FWD3_CHUNK', chunk)
        p1 = t.forward(X, nlevels=1)
        assert_pyramids_close(p1, p0, XFM_TOL)
        want = o.Transform3d(biort(bname), qshift('qshift_a')).forward(as_f64(X), nlevels=1)
        assert_pyramids_close(p1, want, XFM_TOL, same_dtype=False)
        assert_close(t.inverse(p1), X, INV_TOL, 'PR')


@pytest.mark.parametrize('shape', [(20, 40, 40), (24, 44, 132), (64, 48, 256), (42, 40, 260), (100, 64, 64), (36, 130, 72)])
@pytest.mark.parametrize('chunks', [(None, None, None), ('20', '4', '1'), ('14', '7', '0')])
def test_3d_level1_long_filters_match_generic_and_oracle(shape, chunks, monkeypatch):
    """near_sym_b (13 / 19 taps: "qbgn-style", the set the reference's own 3-D MATLAB vectors use) through the two-launch level 1
    of fused3d_long.hpp -- k_fwd1m / k_inv1m with plane volumes, k_fwd3l_axis0, k_inv3l_axis0 -- against the axis-by-axis generic
    kernels (DTCWT_HIP_LONG3D=0: the path before round 5) and the oracle: one and two strips of cells (a second strip of 2 and of
    66 cells), two strips of the in-slice march, slice counts that are not whole ring periods or chunks, one chunk and several;
    the in-slice marches as one strip without halo lanes (the EDGE builds: mirror columns selected in-lane at both faces, with
    and without idle lanes) wherever the rows have at most 256 columns, and with halo lanes everywhere."""
    X = np.random.RandomState(37).standard_normal(shape).astype(np.float32)
    t = Transform3d(biort='near_sym_b', qshift='qshift_b')
    monkeypatch.setenv('DTCWT_HIP_LONG3D', '0')
    p0 = t.forward(X, nlevels=1)
    z0 = t.inverse(p0)
    monkeypatch.setenv('DTCWT_HIP_LONG3D', '1')
    if chunks[0]:
        monkeypatch.setenv('DTCWT_HIP_LONG3D_CHUNK', chunks[0])
        monkeypatch.setenv('DTCWT_HIP_LONG3D_ICHUNK', chunks[1])
        monkeypatch.setenv('DTCWT_HIP_LONG3D_EDGE', chunks[2])
    p1 = t.forward(X, nlevels=1)
    assert_pyramids_close(p1, p0, XFM_TOL)
    if X.size <= 900000:
        want = o.Transform3d(biort('near_sym_b'), qshift('qshift_b')).forward(as_f64(X), nlevels=1)
        assert_pyramids_close(p1, want, XFM_TOL, same_dtype=False)
    # round 6: the default forward is axis 0 first, then k_fwd3l_slices (two intermediate volumes); '2' keeps the round-5 cut
    # (four plane volumes, k_fwd1m<PLANES> + k_fwd3l_axis0) -- both against the generic path, and the new one with short bands
    monkeypatch.setenv('DTCWT_HIP_LONG3D', '2')
    assert_pyramids_close(t.forward(X, nlevels=1), p0, XFM_TOL)
    monkeypatch.setenv('DTCWT_HIP_LONG3D', '1')
    monkeypatch.setenv('DTCWT_HIP_LONG3D_BAND', '20')
    assert_pyramids_close(t.forward(X, nlevels=1), p0, XFM_TOL)
    monkeypatch.delenv('DTCWT_HIP_LONG3D_BAND')
    z1 = t.inverse(p1)
    assert_close(z1, X, INV_TOL, 'PR')
    assert_close(t.inverse(p0), z0, INV_TOL, 'long vs generic inverse')
    # the inverse likewise: k_inv3l_slices + the axis-0 sum filter (default), the round-5 cut ('2'), short bands of the new launch
    monkeypatch.setenv('DTCWT_HIP_LONG3D', '2')
    assert_close(t.inverse(p0), z0, INV_TOL, 'round-5 cut vs generic inverse')
    monkeypatch.setenv('DTCWT_HIP_LONG3D', '1')
    monkeypatch.setenv('DTCWT_HIP_LONG3D_BAND', '8')
    assert_close(t.inverse(p0), z0, INV_TOL, 'long (8-row bands) vs generic inverse')
    monkeypatch.delenv('DTCWT_HIP_LONG3D_BAND')
    # the whole transform as one native plan: levels >= 2 on the q-shift tile programs
    p2 = t.forward(X, nlevels=2)
    assert_close(t.inverse(p2), X, INV_TOL, 'PR, two levels')


@pytest.mark.parametrize('shape,ext', [((42, 46, 90), 4), ((44, 52, 84), 8), ((80, 80, 80), 4), ((48, 40, 200), 8)])
@pytest.mark.parametrize('qname', ['qshift_a', 'qshift_b', 'qshift_d'])
def test_3d_fused_level2_matches_generic_and_oracle(shape, ext, qname):
    """Levels >= 2 through dtcwt_hip_fwd3_level2 (edge padding of both ext_modes as index math)
    against the generic axis passes and the oracle."""
    X = np.random.RandomState(14).standard_normal(shape).astype(np.float32)
    t = Transform3d(qshift=qname, ext_mode=ext)
    g = Transform3d(qshift=qname, ext_mode=ext)
    g.fused = False
    p, q = t.forward(X, nlevels=3, include_scale=True), g.forward(X, nlevels=3, include_scale=True)
    assert_pyramids_close(p, q, XFM_TOL)
    want = o.Transform3d(biort('near_sym_a'), qshift(qname), ext_mode=ext).forward(X, nlevels=3, include_scale=True)
    assert_pyramids_close(p, want, XFM_TOL)
    assert_close(t.inverse(p), want_inverse(want, qname, ext), INV_TOL, 'inverse')
    assert_close(t.inverse(p), g.inverse(p), INV_TOL, 'fused vs generic inverse')


def want_inverse(pyr, qname, ext):
    return o.Transform3d(biort('near_sym_a'), qshift(qname), ext_mode=ext).inverse(pyr)


def test_3d_fused_vs_generic_random_sweep():
    """Seeded random shapes / wavelets / ext_modes: every fused level kernel (or its fallback
    decision) against the generic axis passes, forward and inverse."""
    rs = np.random.RandomState(2024)
    biorts = ['near_sym_a', 'antonini', 'legall', 'near_sym_b']
    qshifts = ['qshift_a', 'qshift_b', 'qshift_06', 'qshift_c']
    for trial in range(24):
        ext = int(rs.choice([4, 8]))
        mult = 2 if ext == 4 else 4
        shape = tuple(int(mult * rs.randint(12 // mult, 72 // mult + 1)) for _ in range(3))
        bn, qn = biorts[rs.randint(len(biorts))], qshifts[rs.randint(len(qshifts))]
        nl = int(rs.randint(1, 4))
        X = rs.standard_normal(shape).astype(np.float32)
        t, g = Transform3d(bn, qn, ext_mode=ext), Transform3d(bn, qn, ext_mode=ext)
        g.fused = False
        p, q = t.forward(X, nlevels=nl, include_scale=True), g.forward(X, nlevels=nl, include_scale=True)
        assert_pyramids_close(p, q, XFM_TOL)
        zt, zg = t.inverse(p), g.inverse(p)
        assert zt.shape == zg.shape == X.shape
        assert_close(zt, zg, INV_TOL, 'inverse %s %s %s ext%d nl%d' % (shape, bn, qn, ext, nl))
        assert_close(zt, X, INV_TOL, 'PR')


def test_3d_long_filters_random_sweep():
    """Seeded random volumes through the two-launch level 1 of the 13 / 19-tap filters (fused3d_long.hpp) against the generic
    axis passes: rows of 40 .. 300 columns (one strip with halo lanes, one strip without -- 236 .. 256 --, two strips), 20 .. 90
    slices, one to three levels, both ext_modes."""
    rs = np.random.RandomState(77)
    for trial in range(14):
        ext = int(rs.choice([4, 8]))
        mult = 2 if ext == 4 else 4
        n0 = int(mult * rs.randint(20 // mult, 92 // mult))
        n1 = int(mult * rs.randint(40 // mult, 80 // mult))
        n2 = int(4 * rs.choice([rs.randint(10, 59), rs.randint(59, 65), rs.randint(65, 76)]))
        qn = ['qshift_b', 'qshift_a', 'qshift_d'][rs.randint(3)]
        nl = int(rs.randint(1, 4))
        X = rs.standard_normal((n0, n1, n2)).astype(np.float32)
        t, g = Transform3d('near_sym_b', qn, ext_mode=ext), Transform3d('near_sym_b', qn, ext_mode=ext)
        g.fused = False
        p, q = t.forward(X, nlevels=nl), g.forward(X, nlevels=nl)
        assert_pyramids_close(p, q, XFM_TOL)
        zt = t.inverse(p)
        assert_close(zt, g.inverse(p), INV_TOL, 'inverse %s %s ext%d nl%d' % (X.shape, qn, ext, nl))
        assert_close(zt, X, INV_TOL, 'PR')


def test_native_plans_one_call_per_transform(monkeypatch):
    """Transform3d / Transform1d run as ONE native call (dtcwt_hip_plan3d_* / plan1d_*) and give bit-identical
    results to the level-by-level sequencing from Python (plans switched off)."""
    from dtcwt_amd.hip import _lib
    rs = np.random.RandomState(31)
    V = rs.standard_normal((40, 48, 56)).astype(np.float32)
    t = Transform3d(ext_mode=4)
    assert t._plan(V.shape, 3) is not None
    calls = []
    real = _lib.lib().dtcwt_hip_fwd3_level2
    p = t.forward(V, nlevels=3, include_scale=True)
    z = t.inverse(p)
    monkeypatch.setattr(Transform3d, '_plan', lambda self, shape, nlevels: None)
    t2 = Transform3d(ext_mode=4)
    q = t2.forward(V, nlevels=3, include_scale=True)
    assert np.array_equal(p.lowpass, q.lowpass)
    for l in range(3):
        assert np.array_equal(p.highpasses[l], q.highpasses[l])
        assert np.array_equal(p.scales[l], q.scales[l])
    assert np.array_equal(z, t2.inverse(q))
    monkeypatch.undo()
    # discard_level_1 through the plan: level 1 lowpass only, Yh[0] None, inverse with the lowpass-only merge
    t3 = Transform3d(ext_mode=4)
    pd_ = t3.forward(V, nlevels=2, discard_level_1=True)
    assert pd_.highpasses[0] is None
    to = o.Transform3d(biort('near_sym_a'), qshift('qshift_a'), ext_mode=4)
    want = to.forward(V.astype(np.float64), nlevels=2, discard_level_1=True)
    assert_close(pd_.highpasses[1], want.highpasses[1], XFM_TOL)
    assert_close(t3.inverse(pd_), to.inverse(want), INV_TOL)
    # 1-D: one vector and 64 signals side by side, float32 and float64
    for shape, dt, tol in (((4096,), np.float64, F64_TOL), ((630 * 2, 64), np.float32, XFM_TOL)):
        x = rs.standard_normal(shape).astype(dt)
        t1 = Transform1d()
        p1 = t1.forward(x, nlevels=4, include_scale=True)
        assert len(t1._plans) == 1 and list(t1._plans.values())[0] is not None
        gm = rs.uniform(0.5, 1.5, size=4)
        z1 = t1.inverse(p1, gm)
        monkeypatch.setenv('DTCWT_HIP_PLAN1D', '0')
        t1b = Transform1d()
        q1 = t1b.forward(x, nlevels=4, include_scale=True)
        assert list(t1b._plans.values()) == []
        assert np.array_equal(p1.lowpass, q1.lowpass)
        for l in range(4):
            assert np.array_equal(p1.highpasses[l], q1.highpasses[l])
            assert np.array_equal(p1.scales[l], q1.scales[l])
        assert np.array_equal(z1, t1b.inverse(q1, gm))
        monkeypatch.delenv('DTCWT_HIP_PLAN1D')
        w1 = o.Transform1d(biort('near_sym_a'), qshift('qshift_a')).forward(x.astype(np.float64), nlevels=4)
        assert_close(p1.highpasses[3], w1.highpasses[3], tol)


def test_c4_whole_pyramid_vs_oracle_256cubed():
    """BASELINE config[3]: 256^3 float32, nlevels=3 -- every level and octant against the oracle evaluated in
    float64 on the same samples (not just reconstruction, which any self-consistent pair passes), then
    linearity and the inverse."""
    rs = np.random.RandomState(256)
    V = rs.standard_normal((256, 256, 256)).astype(np.float32)
    t = Transform3d()
    p = t.forward(V, nlevels=3)
    want = o.Transform3d(biort('near_sym_a'), qshift('qshift_a')).forward(V.astype(np.float64), nlevels=3)
    assert_close(p.lowpass, want.lowpass, XFM_TOL, 'Yl')
    for l in range(3):
        assert p.highpasses[l].shape == want.highpasses[l].shape
        assert_close(p.highpasses[l], want.highpasses[l], XFM_TOL, 'Yh[%d]' % l)
        for oct_ in range(7):
            a, b = p.highpasses[l][..., 4 * oct_:4 * oct_ + 4], want.highpasses[l][..., 4 * oct_:4 * oct_ + 4]
            assert_close(a, b, 2 * XFM_TOL, 'Yh[%d] octant %d' % (l, oct_))
    z = t.inverse(p)
    assert_close(z, V.astype(np.float64), INV_TOL, 'reconstruction')
    W = rs.standard_normal((256, 256, 256)).astype(np.float32)
    q = t.forward(W, nlevels=3)
    r = t.forward(1.5 * V - 0.25 * W, nlevels=3)
    for l in range(3):
        assert_close(r.highpasses[l], 1.5 * p.highpasses[l] - 0.25 * q.highpasses[l], 1e-6, 'linearity Yh[%d]' % l)


def test_c4_qbgn_whole_pyramid_vs_oracle_256cubed():
    """BASELINE config[3] as it is worded ("qbgn-style"): the wavelet set and flags of the reference's own 3-D vectors
    (tests/test_againstmatlab.py:114-124: near_sym_b / qshift_b, nlevels=3, include_scale=True) at 256^3 float32 -- the
    lowpass, every scale, every level and every octant against the oracle evaluated in float64 on the same samples, then the
    inverse."""
    rs = np.random.RandomState(257)
    V = rs.standard_normal((256, 256, 256)).astype(np.float32)
    t = Transform3d(biort='near_sym_b', qshift='qshift_b')
    p = t.forward(V, nlevels=3, include_scale=True)
    want = o.Transform3d(biort('near_sym_b'), qshift('qshift_b')).forward(V.astype(np.float64), nlevels=3, include_scale=True)
    assert_close(p.lowpass, want.lowpass, XFM_TOL, 'Yl')
    assert len(p.scales) == 3
    for l in range(3):
        assert p.scales[l].shape == want.scales[l].shape
        assert_close(p.scales[l], want.scales[l], XFM_TOL, 'Yscale[%d]' % l)
        assert p.highpasses[l].shape == want.highpasses[l].shape
        assert_close(p.highpasses[l], want.highpasses[l], XFM_TOL, 'Yh[%d]' % l)
        for oct_ in range(7):
            a, b = p.highpasses[l][..., 4 * oct_:4 * oct_ + 4], want.highpasses[l][..., 4 * oct_:4 * oct_ + 4]
            assert_close(a, b, 2 * XFM_TOL, 'Yh[%d] octant %d' % (l, oct_))
    assert_close(t.inverse(p), V.astype(np.float64), INV_TOL, 'reconstruction')


@pytest.mark.parametrize('lens', [(5, 7), (7, 5), (5, 3), (7, 7), (3, 3)])
def test_3d_level1_asymmetric_user_taps_keep_the_full_tap_vector(lens, monkeypatch):
    """User-supplied level-1 filters that are NOT symmetric (and length pairs outside the shipped sets): the marching pair
    k_fwd3m_l1 folds mirror pairs and must not take them (fused3d.hip: fwd3m_ok) -- whatever kernel runs, the forward
    level 1 equals the oracle's with the same taps (the reference filters with the vector as given, lowlevel.py:47-80)."""
    rs = np.random.RandomState(41)
    m0, m1 = lens
    h0o, h1o = rs.standard_normal(m0), rs.standard_normal(m1)
    h0o /= np.abs(h0o).sum(); h1o /= np.abs(h1o).sum()
    g0o, g1o = rs.standard_normal(m1), rs.standard_normal(m0)          # never used by the forward
    taps = (h0o, g0o, h1o, g1o)
    X = rs.standard_normal((16, 24, 64)).astype(np.float32)
    monkeypatch.setenv('DTCWT_HIP_FWD3_MARCH', '1')                     # "wherever it applies": it does not apply here
    p = Transform3d(biort=taps).forward(X, nlevels=1)
    want = o.Transform3d(taps, qshift('qshift_a')).forward(as_f64(X), nlevels=1)
    assert_pyramids_close(p, want, XFM_TOL, same_dtype=False)
    # ... and the symmetric part of the same taps does take the march where the lengths are the shipped ones, with the same answer
    hs0, hs1 = (h0o + h0o[::-1]) / 2, (h1o + h1o[::-1]) / 2
    ps = Transform3d(biort=(hs0, g0o, hs1, g1o)).forward(X, nlevels=1)
    ws = o.Transform3d((hs0, g0o, hs1, g1o), qshift('qshift_a')).forward(as_f64(X), nlevels=1)
    assert_pyramids_close(ps, ws, XFM_TOL, same_dtype=False)
