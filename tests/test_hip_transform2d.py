"""Transform2d of the hip backend against the oracle and the reference's golden vectors.

Pattern: the reference's tests/test_openclxfm2.py:25-90 (accelerated vs gold, odd sizes,
_bp wavelets, nlevels=0, 1-D row input), tests/test_xfm2.py / test_ifm2.py (shapes, dtypes,
perfect reconstruction) and tests/test_tfTransform2d.py:193-263,434-453 (batched layouts,
random gain masks).  Everything goes through the C ABI."""
import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Transform2d, Pyramid, DeviceArray, default_context
from tests import _golden as G
from tests._hip import (assert_close, assert_pyramids_close, as_f64, cast_pyramid, XFM_TOL, INV_TOL,
                        F64_TOL)

pytestmark = pytest.mark.gpu

WAVES = [('near_sym_a', 'qshift_a'), ('antonini', 'qshift_06'), ('legall', 'qshift_c'),
         ('near_sym_b', 'qshift_d'), ('near_sym_b', 'qshift_b'), ('near_sym_b_bp', 'qshift_b_bp'),
         ('near_sym_a', 'qshift_32')]


def _mandrill():
    return G.load('mandrill')['mandrill']


def test_plan_is_used_for_float32():
    t = Transform2d()
    assert t.plan(1, 256, 256, 3) is not None
    assert Transform2d('near_sym_b_bp', 'qshift_b_bp').plan(1, 256, 256, 3) is not None
    with pytest.raises(NotImplementedError):          # band-pass pair of another length than the q-shift set
        Transform2d('near_sym_a', (np.ones(10),) * 8 + (np.ones(12),) * 4).plan(1, 256, 256, 3)
    with pytest.raises(NotImplementedError):          # levels narrower than 40 samples: generic kernels
        t.plan(1, 64, 64, 3)


@pytest.mark.parametrize('bn,qn', WAVES)
@pytest.mark.parametrize('shape', [(64, 64), (36, 52), (33, 47), (130, 70), (8, 8), (2, 2), (100, 6)])
def test_forward_inverse_vs_oracle_f32(bn, qn, shape):
    rs = np.random.RandomState(17)
    X = rs.standard_normal(shape).astype(np.float32)
    t, to = Transform2d(bn, qn), o.Transform2d(biort(bn), qshift(qn))
    for nl in (1, 2, 3, 4):
        # gold = the oracle evaluated in float64 on the same float32 samples: float32
        # parity is max|a-b|/max|b| <= 1e-6 per subband
        try:
            want = to.forward(as_f64(X), nlevels=nl, include_scale=True)
        except Exception:
            continue                       # shapes the reference itself cannot transform
        p = t.forward(X, nlevels=nl, include_scale=True)
        assert p.lowpass.dtype == np.float32 and all(y.dtype == np.complex64 for y in p.highpasses)
        assert_pyramids_close(p, want, XFM_TOL, same_dtype=False)
        gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.25)
        for g in (None, gm):
            z = t.inverse(cast_pyramid(want, np.float32), g)
            assert z.dtype == np.float32
            assert_close(z, to.inverse(want, g), INV_TOL, 'inverse nl=%d' % nl)


@pytest.mark.parametrize('bn,qn', [('near_sym_a', 'qshift_a'), ('near_sym_b_bp', 'qshift_b_bp')])
def test_forward_inverse_vs_oracle_f64(bn, qn):
    rs = np.random.RandomState(18)
    X = rs.standard_normal((36, 52))
    t, to = Transform2d(bn, qn), o.Transform2d(biort(bn), qshift(qn))
    want = to.forward(X, nlevels=3, include_scale=True)
    p = t.forward(X, nlevels=3, include_scale=True)
    assert p.lowpass.dtype == np.float64 and p.highpasses[0].dtype == np.complex128
    assert_pyramids_close(p, want, F64_TOL)
    if 'bp' not in bn:                    # the _bp sets are not perfect-reconstruction in the reference either
        assert_close(t.inverse(p), X, 1e-11, 'PR f64')
    assert_close(t.inverse(p), to.inverse(want), 1e-11, 'inverse f64')
    gm = rs.uniform(0.3, 1.4, size=(6, 3))
    assert_close(t.inverse(want, gm), to.inverse(want, gm), F64_TOL)


def test_golden_fixtures():
    s = G.load('transform2d')
    for case in s['cases']:
        xn, bn, qn, nl, dt = G.parse_case(str(case))
        nl = int(nl[2:])
        tol = 1e-12 if dt == 'float64' else 2e-6
        t = Transform2d(bn, qn)
        p = t.forward(s[xn].astype(dt), nlevels=nl, include_scale=True)
        G.check_pyramid(s, case + '/fwd', p, tol)
        if case + '/fwd/Yl/data' in s:
            ref = G.StoredPyramid(s, case + '/fwd')
            G.check_stored(s, case + '/inv', t.inverse(ref), tol * 10)
            if nl:
                G.check_stored(s, case + '/inv_gain', t.inverse(ref, s[case + '/gain_mask']), tol * 10)


def test_mandrill_golden_summaries_config_c1():
    """BASELINE config[0]: 512x512 mandrill, near_sym_a/qshift_a, nlevels=3 (and 4), checked
    against the reference's outputs through its own test reduction (tests/util.py:46-60)."""
    s = G.load('mandrill')
    mand = s['mandrill']
    for nl in (3, 4):
        for bn, qn in (('near_sym_a', 'qshift_a'), ('near_sym_b_bp', 'qshift_b_bp')):
            key = 'nl%d-%s-%s-float32' % (nl, bn, qn)
            p = Transform2d(bn, qn).forward(mand, nlevels=nl, include_scale=True)
            assert np.abs(G.summarise_mat(p.lowpass) - s[key + '/Yl']).max() <= 1e-5
            for l in range(nl):
                assert np.abs(G.summarise_mat(p.highpasses[l]) - s[key + '/Yh%d' % l]).max() <= 1e-5
                assert np.abs(G.summarise_mat(p.scales[l]) - s[key + '/Ys%d' % l]).max() <= 1e-5
            # and against the float64 reference outputs
            key64 = 'nl%d-%s-%s-float64' % (nl, bn, qn)
            for l in range(nl):
                e = (np.abs(p.highpasses[l].astype(np.complex128)) ** 2).sum()
                assert abs(e - float(s[key64 + '/Yh%d_energy' % l])) <= 1e-5 * float(s[key64 + '/Yh%d_energy' % l])
    p = Transform2d().forward(mand, nlevels=3)
    assert p.lowpass.shape == (128, 128)
    assert abs(p.lowpass.astype(np.float64).sum() - 33183.462677941905) < 0.05
    assert abs(p.highpasses[2][5, 7, 3] - (-0.2492147741 - 0.0642030041j)) < 1e-6


def test_odd_sizes_and_crops():
    s = G.load('mandrill')
    mand = s['mandrill']
    t = Transform2d()
    for name, crop, nl in (('r509', mand[:509, :], 3), ('c509', mand[:, :509], 3),
                           ('rc509', mand[:509, :509], 3), ('crop233x301', mand[:233, :301], 4)):
        p = t.forward(crop, nlevels=nl)
        G.check_pyramid(s, name + '/fwd', p, 2e-6, check_dtype=False)
        z = t.inverse(p)
        G.check_stored(s, name + '/inv', z, 1e-5, check_dtype=False)
        assert np.abs(z[:crop.shape[0], :crop.shape[1]] - crop).max() < 5e-6


def test_shapes_like_reference():
    t = Transform2d()
    m = _mandrill()
    p = t.forward(m[:509, :], nlevels=2)
    assert p.lowpass.shape == (256, 256)
    assert [y.shape for y in p.highpasses] == [(255, 256, 6), (128, 128, 6)]
    assert t.inverse(p).shape == (510, 512)
    p = t.forward(m[:36, :52], nlevels=3)
    assert p.lowpass.shape == (10, 14)
    assert [y.shape[:2] for y in p.highpasses] == [(18, 26), (9, 13), (5, 7)]
    p = t.forward(m[:233, :301], nlevels=4)
    assert p.lowpass.shape == (30, 38)
    assert [y.shape[:2] for y in p.highpasses] == [(117, 151), (59, 76), (30, 38), (15, 19)]


def test_zero_levels_and_row_vector_and_errors():
    m = _mandrill()
    t = Transform2d()
    p = t.forward(m, nlevels=0)
    assert np.array_equal(p.lowpass, m) and p.highpasses == ()
    assert np.array_equal(t.inverse(p), m)
    p = t.forward(m[0, :], nlevels=3)                     # tests/test_openclxfm2.py:44-48
    want = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(m[0, :], nlevels=3)
    assert_pyramids_close(p, want, XFM_TOL)
    with pytest.raises(ValueError):
        t.forward(np.dstack((m, m)))
    with pytest.raises(ValueError):
        Transform2d(biort=biort('near_sym_a')[:3]).forward(m)
    # inconsistent pyramid
    p = t.forward(m, nlevels=3)
    bad = Pyramid(p.lowpass, (p.highpasses[0], p.highpasses[2], p.highpasses[2]))
    with pytest.raises(ValueError):
        t.inverse(bad)


def test_integer_input_is_float64():
    m = (np.arange(64 * 48).reshape(64, 48) % 17).astype(np.int32)
    p = Transform2d().forward(m, nlevels=2)
    assert p.lowpass.dtype == np.float64 and p.highpasses[0].dtype == np.complex128
    want = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(m, nlevels=2)
    assert_pyramids_close(p, want, F64_TOL)


def test_device_resident_pyramid():
    ctx = default_context()
    m = _mandrill()
    t = Transform2d()
    d = ctx.to_device(m)
    p = t.forward(d, nlevels=4, include_scale=True)
    assert isinstance(p.hip_lowpass, DeviceArray)
    assert all(isinstance(y, DeviceArray) for y in p.hip_highpasses)
    assert all(isinstance(y, DeviceArray) for y in p.hip_scales)
    z = t.inverse(p, device_output=True)
    assert isinstance(z, DeviceArray)
    assert np.abs(z.get() - m).max() < 5e-6
    assert p.lowpass is p.lowpass                          # memoised host copy


@pytest.mark.parametrize('fmt', ['nhw', 'chw', 'hwn', 'hwc', 'nchw', 'nhwc'])
def test_batched_channels(fmt):
    rs = np.random.RandomState(4)
    imgs = rs.standard_normal((6, 40, 72)).astype(np.float32)          # [N, h, w]
    t, to = Transform2d(), o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    if fmt in ('nhw', 'chw'):
        X = imgs
        pick = lambda A, i: A[i]
    elif fmt in ('hwn', 'hwc'):
        X = np.moveaxis(imgs, 0, 2)
        pick = lambda A, i: A[:, :, i]
    elif fmt == 'nchw':
        X = imgs.reshape(2, 3, 40, 72)
        pick = lambda A, i: A[i // 3, i % 3]
    else:
        X = np.moveaxis(imgs.reshape(2, 3, 40, 72), 1, 3)
        pick = lambda A, i: A[i // 3, :, :, i % 3]
    p = t.forward_channels(X, fmt, nlevels=3, include_scale=True)
    for i in range(6):
        want = to.forward(imgs[i], nlevels=3, include_scale=True)
        assert_close(pick(p.lowpass, i), want.lowpass, XFM_TOL)
        for l in range(3):
            assert_close(pick(p.highpasses[l], i), want.highpasses[l], XFM_TOL)
            assert_close(pick(p.scales[l], i), want.scales[l], XFM_TOL)
    gm = rs.uniform(0.5, 1.5, size=(6, 3))
    z = t.inverse_channels(p, fmt, gain_mask=gm)
    assert z.shape == X.shape
    for i in range(6):
        want = to.forward(imgs[i], nlevels=3)
        assert_close(pick(z, i), to.inverse(want, gm), INV_TOL)


def test_perfect_reconstruction_large_and_linear():
    """Size-independent properties at BASELINE's bench size (config[1]): 4096x4096 float32,
    nlevels=4 -- perfect reconstruction, linearity, energy of the tight-ish frame."""
    rs = np.random.RandomState(0)
    X = rs.standard_normal((4096, 4096)).astype(np.float32)
    Y = rs.standard_normal((4096, 4096)).astype(np.float32)
    t = Transform2d()
    p = t.forward(X, nlevels=4)
    assert p.lowpass.shape == (512, 512)
    assert [y.shape for y in p.highpasses] == [(2048, 2048, 6), (1024, 1024, 6), (512, 512, 6), (256, 256, 6)]
    z = t.inverse(p)
    assert np.abs(z - X).max() < 2e-5 * np.abs(X).max()        # reference's own f32 PR: 1.2e-6 * max on N(0,1)
    q = t.forward(Y, nlevels=4)
    r = t.forward(2.0 * X - 0.5 * Y, nlevels=4)
    assert_close(r.lowpass, 2.0 * p.lowpass - 0.5 * q.lowpass, 1e-6, 'linearity Yl')
    for l in range(4):
        assert_close(r.highpasses[l], 2.0 * p.highpasses[l] - 0.5 * q.highpasses[l], 1e-6, 'linearity Yh[%d]' % l)
    # a corner block of the big transform equals the oracle on a crop that contains its support
    crop = X[:256, :256]
    want = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(crop, nlevels=2)
    assert_close(p.highpasses[0][:96, :96], want.highpasses[0][:96, :96], XFM_TOL)
    assert_close(p.highpasses[1][:40, :40], want.highpasses[1][:40, :40], XFM_TOL)


def test_compat_wrappers_follow_backend():
    """dtcwt_amd.compat mirrors dtcwt/compat.py on the active (hip) backend."""
    from dtcwt_amd import compat
    m = _mandrill()
    Yl, Yh = compat.dtwavexfm2(m, 3)
    want = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(as_f64(m), nlevels=3)
    assert_close(Yl, want.lowpass, XFM_TOL)
    assert_close(Yh[2], want.highpasses[2], XFM_TOL)
    assert np.abs(compat.dtwaveifm2(Yl, Yh) - m).max() < 5e-6
    Yl, Yh, Ys = compat.dtwavexfm2b(m[:64, :64], 2, 'near_sym_b_bp', 'qshift_b_bp', include_scale=True)
    assert len(Ys) == 2 and Yh[0].shape == (32, 32, 6)
    v = m[:, 0].astype(np.float64)
    Yl1, Yh1 = compat.dtwavexfm(v, 4)
    assert np.abs(compat.dtwaveifm(Yl1, Yh1) - v).max() < 1e-10
    V = np.random.RandomState(1).standard_normal((16, 16, 16))
    Yl3, Yh3 = compat.dtwavexfm3(V, 2)
    assert np.abs(compat.dtwaveifm3(Yl3, Yh3) - V).max() < 1e-10


def test_batched_config_c3_shape(monkeypatch):
    """BASELINE config[2] (batched 64 x 1024 x 1024 f32, nlevels=5), reduced to 6 images for the
    oracle comparison; the full batch is checked through per-image consistency + PR."""
    rs = np.random.RandomState(1)
    X = rs.standard_normal((6, 1024, 1024)).astype(np.float32)
    t = Transform2d()
    p = t.forward_channels(X, 'nhw', nlevels=5)
    assert p.lowpass.shape == (6, 64, 64)
    assert [y.shape for y in p.highpasses] == [(6, 512, 512, 6), (6, 256, 256, 6), (6, 128, 128, 6),
                                               (6, 64, 64, 6), (6, 32, 32, 6)]
    to = o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    for i in (0, 5):
        want = to.forward(as_f64(X[i]), nlevels=5)
        assert_close(p.lowpass[i], want.lowpass, XFM_TOL)
        for l in range(5):
            assert_close(p.highpasses[l][i], want.highpasses[l], XFM_TOL)
    z = t.inverse_channels(p, 'nhw')
    assert np.abs(z - X).max() < 2e-5 * np.abs(X).max()
    # the FULL batch of 64 (the BASELINE configuration itself): every level of every image equals the same image
    # transformed alone bit for bit, three of them also agree with the oracle, and the batch reconstructs
    ctx = default_context()
    Xb = rs.standard_normal((64, 1024, 1024)).astype(np.float32)
    pb = t.forward_channels(ctx.to_device(Xb), 'nhw', nlevels=5)
    low, high = pb.lowpass, pb.highpasses
    # (left alone, the plan of ONE 1024^2 image takes the per-level tile programs and the batch the marching launches --
    # profiles/r04/ab_march_sizes.txt -- which agree to rounding, not to the bit: first as chosen, then the same program)
    single = t.forward(Xb[7], nlevels=5)
    assert_close(low[7], single.lowpass, XFM_TOL, 'batch vs one image, programs as chosen')
    assert t.plan(1, 1024, 1024, 5).launches() == (False, False) and t.plan(64, 1024, 1024, 5).launches() == (True, True)
    monkeypatch.setenv('DTCWT_HIP_MARCH', '1')
    for i in range(64):
        single = t.forward(Xb[i], nlevels=5)
        assert np.array_equal(low[i], single.lowpass), i
        for l in range(5):
            assert np.array_equal(high[l][i], single.highpasses[l]), (i, l)
    monkeypatch.delenv('DTCWT_HIP_MARCH')
    for i in (0, 31, 63):
        want = to.forward(as_f64(Xb[i]), nlevels=5)
        assert_close(low[i], want.lowpass, XFM_TOL, 'image %d Yl' % i)
        for l in range(5):
            assert_close(high[l][i], want.highpasses[l], XFM_TOL, 'image %d Yh[%d]' % (i, l))
    zb = t.inverse_channels(pb, 'nhw')
    assert np.abs(zb - Xb).max() < 2e-5 * np.abs(Xb).max()


def test_two_contexts_from_two_threads():
    """a context is single-threaded, distinct contexts may be driven
    from distinct host threads (ctypes releases the GIL during the calls)."""
    import threading
    from dtcwt_amd.hip import Context
    rs = np.random.RandomState(21)
    images = [rs.standard_normal((384, 512)).astype(np.float32) for _ in range(2)]
    want = [o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(x.astype(np.float64), nlevels=3) for x in images]
    errors = []

    def work(k):
        try:
            ctx = Context(0)
            t = Transform2d(ctx=ctx)
            for _ in range(20):
                p = t.forward(images[k], nlevels=3)
                z = t.inverse(p)
            assert_pyramids_close(p, want[k], XFM_TOL, same_dtype=False)
            assert_close(z, images[k], INV_TOL, 'PR')
        except Exception as e:           # surfaced in the main thread below
            errors.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errors, errors


@pytest.mark.parametrize('bn,qn', [('near_sym_b_bp', 'qshift_b_bp'), ('near_sym_b_bp', 'qshift_a'),
                                   ('near_sym_a', 'qshift_b_bp'), ('near_sym_b_bp', 'qshift_d')])
def test_bandpass_sets_use_the_fused_plan(bn, qn):
    """6-vector biort / 12-vector q-shift sets (third filter for the diagonal subbands,
    transform2d.py:116-129, :145-155, :250-271, :283-291): fused kernels, any mix with the
    ordinary sets, against the oracle."""
    rs = np.random.RandomState(31)
    X = rs.standard_normal((200, 264)).astype(np.float32)
    t = Transform2d(bn, qn)
    assert t.plan(1, 200, 264, 3) is not None
    to = o.Transform2d(biort(bn), qshift(qn))
    p = t.forward(X, nlevels=3, include_scale=True)
    want = to.forward(X.astype(np.float64), nlevels=3, include_scale=True)
    assert_pyramids_close(p, want, XFM_TOL, same_dtype=False)
    gain = rs.uniform(0.2, 1.5, (6, 3))
    assert_close(t.inverse(p, gain), to.inverse(want, gain), INV_TOL, 'inverse with gains')


def test_inverse_of_a_device_pyramid_never_touches_the_host():
    """A device-resident pyramid goes back through the inverse (and through registration /
    re-sampling) without any of its levels being copied to the host: the NumPy attributes are
    only materialised when somebody reads them."""
    from dtcwt_amd.hip import Transform1d, Transform3d
    from dtcwt_amd import registration
    rs = np.random.RandomState(3)
    X = rs.standard_normal((256, 320)).astype(np.float32)
    t = Transform2d()
    p = t.forward(X, nlevels=5)
    z = t.inverse(p, device_output=True)
    assert isinstance(z, DeviceArray) and p._host == {} and p.nlevels == 5
    q = t.forward(np.roll(X, 1, axis=0), nlevels=5)
    registration.estimatereg(p, q, device_output=True)
    w = registration.warptransform(p, np.zeros((16, 20, 6)), [2, 3])
    assert p._host == {} and q._host == {} and w._host == {}
    V = rs.standard_normal((48, 40, 56)).astype(np.float32)
    t3 = Transform3d()
    p3 = t3.forward(V, nlevels=2)
    t3.inverse(p3, device_output=True)
    assert p3._host == {}
    t1 = Transform1d()
    p1 = t1.forward(rs.standard_normal(128).astype(np.float32), nlevels=3)
    t1.inverse(p1, device_output=True)
    assert p1._host == {}
    assert_close(z.get(), X, INV_TOL, 'PR')


# ---- the two-launch levels of generic2d.hip (every dtype / wavelet without a fused plan) ----
GENERIC_WAVES = [w for w in WAVES if 'bp' not in w[0]]


@pytest.mark.parametrize('bn,qn', GENERIC_WAVES)
@pytest.mark.parametrize('shape', [(96, 128), (97, 123), (130, 70), (200, 88), (64, 1030), (520, 300)])
def test_float64_levels_vs_oracle(bn, qn, shape):
    rs = np.random.RandomState(23)
    X = rs.standard_normal(shape)
    t, to = Transform2d(bn, qn), o.Transform2d(biort(bn), qshift(qn))
    for nl in (1, 2, 4):
        want = to.forward(X, nlevels=nl, include_scale=True)
        p = t.forward(X, nlevels=nl, include_scale=True)
        assert p.lowpass.dtype == np.float64 and p.highpasses[0].dtype == np.complex128
        assert_pyramids_close(p, want, F64_TOL)
        gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.25)
        for g in (None, gm):
            assert_close(t.inverse(want, g), to.inverse(want, g), 1e-11, 'inverse f64 nl=%d' % nl)


@pytest.mark.parametrize('two_pass', [False, True])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('bn,qn', [('near_sym_a', 'qshift_a'), ('near_sym_b', 'qshift_d'), ('antonini', 'qshift_06'),
                                   ('legall', 'qshift_c'), ('near_sym_b', 'qshift_b')])
def test_two_launch_level_matches_filter_by_filter(dtype, bn, qn, two_pass, monkeypatch):
    """dtcwt_hip_level2d_forward / _inverse (the one-launch form, and the two-launch form it
    replaced, DTCWT_HIP_TWO_PASS=1) against the same level built from the public colfilter /
    coldfilt / colifilt + q2c / c2q launches, batch of 3, padded and cropped."""
    from dtcwt_amd.hip import lowlevel as ll
    if two_pass:
        monkeypatch.setenv('DTCWT_HIP_TWO_PASS', '1')
    ctx = default_context()
    rs = np.random.RandomState(5)
    h0o, g0o, h1o, g1o = biort(bn)[:4]
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = qshift(qn)[:8]
    tol = 1e-6 if dtype == np.float32 else 1e-13
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    X = ctx.to_device(rs.standard_normal((3, 91, 118)).astype(dtype))
    # level 1, odd rows extended by one (pad (0, 1))
    got = ll.level2d_forward(X, 0, (0, 1), (0, 0), h0o, h1o)
    assert got is not None
    Lo, Hi = ll.axis_colfilter2(X, h0o, h1o, axis=1, pad=(0, 1))
    LoLo, LoHi = ll.axis_colfilter2(Lo, h0o, h1o, axis=2)
    HiLo, HiHi = ll.axis_colfilter2(Hi, h0o, h1o, axis=2)
    y = DeviceArray(ctx, (3, 46, 59, 6), cdt)
    ll.q2c(LoHi, y, 2, 3); ll.q2c(HiLo, y, 0, 5); ll.q2c(HiHi, y, 1, 4)
    assert_close(got[0].get(), LoLo.get(), tol, 'LoLo level 1')
    assert_close(got[1].get(), y.get(), tol, 'Yh level 1')
    # level 2 on the 92 x 118 lowpass: columns padded by one each side to a multiple of 4
    lo, hi = (h0b, h0a), (h1b, h1a)
    got2 = ll.level2d_forward(LoLo, 1, (0, 0), (1, 1), lo, hi)
    assert got2 is not None
    Lo, Hi = ll.axis_coldfilt2(LoLo, lo, hi, axis=1)
    L2, LH = ll.axis_coldfilt2(Lo, lo, hi, axis=2, pad=(1, 1))
    HL, HH = ll.axis_coldfilt2(Hi, lo, hi, axis=2, pad=(1, 1))
    y2 = DeviceArray(ctx, (3, 23, 30, 6), cdt)
    ll.q2c(LH, y2, 2, 3); ll.q2c(HL, y2, 0, 5); ll.q2c(HH, y2, 1, 4)
    assert_close(got2[0].get(), L2.get(), tol, 'LoLo level 2')
    assert_close(got2[1].get(), y2.get(), tol, 'Yh level 2')
    # inverse of level 2 with gains and the column crop that undoes the padding
    g = rs.uniform(0.4, 1.3, size=6)
    lo, hi = (g0b, g0a), (g1b, g1a)
    Z = ll.level2d_inverse(L2, y2, 1, g, 0, 1, lo, hi)
    assert Z is not None and Z.shape == (3, 92, 118)
    lh, hl, hh = ll.c2q(y2, 0, 5, g[0], g[5]), ll.c2q(y2, 2, 3, g[2], g[3]), ll.c2q(y2, 1, 4, g[1], g[4])
    y1 = ll.axis_colifilt_sum2(L2, lh, lo, hi, axis=1)
    yb = ll.axis_colifilt_sum2(hl, hh, lo, hi, axis=1)
    want = ll.axis_colifilt_sum2(y1, yb, lo, hi, axis=2, crop=(1, 1))
    assert_close(Z.get(), want.get(), tol, 'inverse level 2')
    # inverse of level 1
    Z1 = ll.level2d_inverse(Z, y, 0, g, 0, 0, g0o, g1o)
    assert Z1 is not None
    lh, hl, hh = ll.c2q(y, 0, 5, g[0], g[5]), ll.c2q(y, 2, 3, g[2], g[3]), ll.c2q(y, 1, 4, g[1], g[4])
    y1 = ll.axis_colfilter_sum2(Z, lh, g0o, g1o, axis=1)
    yb = ll.axis_colfilter_sum2(hl, hh, g0o, g1o, axis=1)
    want = ll.axis_colfilter_sum2(y1, yb, g0o, g1o, axis=2)
    assert_close(Z1.get(), want.get(), tol, 'inverse level 1')


def test_two_launch_level_declines_what_it_cannot_do():
    from dtcwt_amd.hip import lowlevel as ll
    ctx = default_context()
    X = ctx.to_device(np.zeros((1, 8, 8)))
    h0o, g0o, h1o, g1o = biort('near_sym_a')[:4]
    assert ll.level2d_forward(X, 0, (0, 0), (0, 0), h0o, h1o) is None            # plane too small
    X = ctx.to_device(np.zeros((1, 128, 128)))
    assert ll.level2d_forward(X, 0, (0, 0), (0, 0), np.ones(4) / 4, np.ones(4) / 4) is None   # even-length level 1
    assert ll.level2d_forward(X, 0, (0, 0), (0, 0), np.ones(23) / 23, h1o) is None          # longer than the buckets


def test_plan_graph_replays_forward_and_inverse():
    """dtcwt_hip_plan2d_capture: the level loops as a hipGraph on fixed buffers give what the plain
    launches give, and follow new contents of the input buffer."""
    ctx = default_context()
    t = Transform2d()
    plan = t.plan(1, 256, 320, 3)
    rs = np.random.RandomState(3)
    X = ctx.to_device(rs.standard_normal((1, 256, 320)).astype(np.float32))
    Yl, Yh, _ = plan.forward(X, False)
    Z = plan.inverse(Yl, Yh, None)
    want_lo, want_hi, want_z = Yl.get(), [y.get() for y in Yh], Z.get()
    Yl2 = DeviceArray(ctx, Yl.shape, np.float32)
    Yh2 = [DeviceArray(ctx, y.shape, np.complex64) for y in Yh]
    Z2 = DeviceArray(ctx, Z.shape, np.float32)
    g = plan.capture(X, Yl2, Yh2, Z2)
    g.launch()
    assert np.array_equal(Yl2.get(), want_lo) and np.array_equal(Z2.get(), want_z)
    assert all(np.array_equal(a.get(), b) for a, b in zip(Yh2, want_hi))
    X.set(rs.standard_normal((1, 256, 320)).astype(np.float32))       # same buffer, new image
    g.launch()
    assert_close(Z2.get(), X.get(), INV_TOL, 'graph replay on new data')


@pytest.mark.parametrize('dt', [np.uint8, np.int8, np.uint16, np.int16, np.int32, np.uint32, np.int64, np.uint64, np.bool_])
def test_integer_images_are_widened_on_the_device(dt):
    """Integer / bool input is float64 for the transform (dtcwt/utils.py:98-105); it is uploaded in its
    own width and widened by dtcwt_hip_to_float: same pyramid as for the float64 copy of the image."""
    rs = np.random.RandomState(9)
    if dt == np.bool_:
        X = rs.uniform(size=(70, 90)) > 0.5
    else:
        info = np.iinfo(dt)
        X = rs.randint(max(info.min, -2 ** 62), min(info.max, 2 ** 62), size=(70, 90), dtype=np.int64).astype(dt) \
            if dt != np.uint64 else rs.randint(0, 2 ** 62, size=(70, 90), dtype=np.int64).astype(dt) * 3
    t = Transform2d()
    p, q = t.forward(X, nlevels=2), t.forward(X.astype(np.float64), nlevels=2)
    assert p.lowpass.dtype == np.float64
    assert np.array_equal(p.lowpass, q.lowpass)
    assert all(np.array_equal(a, b) for a, b in zip(p.highpasses, q.highpasses))
    # strided / byte-swapped views still arrive right (host fallback or contiguous copy)
    Xs = np.asfortranarray(X)
    assert np.array_equal(t.forward(Xs, nlevels=1).lowpass, t.forward(X.astype(np.float64), nlevels=1).lowpass)


def test_unpack_device_pyramid_and_env_backend():
    """utils.unpack(p, 'hip') on a DEVICE pyramid hands out the resident buffers (no host copy), 'numpy' the
    arrays (dtcwt/utils.py:9-42); DTCWT_BACKEND selects the backend of a fresh process and the transform
    then runs on the GPU (dtcwt/__init__.py:133-143)."""
    import os
    import subprocess
    import sys
    from dtcwt_amd.utils import unpack
    m = _mandrill()
    t = Transform2d()
    p = t.forward(m, nlevels=2, include_scale=True)
    yl, yh, ys = unpack(p, 'hip')
    assert isinstance(yl, DeviceArray) and all(isinstance(y, DeviceArray) for y in yh) and len(ys) == 2
    assert p._host == {}                                   # nothing was copied to the host
    nl, nh, ns = unpack(p, 'numpy')
    assert isinstance(nl, np.ndarray) and np.array_equal(nl, yl.get()) and np.array_equal(nh[1], yh[1].get())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import numpy as np, dtcwt_amd as d; assert d.backend_name == "hip"; t = d.Transform2d(); '
            'from dtcwt_amd.hip import Transform2d as T; assert isinstance(t, T); '
            'p = t.forward(np.ones((64, 64), np.float32), nlevels=2); print(p.lowpass.shape)')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=root,
                       env=dict(os.environ, DTCWT_BACKEND='hip'), timeout=300)
    assert r.returncode == 0 and '(32, 32)' in r.stdout, r.stderr[-1500:]


def test_host_edit_of_device_pyramid_is_honoured():
    """The reference's idiom p.highpasses[l][...] = 0 before inverse(): once a NumPy view has been handed out it
    is the authoritative copy (the edit used to be ignored silently)."""
    rs = np.random.RandomState(12)
    X = rs.standard_normal((128, 192)).astype(np.float32)
    t, to = Transform2d(), o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    p = t.forward(X, nlevels=3)
    want = to.forward(as_f64(X), nlevels=3)
    p.highpasses[1][:, :, 2] = 0
    p.highpasses[0][10:40] *= 0.5
    hp = list(want.highpasses)
    hp[1] = hp[1].copy(); hp[1][:, :, 2] = 0
    hp[0] = hp[0].copy(); hp[0][10:40] *= 0.5
    z = t.inverse(p)
    assert_close(z, to.inverse(o.Pyramid(want.lowpass, tuple(hp))), INV_TOL)
    # untouched device pyramids keep running from HBM
    q = t.forward(X, nlevels=3)
    assert_close(t.inverse(q), as_f64(X), INV_TOL)
    assert q._host == {}


def test_plan_cache_is_bounded_and_complex_input_is_refused():
    t = Transform2d()
    for k in range(12):
        t.forward(np.zeros((64 + 4 * k, 64), np.float32), nlevels=1)
    assert len(t._plans) <= Transform2d.MAX_PLANS
    t.clear_plans()
    assert len(t._plans) == 0
    with pytest.raises(TypeError):
        t.forward(np.zeros((64, 64), np.complex64), nlevels=1)


@pytest.mark.parametrize('flip', ['g0', 'g1', 'both', 'h'])
def test_user_taps_with_other_filter_phases(flip):
    """Every shipped q-shift set has sum(g0a g0b) > 0 > sum(g1a g1b) (and the analysis pairs likewise), which the
    fused level >= 2 kernels take as compile-time constants; user-supplied taps with the other signs must take the
    run-time path and still agree with the oracle (no perfect reconstruction: the transform is linear either way)."""
    q = [np.array(v, dtype=np.float64) for v in qshift('qshift_a')]
    # (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b)
    if flip in ('g0', 'both'):
        q[3] = -q[3]
    if flip in ('g1', 'both'):
        q[7] = -q[7]
    if flip == 'h':
        q[1] = -q[1]; q[5] = -q[5]
    q = tuple(q)
    rs = np.random.RandomState(23)
    X = rs.standard_normal((192, 256)).astype(np.float32)
    t, to = Transform2d('near_sym_a', q), o.Transform2d(biort('near_sym_a'), q)
    assert t.plan(1, 192, 256, 3) is not None
    want = to.forward(as_f64(X), nlevels=3)
    assert_pyramids_close(t.forward(X, nlevels=3), want, XFM_TOL, same_dtype=False)
    assert_close(t.inverse(cast_pyramid(want, np.float32)), to.inverse(want), INV_TOL, 'inverse, user taps ' + flip)


@pytest.mark.parametrize('shape,nl', [((256, 256), 7), ((1024, 512), 8), ((200, 328), 7), ((64, 48), 5), ((6, 10), 3)])
def test_deep_pyramids_keep_the_fused_plan_for_their_leading_levels(shape, nl):
    """A float32 pyramid deep enough that some level is <= 8 samples wide: the coarse tail runs in float64 (its filters
    reflect several times over the same few samples), the leading levels keep the fused float32 plan (until
    round 5 the WHOLE transform went to the generic float64 path).  Every level against the oracle, include_scale, the
    inverse with a gain mask, and the fused plan really is the one that ran."""
    from dtcwt_amd.hip import transform2d as T
    rs = np.random.RandomState(nl * 7 + shape[0])
    X = rs.standard_normal(shape).astype(np.float32)
    assert T._degenerate(shape[0], shape[1], nl)
    t = Transform2d()
    p = t.forward(X, nlevels=nl, include_scale=True)
    k = T._fused_levels(shape[0], shape[1], nl)
    assert (k > 0) == (min(shape) >= 40)
    if k > 0:
        assert any(key[:4] == (1, shape[0], shape[1], k) for key in t._plans)          # the fused plan of the leading levels
    to = o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    want = to.forward(as_f64(X), nlevels=nl, include_scale=True)
    assert_pyramids_close(p, want, XFM_TOL, same_dtype=False)
    gm = rs.uniform(0.3, 1.4, size=(6, nl))
    assert_close(t.inverse(p, gm), to.inverse(want, gm), INV_TOL, 'inverse with gains')
    assert_close(t.inverse(p), X, INV_TOL, 'PR')
