"""The one-launch levels 1 + 2 (forward) / 2 + 1 (inverse) of the float32 2-D plan -- the marching wavefront programs
of dtcwt_amd/csrc/march2d.hpp -- against the oracle and against the per-level tile programs they replace, at sizes that
put the band and strip boundaries, the mirrored halo lanes and the reflected rows everywhere they can be.

Reference behaviour: dtcwt/numpy/transform2d.py:112-160 (forward levels 1, 2), :242-293 (inverse levels 2, 1).
Also: the page-locked / overlapped host path of the Pyramid (prefetch, dropped pyramids, views outliving them)."""
import gc

import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Transform2d, Pyramid, DeviceArray, default_context
from tests._hip import assert_close, assert_pyramids_close, as_f64, XFM_TOL, INV_TOL

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64), (256, 320), (96, 1036), (520, 236), (1024, 232), (200, 464), (44, 940)]


def _fwd_inv(X, nl, gm=None):
    t = Transform2d()
    assert t.plan(X.shape[0] if X.ndim == 3 else 1, X.shape[-2], X.shape[-1], nl) is not None
    p = t.forward(X, nlevels=nl)
    ys = [np.array(y) for y in p.highpasses]
    yl = np.array(p.lowpass)
    z = np.array(t.inverse(Pyramid(yl, tuple(ys)), gm))
    return yl, ys, z


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('band', [None, 8, 24])
def test_march_matches_tile_programs_and_oracle(shape, band, monkeypatch):
    rs = np.random.RandomState(5)
    X = rs.standard_normal(shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
    monkeypatch.setenv('DTCWT_HIP_MARCH', '0')
    yl0, ys0, z0 = _fwd_inv(X, nl, gm)
    monkeypatch.setenv('DTCWT_HIP_MARCH', '1')
    if band:
        monkeypatch.setenv('DTCWT_HIP_MARCH_BAND', str(band))
    t = Transform2d()
    f12, i21 = t.plan(1, shape[0], shape[1], nl).launches()
    assert f12 and i21
    yl1, ys1, z1 = _fwd_inv(X, nl, gm)
    # the two paths sum in different orders (mirror pairs first, rows before columns): not bit-identical, both right
    assert_close(yl1, yl0, 1e-6, 'Yl march vs tiles')
    for a, b in zip(ys1, ys0):
        assert_close(a, b, 1e-6, 'Yh march vs tiles')
    assert_close(z1, z0, 1e-6, 'inverse march vs tiles')
    to = o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    want = to.forward(as_f64(X), nlevels=nl)
    assert_close(yl1, want.lowpass, XFM_TOL, 'Yl')
    for l, (a, b) in enumerate(zip(ys1, want.highpasses)):
        assert_close(a, b, XFM_TOL, 'Yh[%d]' % l)
    assert_close(z1, to.inverse(want, gm), INV_TOL, 'inverse')


L1_SHAPES = [(64, 64), (256, 320), (96, 1036), (520, 236), (1024, 232), (200, 464), (44, 940), (62, 468), (40, 224)]


@pytest.mark.parametrize('shape', L1_SHAPES)
@pytest.mark.parametrize('bn,qn', [('near_sym_b', 'qshift_b'), ('antonini', 'qshift_a'), ('near_sym_b', 'qshift_d')])
@pytest.mark.parametrize('band', [None, 20, 40])
def test_level1_march_matches_tile_programs_and_oracle(shape, bn, qn, band, monkeypatch):
    """Level 1 alone as a marching launch (march2d_l1.hpp: near_sym_b's 13 / 19 taps, antonini's 9 / 7) against the tile
    programs it replaces and against the oracle, at sizes that put strip and band boundaries, the three mirrored halo lanes
    and the reflected rows everywhere (rows not a multiple of 4 included: level 1 needs even rows only), with gains."""
    rs = np.random.RandomState(15)
    X = rs.standard_normal(shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
    tt, tm = Transform2d(bn, qn, program='tiles'), Transform2d(bn, qn, program='march')
    if band:
        monkeypatch.setenv('DTCWT_HIP_MARCH_BAND', str(band))
    assert tt.plan(1, shape[0], shape[1], nl).level1_march() == (False, False)
    assert tm.plan(1, shape[0], shape[1], nl).level1_march() == (True, True)
    assert tm.plan(1, shape[0], shape[1], nl).launches() == (False, False)        # the fused levels 1 + 2 are not built for these
    p0, p1 = tt.forward(X, nlevels=nl, include_scale=True), tm.forward(X, nlevels=nl, include_scale=True)
    assert_close(p1.lowpass, p0.lowpass, 1e-6, 'Yl march vs tiles')
    for a, b in zip(p1.highpasses, p0.highpasses):
        assert_close(a, b, 1e-6, 'Yh march vs tiles')
    assert_close(p1.scales[0], p0.scales[0], 1e-6, 'level-1 lowpass march vs tiles')
    to = o.Transform2d(biort(bn), qshift(qn))
    want = to.forward(as_f64(X), nlevels=nl, include_scale=True)
    assert_pyramids_close(p1, want, XFM_TOL, same_dtype=False)
    pw = Pyramid(np.asarray(want.lowpass, np.float32), tuple(np.asarray(y, np.complex64) for y in want.highpasses))
    z0, z1 = tt.inverse(pw, gm), tm.inverse(pw, gm)
    assert_close(z1, z0, 1e-6, 'inverse march vs tiles')
    assert_close(z1, to.inverse(want, gm), INV_TOL, 'inverse')
    assert_close(tm.inverse(p1), X, INV_TOL, 'reconstruction')


def test_level1_march_on_a_batch_and_where_it_is_chosen(monkeypatch):
    monkeypatch.delenv('DTCWT_HIP_MARCH', raising=False)
    t = Transform2d('near_sym_b', 'qshift_b')
    assert t.plan(1, 512, 512, 3).level1_march() == (False, False)             # below the crossover: the tile programs
    assert t.plan(1, 2048, 2048, 4).level1_march() == (True, True)
    assert t.plan(1, 2050, 2048, 4).level1_march() == (True, True)             # even rows suffice: level 2 pads
    assert t.plan(1, 2047, 2048, 4).level1_march() == (False, False)           # odd-size extension: tiles
    assert t.plan(1, 2048, 2046, 4).level1_march() == (False, False)           # columns in fours
    assert Transform2d('near_sym_b_bp', 'qshift_b_bp').plan(1, 2048, 2048, 3).level1_march() == (False, False)
    assert Transform2d().plan(1, 2048, 2048, 4).level1_march() == (False, False)       # near_sym_a: the fused levels 1 + 2
    rs = np.random.RandomState(16)
    X = rs.standard_normal((5, 160, 476)).astype(np.float32)
    tm = Transform2d('near_sym_b', 'qshift_b', program='march')
    to = o.Transform2d(biort('near_sym_b'), qshift('qshift_b'))
    p = tm.forward_channels(X, 'nhw', nlevels=3)
    for b in (0, 4):
        want = to.forward(as_f64(X[b]), nlevels=3)
        assert_close(p.lowpass[b], want.lowpass, XFM_TOL, 'Yl')
        for l in range(3):
            assert_close(p.highpasses[l][b], want.highpasses[l], XFM_TOL, 'Yh[%d]' % l)
    single = tm.forward(X[3], nlevels=3)
    assert np.array_equal(single.highpasses[0], p.highpasses[0][3]) and np.array_equal(single.lowpass, p.lowpass[3])
    assert_close(tm.inverse_channels(p, 'nhw'), X, INV_TOL, 'reconstruction')


@pytest.mark.parametrize('shape', [(256, 320), (96, 1036), (44, 940)])
def test_march_with_include_scale(shape):
    """`scales` no longer sends levels 1 + 2 of the forward back to the tile programs: the one launch stores the level-1
    lowpass as well (two 16-byte stores per step)."""
    rs = np.random.RandomState(8)
    X = rs.standard_normal((2,) + shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    tm, tt = Transform2d(program='march'), Transform2d(program='tiles')
    assert tm.plan(2, shape[0], shape[1], nl).launches() == (True, True)
    pm, pt = tm.forward_channels(X, 'nhw', nlevels=nl, include_scale=True), tt.forward_channels(X, 'nhw', nlevels=nl, include_scale=True)
    to = o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    for l in range(nl):
        assert_close(pm.scales[l], pt.scales[l], 1e-6, 'scale %d march vs tiles' % l)
        assert_close(pm.highpasses[l], pt.highpasses[l], 1e-6, 'Yh[%d] march vs tiles' % l)
    # bit-identical to the launch without `scales`
    pn = tm.forward_channels(X, 'nhw', nlevels=nl)
    assert np.array_equal(pn.lowpass, pm.lowpass) and all(np.array_equal(a, b) for a, b in zip(pn.highpasses, pm.highpasses))
    want = to.forward(as_f64(X[1]), nlevels=nl, include_scale=True)
    for l in range(nl):
        assert_close(pm.scales[l][1], want.scales[l], XFM_TOL, 'scale %d' % l)


@pytest.mark.parametrize('shape', SHAPES + [(48, 212), (1024, 208)])
@pytest.mark.parametrize('bn,qn', [('near_sym_a', 'qshift_b'), ('near_sym_a', 'qshift_d'), ('legall', 'qshift_b'), ('legall', 'qshift_d')])
@pytest.mark.parametrize('band', [None, 8, 24])
def test_forward_pair_matches_tile_programs_and_oracle(shape, bn, qn, band, monkeypatch):
    """Levels 1 + 2 of the forward as a marching PAIR of wavefronts (march2d_pair.hpp: k_fwd12p; near_sym_a with the 14- / 18-tap
    q-shift sets) against the tile programs and the oracle, at sizes that put the strip boundaries (224- / 216-column strips), band
    boundaries, mirrored halo lanes and reflected rows everywhere."""
    rs = np.random.RandomState(25)
    X = rs.standard_normal(shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    tt, tm = Transform2d(bn, qn, program='tiles'), Transform2d(bn, qn, program='march')
    if band:
        monkeypatch.setenv('DTCWT_HIP_MARCH_BAND', str(band))
    assert tm.plan(1, shape[0], shape[1], nl).launches()[0] is True
    p0, p1 = tt.forward(X, nlevels=nl), tm.forward(X, nlevels=nl)
    assert_close(p1.lowpass, p0.lowpass, 1e-6, 'Yl pair vs tiles')
    for a, b in zip(p1.highpasses, p0.highpasses):
        assert_close(a, b, 1e-6, 'Yh pair vs tiles')
    want = o.Transform2d(biort(bn), qshift(qn)).forward(as_f64(X), nlevels=nl)
    assert_pyramids_close(p1, want, XFM_TOL, same_dtype=False)
    assert_close(tm.inverse(p1), X, INV_TOL, 'reconstruction')
    monkeypatch.setenv('DTCWT_HIP_MARCH_PARTS', str(255 & ~12))  # without the pairs (bits 4, 8): back to level 1 alone + a level-2 tile launch
    assert tm.plan(1, shape[0], shape[1], nl).launches()[0] is False


@pytest.mark.parametrize('shape', SHAPES + [(48, 212), (1024, 208)])
@pytest.mark.parametrize('bn,qn', [('near_sym_a', 'qshift_b'), ('near_sym_a', 'qshift_d'), ('legall', 'qshift_b')])
@pytest.mark.parametrize('band', [None, 8, 24])
def test_inverse_pair_matches_tile_programs_and_oracle(shape, bn, qn, band, monkeypatch):
    """Levels 2 + 1 of the inverse as a marching PAIR of wavefronts (march2d_ipair.hpp: k_inv21p; the 14- / 18-tap q-shift sets with
    the 7 / 5-tap synthesis filters of near_sym_a and legall's 3 / 5) against the tile programs and the oracle, with a gain mask, at
    sizes that put the strip boundaries (224- / 216-column strips), band boundaries, mirrored lanes and reflected rows everywhere."""
    rs = np.random.RandomState(27)
    X = rs.standard_normal(shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
    tt, tm = Transform2d(bn, qn, program='tiles'), Transform2d(bn, qn, program='march')
    if band:
        monkeypatch.setenv('DTCWT_HIP_MARCH_BAND', str(band))
    assert tm.plan(1, shape[0], shape[1], nl).launches()[1] is True
    p0 = tt.forward(X, nlevels=nl)
    pyr = Pyramid(np.array(p0.lowpass), tuple(np.array(y) for y in p0.highpasses))
    z0, z1 = np.array(tt.inverse(pyr, gm)), np.array(tm.inverse(pyr, gm))
    assert_close(z1, z0, 1e-6, 'inverse pair vs tiles')
    to = o.Transform2d(biort(bn), qshift(qn))
    want = to.forward(as_f64(X), nlevels=nl)
    assert_close(z1, to.inverse(want, gm), INV_TOL, 'inverse')
    assert_close(tm.inverse(tm.forward(X, nlevels=nl)), X, INV_TOL, 'reconstruction')
    monkeypatch.setenv('DTCWT_HIP_MARCH_PARTS', str(255 & ~12))  # without the pairs: back to a level-2 tile launch + a level-1 tile launch
    assert tm.plan(1, shape[0], shape[1], nl).launches()[1] is False


@pytest.mark.parametrize('shape', [(64, 64), (256, 320), (96, 1036), (520, 236), (1024, 232), (44, 940)])
@pytest.mark.parametrize('bn', ['near_sym_a', 'legall'])
def test_headline_inverse_as_a_pair_is_bit_identical(shape, bn, monkeypatch):
    """One transform at a time on the whole device (up to 4096^2) runs levels 2 + 1 of the inverse as a marching pair of wavefronts
    (k_inv21p<7, 5, 10>) instead of k_inv21m: the same sums in the same order, whatever the band heights -- not a bit differs."""
    monkeypatch.setenv('DTCWT_HIP_MARCH', '1')
    rs = np.random.RandomState(29)
    X = rs.standard_normal(shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
    t = Transform2d(bn, 'qshift_a' if bn == 'near_sym_a' else 'qshift_06')
    assert t.plan(1, shape[0], shape[1], nl).launches()[1] is True
    p = t.forward(X, nlevels=nl)
    pyr = Pyramid(np.array(p.lowpass), tuple(np.array(y) for y in p.highpasses))
    out = {}
    for arm, parts in (('0', 255 & ~128), ('1', 255 | 256)):        # DTCWT_HIP_MARCH_PARTS: never / always as a pair (bits 128, 256)
        monkeypatch.setenv('DTCWT_HIP_MARCH_PARTS', str(parts))
        assert ('k_inv21p' in t.plan(1, shape[0], shape[1], nl).describe()) == (arm == '1')
        out[arm] = np.array(t.inverse(pyr, gm))
    assert np.array_equal(out['0'], out['1'])
    monkeypatch.delenv('DTCWT_HIP_MARCH_PARTS')
    assert np.array_equal(np.array(t.inverse(pyr, gm)), out['0'])           # the library's own choice
    to = o.Transform2d(biort(bn), qshift('qshift_a' if bn == 'near_sym_a' else 'qshift_06'))
    assert_close(out['1'], to.inverse(to.forward(as_f64(X), nlevels=nl), gm), INV_TOL, 'inverse')


def test_pairs_on_a_share_of_the_compute_units():
    """On a partition context k_fwd12p<5, 7, 14> and k_inv21p<7, 5, 14> are the builds for three wavefronts per SIMD: the same arithmetic as on the whole
    device, bit for bit, and right."""
    from dtcwt_amd.hip import Context
    rs = np.random.RandomState(33)
    X = rs.standard_normal((520, 696)).astype(np.float32)
    gm = rs.uniform(0.3, 1.4, size=(6, 3))
    tw = Transform2d('near_sym_a', 'qshift_b', program='march')
    tq = Transform2d('near_sym_a', 'qshift_b', program='march', ctx=Context(0, partition=(1, 4)))
    assert tq.plan(1, 520, 696, 3).launches() == (True, True)
    p, pq = tw.forward(X, nlevels=3), tq.forward(X, nlevels=3)              # k_fwd12p<5, 7, 14> for two / three wavefronts per SIMD
    assert np.array_equal(np.array(p.lowpass), np.array(pq.lowpass))
    assert all(np.array_equal(np.array(a), np.array(b)) for a, b in zip(p.highpasses, pq.highpasses))
    pyr = Pyramid(np.array(p.lowpass), tuple(np.array(y) for y in p.highpasses))
    zw, zq = np.array(tw.inverse(pyr, gm)), np.array(tq.inverse(pyr, gm))
    assert np.array_equal(zw, zq)
    to = o.Transform2d(biort('near_sym_a'), qshift('qshift_b'))
    assert_close(zq, to.inverse(to.forward(as_f64(X), nlevels=3), gm), INV_TOL, 'inverse on a share')


@pytest.mark.parametrize('shape', [(64, 64), (256, 320), (96, 1036), (520, 236), (1024, 232), (200, 464), (48, 212)])
@pytest.mark.parametrize('bn,qn', [('near_sym_b', 'qshift_b'), ('near_sym_b', 'qshift_d'), ('antonini', 'qshift_b')])
@pytest.mark.parametrize('band', [None, 8, 24])
def test_level2_inverse_march_matches_tile_program_and_oracle(shape, bn, qn, band, monkeypatch):
    """Level 2 of the inverse alone as a march (march2d_ipair.hpp: k_inv2m, the level-2 wavefront of the inverse pair storing its
    groups of Z1 rows) for the 14- / 18-tap q-shift sets whose level 1 no pair takes: against the tile program (DTCWT_HIP_MARCH_PARTS without bit 32)
    and the oracle, with a gain mask."""
    rs = np.random.RandomState(35)
    X = rs.standard_normal(shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
    tm = Transform2d(bn, qn, program='march')
    if band:
        monkeypatch.setenv('DTCWT_HIP_MARCH_BAND', str(band))
    p = tm.forward(X, nlevels=nl)
    pyr = Pyramid(np.array(p.lowpass), tuple(np.array(y) for y in p.highpasses))
    z1 = np.array(tm.inverse(pyr, gm))
    monkeypatch.setenv('DTCWT_HIP_MARCH_PARTS', str(255 & ~32))
    z0 = np.array(tm.inverse(pyr, gm))
    assert not np.array_equal(z0, z1)            # two programs: they agree to rounding, not to the bit
    assert_close(z1, z0, 1e-6, 'level-2 inverse march vs tile program')
    to = o.Transform2d(biort(bn), qshift(qn))
    assert_close(z1, to.inverse(to.forward(as_f64(X), nlevels=nl), gm), INV_TOL, 'inverse')


@pytest.mark.parametrize('shape', [(64, 64), (256, 320), (96, 1036), (520, 236), (1024, 232), (200, 464), (48, 212)])
@pytest.mark.parametrize('bn,qn', [('near_sym_b', 'qshift_b'), ('near_sym_b', 'qshift_d'), ('antonini', 'qshift_b')])
@pytest.mark.parametrize('band', [None, 8, 24])
def test_level2_forward_march_matches_tile_program_and_oracle(shape, bn, qn, band, monkeypatch):
    """Level 2 of the forward alone as a march (march2d_pair.hpp: k_fwd2m, the level-2 wavefront of the forward pair fed from memory)
    for the 14- / 18-tap q-shift sets whose level 1 no pair takes: against the tile program (DTCWT_HIP_MARCH_PARTS without bit 16) and the oracle."""
    rs = np.random.RandomState(37)
    X = rs.standard_normal(shape).astype(np.float32)
    nl = 2 if min(shape) < 160 else 3
    tm = Transform2d(bn, qn, program='march')
    if band:
        monkeypatch.setenv('DTCWT_HIP_MARCH_BAND', str(band))
    p1 = tm.forward(X, nlevels=nl)
    yl1, ys1 = np.array(p1.lowpass), [np.array(y) for y in p1.highpasses]
    monkeypatch.setenv('DTCWT_HIP_MARCH_PARTS', str(255 & ~16))
    p0 = tm.forward(X, nlevels=nl)
    assert np.array_equal(ys1[0], np.array(p0.highpasses[0]))            # level 1 is the same launch either way
    assert not np.array_equal(ys1[1], np.array(p0.highpasses[1]))        # two programs: they agree to rounding, not to the bit
    assert_close(yl1, p0.lowpass, 1e-6, 'Yl level-2 march vs tile program')
    for a, b in zip(ys1, p0.highpasses):
        assert_close(a, b, 1e-6, 'Yh level-2 march vs tile program')
    want = o.Transform2d(biort(bn), qshift(qn)).forward(as_f64(X), nlevels=nl)
    assert_close(yl1, want.lowpass, XFM_TOL, 'Yl')
    for l, (a, b) in enumerate(zip(ys1, want.highpasses)):
        assert_close(a, b, XFM_TOL, 'Yh[%d]' % l)
    monkeypatch.delenv('DTCWT_HIP_MARCH_PARTS')
    ps = tm.forward(X, nlevels=nl, include_scale=True)                   # with `scales`: the same launches, LoLo2 into its scale buffer
    assert np.array_equal(np.array(ps.highpasses[1]), ys1[1]) and np.array_equal(np.array(ps.lowpass), yl1)


def test_forward_pair_on_a_batch():
    rs = np.random.RandomState(26)
    X = rs.standard_normal((5, 128, 424)).astype(np.float32)
    tm = Transform2d('near_sym_a', 'qshift_b', program='march')
    to = o.Transform2d(biort('near_sym_a'), qshift('qshift_b'))
    assert tm.plan(5, 128, 424, 3).launches() == (True, True)
    # where the library chooses by itself: batches and shared devices, not one image alone (profiles/r05/pair_forward.txt)
    ta = Transform2d('near_sym_a', 'qshift_b')
    assert ta.plan(1, 4096, 4096, 4).launches()[0] is False and ta.plan(64, 1024, 1024, 4).launches()[0] is True
    pl = ta.plan(1, 4096, 4096, 4); pl.set_concurrency(4); assert pl.launches()[0] is True; pl.set_concurrency(1)
    p = tm.forward_channels(X, 'nhw', nlevels=3)
    for b in (0, 4):
        want = to.forward(as_f64(X[b]), nlevels=3)
        assert_close(p.lowpass[b], want.lowpass, XFM_TOL, 'Yl')
        for l in range(3):
            assert_close(p.highpasses[l][b], want.highpasses[l], XFM_TOL, 'Yh[%d]' % l)
    single = tm.forward(X[2], nlevels=3)
    assert all(np.array_equal(a, b[2]) for a, b in zip(single.highpasses, p.highpasses))
    assert_close(tm.inverse_channels(p, 'nhw'), X, INV_TOL, 'reconstruction')


def test_march_on_a_batch_and_other_level1_filters(monkeypatch):
    monkeypatch.setenv('DTCWT_HIP_MARCH', '1')              # wherever it applies, not only where it pays
    rs = np.random.RandomState(6)
    X = rs.standard_normal((3, 128, 248)).astype(np.float32)
    for bn in ('near_sym_a', 'legall', 'antonini'):       # up to 7 taps (legall's 3-tap g0o as a centred zero-padded 7-tap one)
        t, to = Transform2d(bn, 'qshift_a'), o.Transform2d(biort(bn), qshift('qshift_a'))
        f12, i21 = t.plan(3, 128, 248, 3).launches()
        assert f12 == (bn != 'antonini') and i21 == (bn != 'antonini')
        p = t.forward_channels(X, 'nhw', nlevels=3)
        for b in range(3):
            want = to.forward(as_f64(X[b]), nlevels=3)
            assert_close(p.lowpass[b], want.lowpass, XFM_TOL, '%s Yl' % bn)
            for l in range(3):
                assert_close(p.highpasses[l][b], want.highpasses[l], XFM_TOL, '%s Yh[%d]' % (bn, l))
        assert_close(t.inverse_channels(p, 'nhw'), X, INV_TOL, '%s reconstruction' % bn)


def test_march_is_used_where_it_pays(monkeypatch):
    """Unset, DTCWT_HIP_MARCH leaves the choice to the plan: the one-launch form needs ~3.1 M useful pixels per call on
    the whole device (a marching launch takes ~30 us however small the image; profiles/r04/ab_march_sizes.txt, profiles/r05/march_sizes_dir.txt), fewer with
    other transforms in flight beside it (profiles/r04/hint_sizes*.txt)."""
    from dtcwt_amd.hip import Context
    monkeypatch.delenv('DTCWT_HIP_MARCH', raising=False)
    t = Transform2d()
    assert t.plan(1, 512, 512, 3).launches() == (False, False)
    assert t.plan(1, 1536, 1536, 4).launches() == (False, False)
    assert t.plan(1, 1792, 1792, 4).launches() == (True, True)
    assert t.plan(16, 512, 512, 4).launches() == (False, False)          # 4.2 M pixels, a third of the lanes idle
    assert t.plan(1, 2048, 2048, 4).launches() == (True, True)
    assert t.plan(32, 512, 512, 4).launches() == (True, True)
    pl = t.plan(1, 1536, 1536, 4)
    assert pl.launches() == (False, False)
    pl.set_concurrency(4)                                                # others in flight beside it: from 2.2 M pixels
    assert pl.launches() == (True, True)
    pl.set_concurrency(1)
    tq = Transform2d(ctx=Context(0, partition=(0, 4)))                  # on a quarter of the CUs: from 1.4 M pixels
    assert tq.plan(1, 1536, 1536, 4).launches() == (True, True) and tq.plan(1, 1080, 1920, 4).launches() == (True, True)
    assert tq.plan(1, 1024, 1024, 4).launches() == (False, False)
    monkeypatch.setenv('DTCWT_HIP_MARCH', '1')
    assert t.plan(1, 512, 512, 3).launches() == (True, True)
    monkeypatch.setenv('DTCWT_HIP_MARCH', '0')
    assert t.plan(1, 4096, 4096, 4).launches() == (False, False)


def test_program_pin_is_an_argument_and_wins_over_the_environment(monkeypatch):
    """dtcwt_hip_plan2d_set_program / Transform2d(program=...): a caller that needs a batch and its single images (or a
    forward and the inverse of its pyramid) computed by the same arithmetic pins one program -- bit-identical results
    whatever the call shape -- instead of relying on DTCWT_HIP_MARCH."""
    monkeypatch.delenv('DTCWT_HIP_MARCH', raising=False)
    rs = np.random.RandomState(21)
    Xb = rs.standard_normal((12, 512, 512)).astype(np.float32)
    for prog, want in (('tiles', (False, False)), ('march', (True, True))):
        t = Transform2d(program=prog)
        assert t.plan(1, 512, 512, 3).launches() == want and t.plan(64, 1024, 1024, 3).launches() == want
        pb = t.forward_channels(Xb, 'nhw', nlevels=3)
        for i in (0, 5, 11):
            single = t.forward(Xb[i], nlevels=3)
            assert np.array_equal(pb.lowpass[i], single.lowpass)
            for l in range(3):
                assert np.array_equal(pb.highpasses[l][i], single.highpasses[l]), (prog, i, l)
        monkeypatch.setenv('DTCWT_HIP_MARCH', '0' if prog == 'march' else '1')       # the pin wins
        assert t.plan(1, 512, 512, 3).launches() == want
        monkeypatch.delenv('DTCWT_HIP_MARCH')
    auto = Transform2d()
    assert auto.plan(1, 512, 512, 3).launches() == (False, False) and auto.plan(64, 1024, 1024, 3).launches() == (True, True)
    pl = auto.plan(1, 512, 512, 3)
    pl.set_program('march'); assert pl.launches() == (True, True)
    pl.set_program('auto'); assert pl.launches() == (False, False)
    with pytest.raises(ValueError):
        Transform2d(program='fastest')
    # 'march' where it does not apply: the tile programs, silently
    assert Transform2d(program='march').plan(1, 255, 256, 3).launches() == (False, False)


def test_partition_context_beside_null_stream_work():
    """The stream of a partition context is a blocking stream (hipExtStreamCreateWithCUMask has no flags): it orders itself
    against the legacy NULL stream.  Results must be right either way; what the header documents is the serialisation."""
    import ctypes
    from dtcwt_amd.hip import Context, _lib
    # the HIP runtime this process already has (the library's dependency): dlopen of the same file returns the same handle
    default_context()
    path = next((ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64.so' in ln), None)
    if path is None:
        pytest.skip('no libamdhip64 mapped: cannot enqueue NULL-stream work')
    hip = ctypes.CDLL(path)
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    rs = np.random.RandomState(3)
    X = rs.standard_normal((512, 464)).astype(np.float32)
    want = Transform2d(program='march').forward(X, nlevels=3)
    c = Context(0, partition=(1, 4))
    t = Transform2d(ctx=c, program='march')
    n = 1 << 24
    d = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(n)) == 0
    try:
        for rep in range(4):
            assert hip.hipMemsetAsync(d, rep, ctypes.c_size_t(n), None) == 0          # NULL-stream work in between
            p = t.forward(X, nlevels=3)
            assert hip.hipMemsetAsync(d, rep + 1, ctypes.c_size_t(n), None) == 0
            z = t.inverse(p)
            assert np.array_equal(p.lowpass, want.lowpass) and np.array_equal(p.highpasses[0], want.highpasses[0])
            assert np.abs(z - X).max() < 1e-6 * np.abs(X).max() * 4
    finally:
        hip.hipDeviceSynchronize()
        hip.hipFree(d)


def test_march_is_not_used_where_it_does_not_apply(monkeypatch):
    monkeypatch.setenv('DTCWT_HIP_MARCH', '1')
    t = Transform2d()
    assert t.plan(1, 254, 256, 3).launches() == (False, False)           # level-2 padding (254 % 4)
    assert t.plan(1, 255, 256, 3).launches() == (False, False)           # odd-size extension
    assert Transform2d('near_sym_b', 'qshift_b').plan(1, 256, 256, 3).launches() == (False, False)    # level 1 alone as a march instead
    assert Transform2d('near_sym_a', 'qshift_b').plan(1, 256, 256, 3).launches() == (True, True)      # both directions: marching pairs
    assert Transform2d('near_sym_a', 'qshift_c').plan(1, 256, 256, 3).launches() == (False, False)    # 16 taps: (M - 2) % 4 != 0
    assert Transform2d('antonini', 'qshift_b').plan(1, 256, 256, 3).launches() == (False, False)
    assert t.plan(1, 4096, 4096, 4).launches() == (True, True)
    assert t.plan(64, 1024, 1024, 3).launches() == (True, True)
    # `scales` needs the level-1 lowpass: the forward then keeps its one launch per level, and says so by being right
    rs = np.random.RandomState(7)
    X = rs.standard_normal((256, 256)).astype(np.float32)
    p = t.forward(X, nlevels=3, include_scale=True)
    want = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(as_f64(X), nlevels=3, include_scale=True)
    assert_pyramids_close(p, want, XFM_TOL, same_dtype=False)


def test_contexts_on_shares_of_the_compute_units(monkeypatch):
    monkeypatch.setenv('DTCWT_HIP_MARCH', '1')
    """dtcwt_hip_ctx_create_partition: four contexts, each on a quarter of the CUs, transform four different images
    concurrently; every result matches the oracle, and matches the whole-device context's bit for bit (band heights
    differ -- the plans size their launches for 64 CUs -- the arithmetic of a coefficient does not)."""
    from dtcwt_amd.hip import Context
    from dtcwt_amd.hip._lib import HipError
    rs = np.random.RandomState(12)
    Xs = [rs.standard_normal((512, 696)).astype(np.float32) for _ in range(4)]
    ctxs = [Context(0, partition=(s, 4)) for s in range(4)]
    assert [c.partition for c in ctxs] == [(s, 4) for s in range(4)]
    ts = [Transform2d(ctx=c) for c in ctxs]
    ps = [t.forward(X, nlevels=4) for t, X in zip(ts, Xs)]            # four in flight
    zs = [t.inverse(p) for t, p in zip(ts, ps)]
    to, tw = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')), Transform2d()
    for X, p, z in zip(Xs, ps, zs):
        assert_pyramids_close(p, to.forward(as_f64(X), nlevels=4), XFM_TOL, same_dtype=False)
        assert_close(np.asarray(z), X, INV_TOL, 'reconstruction on a share')
        pw = tw.forward(X, nlevels=4)
        assert np.array_equal(np.asarray(p.lowpass), np.asarray(pw.lowpass))
        assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(p.highpasses, pw.highpasses))
    for bad in ((4, 4), (-1, 2), (0, 0), (0, 17)):
        with pytest.raises(HipError):
            Context(0, partition=bad)
    with pytest.raises(ValueError):
        Context(0, stream=ctxs[0].stream, partition=(0, 2))


# ---- the host path: page-locked pool, overlapped downloads, Pyramid.prefetch ------------------------------------------
def test_prefetched_and_lazy_downloads_agree():
    rs = np.random.RandomState(8)
    X = rs.standard_normal((512, 512)).astype(np.float32)
    t = Transform2d()
    p1 = t.forward(X, nlevels=3)                    # NumPy input: prefetch starts behind the kernels
    p2 = t.forward(default_context().to_device(X), nlevels=3)      # device input: lazy
    assert np.array_equal(p1.lowpass, p2.lowpass)
    for a, b in zip(p1.highpasses, p2.highpasses):
        assert np.array_equal(a, b)
    assert np.array_equal(t.inverse(p1), t.inverse(p2))            # p1 with its downloads read, p2 from the device


def test_dropping_an_unread_prefetched_pyramid_is_safe():
    """The pinned host buffers and the device buffers of a pyramid dropped unread go back to their pools only after
    its downloads have finished: a blocking download of the same size right afterwards must not see them."""
    rs = np.random.RandomState(9)
    t = Transform2d()
    X = rs.standard_normal((1024, 1024)).astype(np.float32)
    Y = rs.standard_normal((1024, 1024)).astype(np.float32)
    want = [np.array(h) for h in t.forward(Y, nlevels=2).highpasses]
    for _ in range(5):
        p = t.forward(X, nlevels=2)                 # prefetch in flight ...
        del p                                       # ... dropped unread
        gc.collect()
        q = t.forward(default_context().to_device(Y), nlevels=2)
        got = [q.hip_highpasses[l].get() for l in range(2)]        # blocking downloads into recycled buffers
        for a, b in zip(got, want):
            assert np.array_equal(a, b)


def test_a_view_may_outlive_its_pyramid_and_inverse_with_pending_downloads():
    rs = np.random.RandomState(10)
    X = rs.standard_normal((768, 512)).astype(np.float32)
    t = Transform2d()
    p = t.forward(X, nlevels=3)
    z = t.inverse(p)                                # downloads still pending: the inverse uses the device buffers
    assert_close(z, X, INV_TOL, 'reconstruction')
    row = p.highpasses[0][5]                        # a view of a pooled page-locked buffer
    keep = row.copy()
    del p
    gc.collect()
    for _ in range(3):
        t.forward(X[::-1].copy(), nlevels=3).highpasses            # would reuse the buffer if the view did not pin it
    assert np.array_equal(row, keep)


def test_pageable_host_buffers(monkeypatch):
    from dtcwt_amd.hip import _lib
    monkeypatch.setattr(_lib.host_pool, 'pinned', False)
    _lib.host_pool.trim()
    rs = np.random.RandomState(11)
    X = rs.standard_normal((512, 640)).astype(np.float32)
    t = Transform2d()
    p = t.forward(X, nlevels=2)
    want = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(as_f64(X), nlevels=2)
    assert_pyramids_close(p, want, XFM_TOL, same_dtype=False)
    _lib.host_pool.trim()


_PAIRS = [('near_sym_a', 'qshift_a'), ('near_sym_a', 'qshift_b'), ('near_sym_a', 'qshift_c'), ('near_sym_a', 'qshift_d'),
          ('near_sym_a', 'qshift_06'), ('near_sym_b', 'qshift_a'), ('near_sym_b', 'qshift_b'), ('near_sym_b', 'qshift_d'),
          ('antonini', 'qshift_a'), ('antonini', 'qshift_b'), ('legall', 'qshift_06'), ('legall', 'qshift_b'),
          ('legall', 'qshift_c'), ('near_sym_b_bp', 'qshift_b_bp')]


@pytest.mark.parametrize('bn,qn', _PAIRS)
@pytest.mark.parametrize('prog', ['march', 'tiles'])
def test_batch_and_its_images_agree_bit_for_bit_for_every_shipped_pair(bn, qn, prog, monkeypatch):
    """with a program pinned, a batch and each of its images transformed alone give the same bits --
    forward (every level, the lowpass) and inverse (with a gain mask) -- for EVERY shipped wavelet pair, whichever kernels
    the pair runs on (fused marches, marching pairs, level-1 / level-2 marches, tile programs, band-pass tiles): band
    heights and job layouts differ between the two calls, the sums and their order do not.  Under 'auto' the library may
    pick different programs for the two call shapes (documented: 2e-7, not bit-identical)."""
    monkeypatch.delenv('DTCWT_HIP_MARCH', raising=False)
    rs = np.random.RandomState(len(bn) * 31 + len(qn))
    Xb = rs.standard_normal((6, 256, 320)).astype(np.float32)
    nl = 3
    gm = rs.uniform(0.3, 1.4, size=(6, nl)) * (rs.uniform(size=(6, nl)) > 0.2)
    t = Transform2d(bn, qn, program=prog)
    pb = t.forward_channels(Xb, 'nhw', nlevels=nl)
    assert t.plan(6, 256, 320, nl).describe() != ''
    for i in (0, 3, 5):
        single = t.forward(Xb[i], nlevels=nl)
        assert np.array_equal(pb.lowpass[i], single.lowpass), (bn, qn, prog, i)
        for l in range(nl):
            assert np.array_equal(pb.highpasses[l][i], single.highpasses[l]), (bn, qn, prog, i, l)
        zs = np.array(t.inverse(single, gm))
        zb = np.array(t.inverse(Pyramid(np.array(pb.lowpass[i]), tuple(np.array(y[i]) for y in pb.highpasses)), gm))
        assert np.array_equal(zs, zb), (bn, qn, prog, i)
    want = o.Transform2d(biort(bn), qshift(qn)).forward(as_f64(Xb[3]), nlevels=nl)
    got = t.forward(Xb[3], nlevels=nl)
    assert_pyramids_close(got, want, XFM_TOL, same_dtype=False)
