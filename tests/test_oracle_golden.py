"""The oracle against the golden vectors generated from the reference itself.

CPU-only.  This is what pins oracle/dtcwt_oracle.py on machines where /root/reference
does not exist (the GPU box); the exhaustive comparison is oracle/check_against_reference.py.
"""
import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
from tests import _golden as G

F64 = 1e-13
F32 = 2e-6


def _haar():
    h0 = np.array((1.0, 1.0)) / 2
    g0 = h0.copy()
    h1 = g0 * np.cumprod(-np.ones_like(g0))
    g1 = -h0 * np.cumprod(-np.ones_like(h0))
    return (h0, g0, h1, g1)


def test_lowlevel_golden():
    s = G.load('lowlevel')
    X = s['X']
    n = 0
    for k in s:
        parts = k.split('/')
        if parts[0] == 'colfilter' and parts[2] in ('float64', 'float32'):
            h = s['colfilter/%s/h' % parts[1]]
            got = o.colfilter(X.astype(parts[2]), h)
            assert got.dtype == s[k].dtype
            assert G.rel_err(got, s[k]) <= (F64 if parts[2] == 'float64' else 2e-7)
            n += 1
        elif parts[0] in ('coldfilt', 'colifilt'):
            ha, hb = s['pair/%s/ha' % parts[1]], s['pair/%s/hb' % parts[1]]
            fn = getattr(o, parts[0])
            got = fn(X.astype(parts[2]), ha, hb)
            assert got.dtype == s[k].dtype
            assert G.rel_err(got, s[k]) <= (F64 if parts[2] == 'float64' else 2e-7), k
            n += 1
    assert n > 60
    d = qshift('qshift_d')
    for r in (2, 4, 8):
        Xs = s['tiny/X%d' % r]
        assert G.rel_err(o.colfilter(Xs, biort('near_sym_b')[2]), s['tiny/colfilter%d' % r]) <= F64
        if r % 4 == 0:
            assert G.rel_err(o.coldfilt(Xs, d[1], d[0]), s['tiny/coldfilt%d' % r]) <= F64
        assert G.rel_err(o.colifilt(Xs, d[1], d[0]), s['tiny/colifilt%d' % r]) <= F64


def test_transform2d_golden():
    s = G.load('transform2d')
    for case in s['cases']:
        xn, bn, qn, nl, dt = G.parse_case(str(case))
        nl = int(nl[2:])
        tol = F64 if dt == 'float64' else F32
        t = o.Transform2d(biort(bn), qshift(qn))
        p = t.forward(s[xn].astype(dt), nlevels=nl, include_scale=True)
        G.check_pyramid(s, case + '/fwd', p, tol)
        G.check_stored(s, case + '/inv', t.inverse(p), tol * 10)
        if nl:
            G.check_stored(s, case + '/inv_gain', t.inverse(p, s[case + '/gain_mask']), tol * 10)


def test_mandrill_golden():
    s = G.load('mandrill')
    mand = s['mandrill']
    assert mand.shape == (512, 512) and mand.dtype == np.float32
    for nl in (3, 4):
        for bn, qn in (('near_sym_a', 'qshift_a'), ('near_sym_b_bp', 'qshift_b_bp')):
            for dt in ('float64', 'float32'):
                key = 'nl%d-%s-%s-%s' % (nl, bn, qn, dt)
                tol = 1e-12 if dt == 'float64' else 1e-5      # tests/test_againstmatlab.py:38 uses 1e-5
                p = o.Transform2d(biort(bn), qshift(qn)).forward(mand.astype(dt), nlevels=nl, include_scale=True)
                assert np.abs(G.summarise_mat(p.lowpass) - s[key + '/Yl']).max() <= tol
                for l in range(nl):
                    assert np.abs(G.summarise_mat(p.highpasses[l]) - s[key + '/Yh%d' % l]).max() <= tol
                    assert np.abs(G.summarise_mat(p.scales[l]) - s[key + '/Ys%d' % l]).max() <= tol
    # known-answer pins
    p = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(mand.astype(np.float64), nlevels=3)
    assert p.lowpass.shape == (128, 128)
    assert abs(p.lowpass.sum() - 33183.462677941905) < 1e-8
    assert abs(p.highpasses[2][5, 7, 3] - (-0.2492147741 - 0.0642030041j)) < 1e-9
    t = o.Transform2d(biort('near_sym_a'), qshift('qshift_a'))
    for name, crop, nl in (('r509', mand[:509, :], 3), ('c509', mand[:, :509], 3),
                           ('rc509', mand[:509, :509], 3), ('crop233x301', mand[:233, :301], 4)):
        p = t.forward(crop.astype(np.float64), nlevels=nl)
        G.check_pyramid(s, name + '/fwd', p, 1e-12)
        G.check_stored(s, name + '/inv', t.inverse(p), 1e-12)


def test_transform1d_golden():
    s = G.load('transform1d')
    for case in s['cases']:
        xn, bn, qn, nl, dt = G.parse_case(str(case))
        nl = int(nl[2:])
        tol = F64 if dt == 'float64' else F32
        t = o.Transform1d(biort(bn), qshift(qn))
        p = t.forward(s[xn].astype(dt), nlevels=nl, include_scale=True)
        G.check_pyramid(s, case + '/fwd', p, tol)
        G.check_stored(s, case + '/inv', t.inverse(p), tol * 10)
        if nl:
            G.check_stored(s, case + '/inv_gain', t.inverse(p, s[case + '/gain_mask']), tol * 10)


def _volume(s, xn):
    if xn == 'e32':
        g = slice(-16, 16)
        X, Y, Z = np.mgrid[g, g, g]
        r = np.sqrt(X * X + (Y * 1.2) ** 2 + (Z * 1.4) ** 2)
        return np.where(r <= 0.4 * 32, 1.0, 0.0)
    return s[xn]


def test_transform3d_golden():
    s = G.load('transform3d')
    for case in s['cases']:
        xn, bn, qn, nl, ext, d, dt = G.parse_case(str(case))
        nl, ext, d = int(nl[2:]), int(ext[3:]), bool(int(d[1:]))
        tol = F64 if dt == 'float64' else F32
        b = _haar() if bn == 'haar' else biort(bn)
        t = o.Transform3d(b, qshift(qn), ext_mode=ext)
        p = t.forward(_volume(s, xn).astype(dt), nlevels=nl, include_scale=True, discard_level_1=d)
        G.check_pyramid(s, case + '/fwd', p, tol)
        z = t.inverse(p)
        if d:
            # the fixture holds what the reference returns: axes 0 and 2 exchanged by _level1_ifm_no_highpass
            # (transform3d.py:454-456).  The literal mode reproduces it, the default is its transpose.
            tq = o.Transform3d(b, qshift(qn), ext_mode=ext, mimic_ifm_no_highpass_quirk=True)
            G.check_stored(s, case + '/inv', tq.inverse(p), tol * 10)
            z = z.transpose(2, 1, 0)
        G.check_stored(s, case + '/inv', z, tol * 10)


def test_perfect_reconstruction_oracle():
    rs = np.random.RandomState(0)
    X = rs.standard_normal((40, 56))
    t = o.Transform2d(biort('near_sym_b'), qshift('qshift_b'))
    assert np.abs(t.inverse(t.forward(X, 3)) - X).max() < 1e-12
    V = rs.standard_normal((16, 12, 20))
    t3 = o.Transform3d(biort('near_sym_a'), qshift('qshift_a'))
    assert np.abs(t3.inverse(t3.forward(V, 2)) - V).max() < 1e-12
