"""Generate the golden fixtures from the REFERENCE ITSELF (rjw57/dtcwt @ /root/reference).

BUILD-CONTAINER ONLY.  Interpreter of record: /opt/conda/bin/python3.9 (NumPy 1.26.4),
which keeps float32 pipelines in single precision:

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

Writes tests/golden/*.npz: inputs and the reference's outputs (data only).  Large
outputs are stored as projections onto 8 seeded +-1 vectors plus the shape ("proj"),
which any indexing/arithmetics error changes; small ones are stored whole.
"""
import logging
import os
import sys
import warnings

import numpy as np

logging.disable(logging.WARNING)
warnings.filterwarnings('ignore')
if not hasattr(np, 'int'):
    np.int = int
if not hasattr(np, 'asfarray'):
    def _asfarray(a, dtype=None):
        a = np.asanyarray(a)
        if dtype is None or not np.issubdtype(np.dtype(dtype), np.inexact):
            dtype = np.float64
        return np.asarray(a, dtype=dtype)
    np.asfarray = _asfarray
if not hasattr(np, 'issubsctype'):
    np.issubsctype = lambda a, b: np.issubdtype(a if isinstance(a, type) else np.dtype(a).type, b)

REF = os.environ.get('DTCWT_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

from dtcwt.numpy import Transform1d, Transform2d, Transform3d       # noqa: E402
from dtcwt.numpy.lowlevel import colfilter, coldfilt, colifilt       # noqa: E402
from dtcwt.coeffs import biort, qshift                               # noqa: E402


def proj(a):
    """8 projections of a (possibly complex) array onto seeded +-1 vectors."""
    a = np.asarray(a)
    flat = a.reshape(-1).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    rs = np.random.RandomState(flat.size % 100003)
    out = []
    for _ in range(8):
        w = rs.randint(0, 2, size=flat.size) * 2.0 - 1.0
        out.append(np.dot(flat, w))
    return np.array(out)


def put(store, key, a, whole):
    a = np.asarray(a)
    store[key + '/shape'] = np.array(a.shape, dtype=np.int64)
    store[key + '/dtype'] = np.array(str(a.dtype))
    if whole:
        store[key + '/data'] = a
    else:
        store[key + '/proj'] = proj(a)
        store[key + '/absmax'] = np.array(np.abs(a).max() if a.size else 0.0)


def put_pyramid(store, key, p, whole):
    put(store, key + '/Yl', p.lowpass, whole)
    store[key + '/nlevels'] = np.array(len(p.highpasses))
    for l, y in enumerate(p.highpasses):
        if y is None:
            store[key + '/Yh%d/none' % l] = np.array(1)
        else:
            put(store, key + '/Yh%d' % l, y, whole)
    if p.scales is not None:
        for l, y in enumerate(p.scales):
            put(store, key + '/Ys%d' % l, y, whole)


# ------------------------------------------------------------------ low level
def gen_lowlevel():
    s = {}
    rs = np.random.RandomState(42)
    X = rs.standard_normal((32, 24))
    s['X'] = X
    filt = {'near_sym_a_h0o': biort('near_sym_a')[0], 'near_sym_a_h1o': biort('near_sym_a')[2],
            'near_sym_b_h1o': biort('near_sym_b')[2], 'm11': np.array([-1.0, 1.0]),
            'qshift_a_h0a': qshift('qshift_a')[0]}
    for name, h in filt.items():
        s['colfilter/%s/h' % name] = np.asarray(h).reshape(-1)
        for dt in (np.float64, np.float32):
            s['colfilter/%s/%s' % (name, np.dtype(dt).name)] = colfilter(X.astype(dt), h)
    pairs = {}
    for q in ('qshift_a', 'qshift_b', 'qshift_c', 'qshift_d'):
        t = qshift(q)
        pairs[q + '_h0'] = (t[1], t[0])      # forward lowpass call order (h0b, h0a)
        pairs[q + '_h1'] = (t[5], t[4])      # forward highpass (h1b, h1a)
        pairs[q + '_g0'] = (t[3], t[2])
        pairs[q + '_g1'] = (t[7], t[6])
    pairs['m11'] = (np.array([-1.0, 1.0]), np.array([1.0, -1.0]))
    pairs['p11'] = (np.array([1.0, 1.0]), np.array([1.0, 1.0]))
    pairs['m1001'] = (np.array([-1.0, 0, 0, 1.0]), np.array([1.0, 0, 0, -1.0]))
    d = qshift('qshift_d')
    pairs['qshift_d_h1a_inner'] = (d[4][1:-1], d[5][1:-1])   # tests/test_openclcolifilt.py:97-109
    for name, (ha, hb) in pairs.items():
        s['pair/%s/ha' % name] = np.asarray(ha).reshape(-1)
        s['pair/%s/hb' % name] = np.asarray(hb).reshape(-1)
        for dt in (np.float64, np.float32):
            dn = np.dtype(dt).name
            s['coldfilt/%s/%s' % (name, dn)] = coldfilt(X.astype(dt), ha, hb)
            s['colifilt/%s/%s' % (name, dn)] = colifilt(X.astype(dt), ha, hb)
    # tiny row counts: multi-bounce reflection
    for r in (2, 4, 8):
        Xs = rs.standard_normal((r, 5))
        s['tiny/X%d' % r] = Xs
        s['tiny/colfilter%d' % r] = colfilter(Xs, biort('near_sym_b')[2])
        ha, hb = pairs['qshift_d_h0']
        if r % 4 == 0:
            s['tiny/coldfilt%d' % r] = coldfilt(Xs, ha, hb)
        s['tiny/colifilt%d' % r] = colifilt(Xs, ha, hb)
    np.savez_compressed(os.path.join(HERE, 'lowlevel.npz'), **s)


# ------------------------------------------------------------------ 2-D
WAVES = [('near_sym_a', 'qshift_a'), ('antonini', 'qshift_06'), ('near_sym_b', 'qshift_d'),
         ('near_sym_b_bp', 'qshift_b_bp'), ('legall', 'qshift_c')]


def gen_2d():
    s = {}
    rs = np.random.RandomState(7)
    cases = []
    X3652 = rs.standard_normal((36, 52))
    X3347 = rs.standard_normal((33, 47))
    X6464 = rs.standard_normal((64, 64))
    s['X3652'] = X3652; s['X3347'] = X3347; s['X6464'] = X6464
    for bn, qn in WAVES:
        cases.append(('X3652', bn, qn, 3, 'float64', True))
    cases += [('X3347', 'near_sym_a', 'qshift_a', 2, 'float32', True),
              ('X3347', 'near_sym_a', 'qshift_a', 0, 'float64', True),
              ('X6464', 'near_sym_a', 'qshift_a', 4, 'float32', True),
              ('X6464', 'near_sym_a', 'qshift_a', 1, 'float64', True),
              ('X6464', 'near_sym_b', 'qshift_d', 4, 'float64', False)]
    names = []
    for (xn, bn, qn, nl, dt, whole) in cases:
        key = '%s-%s-%s-nl%d-%s' % (xn, bn, qn, nl, dt)
        names.append(key)
        X = s[xn].astype(dt)
        t = Transform2d(bn, qn)
        p = t.forward(X, nlevels=nl, include_scale=True)
        put_pyramid(s, key + '/fwd', p, whole)
        put(s, key + '/inv', t.inverse(p), whole)
        if nl:
            gm = np.random.RandomState(nl).uniform(0.3, 1.4, size=(6, nl))
            gm[np.random.RandomState(nl + 1).uniform(size=(6, nl)) < 0.25] = 0.0
            s[key + '/gain_mask'] = gm
            put(s, key + '/inv_gain', t.inverse(p, gm), whole)
    s['cases'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'transform2d.npz'), **s)


def summarise_mat(M, apron=8):
    """The reference tests' own reduction (tests/util.py:46-60), restated."""
    def mean(a, axis):
        return np.expand_dims(np.mean(a, axis=axis), axis)
    centre = mean(mean(M[apron:-apron, apron:-apron, ...], 0), 1)
    return np.vstack((
        np.hstack((M[:apron, :apron, ...], mean(M[:apron, apron:-apron, ...], 1), M[:apron, -apron:, ...])),
        np.hstack((mean(M[apron:-apron, :apron, ...], 0), centre, mean(M[apron:-apron, -apron:, ...], 0))),
        np.hstack((M[-apron:, :apron, ...], mean(M[-apron:, apron:-apron, ...], 1), M[-apron:, -apron:, ...])),
    ))


def gen_mandrill():
    s = {}
    mand = np.load(os.path.join(REF, 'tests', 'mandrill.npz'))['mandrill']
    s['mandrill'] = mand                       # the reference tests' own data file (512x512 float32)
    for nl in (3, 4):
        for bn, qn in (('near_sym_a', 'qshift_a'), ('near_sym_b_bp', 'qshift_b_bp')):
            for dt in ('float64', 'float32'):
                key = 'nl%d-%s-%s-%s' % (nl, bn, qn, dt)
                p = Transform2d(bn, qn).forward(mand.astype(dt), nlevels=nl, include_scale=True)
                s[key + '/Yl'] = summarise_mat(p.lowpass)
                for l in range(nl):
                    s[key + '/Yh%d' % l] = summarise_mat(p.highpasses[l])
                    s[key + '/Ys%d' % l] = summarise_mat(p.scales[l])
                    s[key + '/Yh%d_energy' % l] = np.array((np.abs(p.highpasses[l].astype(np.complex128)) ** 2).sum())
                s[key + '/Yl_sum'] = np.array(p.lowpass.astype(np.float64).sum())
    # odd sizes the reference tests use (tests/test_xfm2.py:41-57, test_ifm2.py:13)
    for name, crop in (('r509', mand[:509, :]), ('c509', mand[:, :509]), ('rc509', mand[:509, :509]),
                       ('crop233x301', mand[:233, :301])):
        p = Transform2d().forward(crop.astype(np.float64), nlevels=4 if 'crop' in name else 3)
        put_pyramid(s, name + '/fwd', p, False)
        put(s, name + '/inv', Transform2d().inverse(p), False)
    np.savez_compressed(os.path.join(HERE, 'mandrill.npz'), **s)


# ------------------------------------------------------------------ 1-D
def gen_1d():
    s = {}
    rs = np.random.RandomState(11)
    s['v630'] = rs.standard_normal(630)
    s['m630x20'] = rs.standard_normal((630, 20))
    s['m64x3'] = rs.standard_normal((64, 3))
    names = []
    for xn, nl, dt in (('v630', 5, 'float64'), ('m630x20', 4, 'float64'), ('m64x3', 3, 'float32'),
                       ('m64x3', 0, 'float64')):
        for bn, qn in (('near_sym_a', 'qshift_a'), ('near_sym_b', 'qshift_d')):
            key = '%s-%s-%s-nl%d-%s' % (xn, bn, qn, nl, dt)
            names.append(key)
            t = Transform1d(bn, qn)
            p = t.forward(s[xn].astype(dt), nlevels=nl, include_scale=True)
            whole = s[xn].size < 2000
            put_pyramid(s, key + '/fwd', p, whole)
            put(s, key + '/inv', t.inverse(p), whole)
            if nl:
                gm = np.random.RandomState(nl).uniform(0.3, 1.4, size=nl)
                s[key + '/gain_mask'] = gm
                put(s, key + '/inv_gain', t.inverse(p, gm), whole)
    s['cases'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'transform1d.npz'), **s)


# ------------------------------------------------------------------ 3-D
def gen_3d():
    s = {}
    rs = np.random.RandomState(23)
    G = 32
    grid = slice(-(G >> 1), (G >> 1))
    Xg, Yg, Zg = np.mgrid[grid, grid, grid]
    r = np.sqrt(Xg * Xg + (Yg * 1.2) ** 2 + (Zg * 1.4) ** 2)
    ellipsoid = np.where(r <= 0.4 * G, 1.0, 0.0)           # tests/test_xfm3.py:11-21
    vols = {'e32': ellipsoid, 'r8': rs.standard_normal((8, 8, 8)),
            'r162432': rs.standard_normal((16, 24, 32)), 'r302622': rs.standard_normal((30, 26, 22)),
            'r362820': rs.standard_normal((36, 28, 20))}
    for k, v in vols.items():
        if k != 'e32':
            s[k] = v
    h0 = np.array((1.0, 1.0)) / 2
    g0 = h0.copy()
    h1 = g0 * np.cumprod(-np.ones_like(g0))
    g1 = -h0 * np.cumprod(-np.ones_like(h0))
    haar = (h0, g0, h1, g1)
    cases = [('r8', 'near_sym_a', 'qshift_a', 2, 4, False, 'float64', True),
             ('r8', 'near_sym_a', 'qshift_a', 2, 4, False, 'float32', True),
             ('e32', 'near_sym_a', 'qshift_a', 3, 4, False, 'float64', False),
             ('e32', 'near_sym_b', 'qshift_b', 3, 4, False, 'float32', False),
             ('e32', 'near_sym_a', 'qshift_a', 4, 4, True, 'float64', False),
             ('e32', 'haar', 'qshift_a', 1, 4, False, 'float64', False),
             ('r162432', 'near_sym_a', 'qshift_a', 2, 4, False, 'float64', False),
             ('r302622', 'near_sym_a', 'qshift_a', 3, 4, False, 'float64', False),
             ('r362820', 'near_sym_a', 'qshift_a', 3, 8, False, 'float64', False)]
    names = []
    for (xn, bn, qn, nl, ext, discard, dt, whole) in cases:
        key = '%s-%s-%s-nl%d-ext%d-d%d-%s' % (xn, bn, qn, nl, ext, discard, dt)
        names.append(key)
        b = haar if bn == 'haar' else biort(bn)
        t = Transform3d(b, qshift(qn), ext_mode=ext)
        p = t.forward(vols[xn].astype(dt), nlevels=nl, include_scale=True, discard_level_1=discard)
        put_pyramid(s, key + '/fwd', p, whole)
        # stored exactly as the reference returns it -- for the discard_level_1 case that includes the
        # axis-0/2 exchange of _level1_ifm_no_highpass (transform3d.py:454-456): the tests replay it with
        # reference_quirks=True and check the default (intended) result against its transpose
        put(s, key + '/inv', t.inverse(p), whole)
    s['cases'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'transform3d.npz'), **s)


if __name__ == '__main__':
    only = sys.argv[1:]
    for name, fn in (('lowlevel', gen_lowlevel), ('2d', gen_2d), ('mandrill', gen_mandrill), ('1d', gen_1d),
                     ('3d', gen_3d)):
        if not only or name in only:
            fn()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
    print('numpy', np.__version__)
