"""Pin the oracle: compare oracle/dtcwt_oracle.py with the reference itself.

BUILD-CONTAINER ONLY.  Needs /root/reference (read-only mount); nothing here runs on
the GPU box.  Usage (either interpreter; conda py3.9 / NumPy 1.26 is the interpreter of
record because it keeps float32 pipelines in single precision):

    /opt/conda/bin/python3.9 oracle/check_against_reference.py
    python oracle/check_against_reference.py

The reference uses NumPy aliases that modern NumPy removed (np.int, np.asfarray,
np.issubsctype); they are re-created here *in this process only* before the import.
"""
import os
import sys
import logging
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True

import dtcwt                                   # noqa: E402  (the reference)
import dtcwt.numpy.lowlevel as rl              # noqa: E402
import dtcwt.numpy.transform2d as r2           # noqa: E402
import dtcwt.numpy.transform3d as r3           # noqa: E402
from dtcwt.numpy import Transform1d as R1d, Transform2d as R2d, Transform3d as R3d  # noqa
from dtcwt.coeffs import biort, qshift         # noqa: E402
import dtcwt_oracle as o                       # noqa: E402

NP1 = int(np.__version__.split('.')[0]) < 2
fails = []
nchecks = 0


def check(name, a, b, tol=0.0):
    global nchecks
    nchecks += 1
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        fails.append('%s: shape %s vs %s' % (name, a.shape, b.shape))
        return
    if a.size == 0:
        return
    d = np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max()
    scale = max(np.abs(b).max(), 1e-300)
    if d > tol * scale:
        fails.append('%s: max diff %.3e (rel %.3e) > %g' % (name, d, d / scale, tol))


rs = np.random.RandomState(1234)

# ---------------------------------------------------------------- low-level filters
odd_filters = [biort('near_sym_a')[0], biort('near_sym_a')[2], biort('near_sym_b')[2],
               biort('antonini')[0], np.array([1.0, 2.0, -1.0])]
even_filters = [np.array([-1.0, 1.0]), qshift('qshift_a')[0], np.array([1.0, 1.0, 3.0, 0.5])]
pairs = []
for q in ('qshift_a', 'qshift_06', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_32'):
    t = qshift(q)
    pairs += [(t[1], t[0]), (t[0], t[1]), (t[5], t[4]), (t[3], t[2]), (t[7], t[6])]
pairs += [(np.array([-1.0, 1.0]), np.array([1.0, -1.0])), (np.array([1.0, 1.0]), np.array([1.0, 1.0])),
          (np.array([-1.0, 0, 0, 1.0]), np.array([1.0, 0, 0, -1.0]))]

for dt, tol in ((np.float64, 1e-15), (np.float32, 2e-7)):
    for r in (1, 2, 3, 4, 8, 12, 64, 100):
        X = rs.standard_normal((r, 7)).astype(dt)
        for n, h in enumerate(odd_filters + even_filters):
            check('colfilter r=%d h#%d %s' % (r, n, dt.__name__), o.colfilter(X, h), rl.colfilter(X, h), tol)
        if r % 4 == 0:
            for n, (ha, hb) in enumerate(pairs):
                check('coldfilt r=%d pair#%d %s' % (r, n, dt.__name__), o.coldfilt(X, ha, hb), rl.coldfilt(X, ha, hb), tol)
        if r % 2 == 0:
            for n, (ha, hb) in enumerate(pairs):
                check('colifilt r=%d pair#%d %s' % (r, n, dt.__name__), o.colifilt(X, ha, hb), rl.colifilt(X, ha, hb), tol)

# bit-exactness in float32 on a bigger array (same op order as the reference)
X = rs.standard_normal((64, 48)).astype(np.float32)
exact = [np.array_equal(o.colfilter(X, odd_filters[1]), rl.colfilter(X, odd_filters[1])),
         np.array_equal(o.coldfilt(X, *pairs[0]), rl.coldfilt(X, *pairs[0])),
         np.array_equal(o.colifilt(X, *pairs[0]), rl.colifilt(X, *pairs[0]))]
print('float32 bit-exact (colfilter, coldfilt, colifilt):', exact)

# the row-0 quirk (lowlevel.py:202)
Xq = np.zeros((8, 3)); Xq[0, 1] = 1.0
check('colifilt quirk mimic', o.colifilt(Xq, *pairs[0], mimic_row0_quirk=True), rl.colifilt(Xq, *pairs[0]))

# error behaviour
for fn_o, fn_r, Xbad, args in ((o.coldfilt, rl.coldfilt, np.zeros((6, 2)), pairs[0]),
                               (o.colifilt, rl.colifilt, np.zeros((5, 2)), pairs[0]),
                               (o.coldfilt, rl.coldfilt, np.zeros((8, 2)), (np.ones(3), np.ones(3))),
                               (o.colifilt, rl.colifilt, np.zeros((8, 2)), (np.ones(4), np.ones(6)))):
    for fn in (fn_o, fn_r):
        try:
            fn(Xbad, *args)
            fails.append('%s did not raise' % fn)
        except ValueError:
            pass
    nchecks += 1

# ---------------------------------------------------------------- q2c / c2q / cube2c
Y = rs.standard_normal((12, 10))
check('q2c', o.q2c(Y), r2.q2c(Y), 1e-15)
W = rs.standard_normal((6, 5, 2)) + 1j * rs.standard_normal((6, 5, 2))
check('c2q', o.c2q(W, np.array([0.7, 1.3])), r2.c2q(W, np.array([0.7, 1.3])), 1e-15)
V = rs.standard_normal((6, 8, 4))
check('cube2c', o.cube2c(V), r3.cube2c(V), 1e-15)
check('c2cube', o.c2cube(o.cube2c(V)), V, 1e-15)
check('c2cube vs ref', o.c2cube(r3.cube2c(V)), r3.c2cube(r3.cube2c(V)), 1e-15)


# ---------------------------------------------------------------- 2-D transform
def cmp_pyr(tag, a, b, tol):
    check(tag + ' Yl', a.lowpass, b.lowpass, tol)
    assert len(a.highpasses) == len(b.highpasses)
    for l, (x, y) in enumerate(zip(a.highpasses, b.highpasses)):
        if x is None or y is None:
            if not (x is None and y is None):
                fails.append(tag + ' None mismatch level %d' % l)
            continue
        check(tag + ' Yh[%d]' % l, x, y, tol)
        if x.dtype != y.dtype and (NP1 or x.ndim == 3):
            fails.append('%s Yh[%d] dtype %s vs %s' % (tag, l, x.dtype, y.dtype))
    if (a.scales is None) != (b.scales is None):
        fails.append(tag + ' scales presence')
    elif a.scales is not None:
        for l, (x, y) in enumerate(zip(a.scales, b.scales)):
            check(tag + ' Yscale[%d]' % l, x, y, tol)


wave_pairs = [('near_sym_a', 'qshift_a'), ('antonini', 'qshift_06'), ('legall', 'qshift_c'),
              ('near_sym_b', 'qshift_d'), ('near_sym_b_bp', 'qshift_b_bp'), ('near_sym_b', 'qshift_b')]
shapes2 = [(64, 64), (36, 52), (33, 47), (8, 8), (16, 20), (100, 6), (2, 2), (1, 16)]
for bn, qn in wave_pairs:
    for shp in shapes2:
        for nlev in (0, 1, 2, 3, 4):
            for dt, tol in ((np.float64, 1e-14), (np.float32, 1e-6)):
                X = rs.standard_normal(shp).astype(dt)
                tr, to = R2d(bn, qn), o.Transform2d(biort(bn), qshift(qn))
                try:
                    pr = tr.forward(X, nlevels=nlev, include_scale=True)
                except Exception as e:           # reference fails => oracle must too
                    try:
                        to.forward(X, nlevels=nlev, include_scale=True)
                        fails.append('2d %s %s %s nl=%d: ref raised %r, oracle did not' % (bn, qn, shp, nlev, e))
                    except Exception:
                        pass
                    continue
                po = to.forward(X, nlevels=nlev, include_scale=True)
                tag = '2d %s/%s %s nl=%d %s' % (bn, qn, shp, nlev, dt.__name__)
                cmp_pyr(tag, po, pr, tol)
                if nlev:
                    gm = rs.uniform(0.2, 1.5, size=(6, nlev)) * (rs.uniform(size=(6, nlev)) > 0.2)
                    for g in (None, gm):
                        check(tag + ' inv', to.inverse(pr, g), tr.inverse(pr, g), tol * 10 if dt == np.float32 else tol)

# ---------------------------------------------------------------- 1-D transform
for bn, qn in wave_pairs[:4]:
    for shp in ((630,), (630, 20), (64, 3), (16,), (12, 2)):
        for nlev in (0, 1, 2, 3, 5):
            X = rs.standard_normal(shp)
            tr, to = R1d(bn, qn), o.Transform1d(biort(bn), qshift(qn))
            pr = tr.forward(X, nlevels=nlev, include_scale=True)
            po = to.forward(X, nlevels=nlev, include_scale=True)
            tag = '1d %s/%s %s nl=%d' % (bn, qn, shp, nlev)
            cmp_pyr(tag, po, pr, 1e-14)
            gm = rs.uniform(0.5, 1.5, size=max(nlev, 1))
            for g in (None, gm):
                check(tag + ' inv', to.inverse(pr, g), tr.inverse(pr, g), 1e-13)

# ---------------------------------------------------------------- 3-D transform
h0 = np.array((1.0, 1.0)) / 2
g0 = h0.copy()
h1 = g0 * np.cumprod(-np.ones_like(g0))
g1 = -h0 * np.cumprod(-np.ones_like(h0))
haar = (h0, g0, h1, g1)
cases3 = [((16, 24, 32), 4), ((32, 32, 32), 4), ((30, 26, 22), 4), ((36, 28, 20), 8), ((8, 8, 8), 4),
          ((16, 16, 16), 8)]
for (shp, ext) in cases3:
    for bn, qn in (('near_sym_a', 'qshift_a'), ('near_sym_b', 'qshift_b'), (haar, 'qshift_a')):
        for nlev in (1, 2, 3):
            for discard in (False, True):
                for dt, tol in ((np.float64, 1e-14), (np.float32, 1e-6)):
                    X = rs.standard_normal(shp).astype(dt)
                    b = bn if not isinstance(bn, str) else biort(bn)
                    tr = R3d(b, qshift(qn), ext_mode=ext)
                    to = o.Transform3d(b, qshift(qn), ext_mode=ext, mimic_ifm_no_highpass_quirk=True)
                    tag = '3d %s/%s %s ext%d nl=%d d=%d %s' % (bn if isinstance(bn, str) else 'haar', qn, shp, ext, nlev,
                                                              discard, dt.__name__)
                    if discard and not isinstance(bn, str):
                        continue    # reference raises: transform3d.py:299-313 cannot hold N+1 samples
                    try:
                        pr = tr.forward(X, nlevels=nlev, include_scale=True, discard_level_1=discard)
                    except Exception as e:
                        try:
                            to.forward(X, nlevels=nlev, include_scale=True, discard_level_1=discard)
                            fails.append(tag + ': ref raised %r, oracle did not' % (e,))
                        except Exception:
                            pass
                        continue
                    po = to.forward(X, nlevels=nlev, include_scale=True, discard_level_1=discard)
                    cmp_pyr(tag, po, pr, tol)
                    if discard and len(set(shp)) > 1:
                        continue    # reference raises here (transform3d.py:456 shape bug)
                    check(tag + ' inv', to.inverse(pr), tr.inverse(pr), tol * 10)

# ---------------------------------------------------------------- known-answer pins
mand = np.load(os.path.join(REF, 'tests', 'mandrill.npz'))['mandrill'].astype(np.float64)
p = o.Transform2d(biort('near_sym_a'), qshift('qshift_a')).forward(mand, nlevels=3, include_scale=True)
kat = [(p.lowpass.sum(), 33183.462677941905), (p.lowpass[0, 0], 1.4150963885999464),
       (p.lowpass[64, 64], 2.922374183512983), ((np.abs(p.highpasses[0]) ** 2).sum(), 1035.8733886852672),
       ((np.abs(p.highpasses[2]) ** 2).sum(), 622.8096883726982)]
for got, want in kat:
    nchecks += 1
    if abs(got - want) > 1e-9 * abs(want):
        fails.append('KAT %r vs %r' % (got, want))

print('numpy', np.__version__, '| checks:', nchecks, '| failures:', len(fails))
for f in fails[:40]:
    print('  FAIL', f)
sys.exit(1 if fails else 0)
