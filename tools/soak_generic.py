"""Randomised soak of the any-wavelet / float64 paths (generic2d.hip: two-launch 2-D levels,
one-launch 1-D levels, dense pair / sum fast paths under the filter-by-filter 3-D levels) against
the oracle (development tool, GPU box): python tools/soak_generic.py [seconds]."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Transform1d, Transform2d, Transform3d
from oracle import dtcwt_oracle as o

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(os.environ.get('SOAK_SEED', '1')))
B = ['near_sym_a', 'near_sym_b', 'antonini', 'legall']
Q = ['qshift_a', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_06', 'qshift_32']


def rel(a, b):
    return float(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)).max() / max(np.abs(b).max(), 1e-30))


def pyr_err(p, want):
    return max([rel(p.lowpass, want.lowpass)] + [rel(a, b) for a, b in zip(p.highpasses, want.highpasses)])


worst, n = {}, {}


def note(key, e, limit, ctx):
    worst[key] = max(worst.get(key, 0), e)
    n[key] = n.get(key, 0) + 1
    assert e < limit, (key,) + ctx + (e,)


t0 = time.time()
while time.time() - t0 < budget:
    kind = rs.randint(4)
    bn, qn = B[rs.randint(len(B))], Q[rs.randint(len(Q))]
    dt = [np.float32, np.float64][rs.randint(2)]
    ftol, itol = (2e-6, 8e-6) if dt == np.float32 else (1e-12, 1e-11)
    tag = 'f32' if dt == np.float32 else 'f64'
    if kind == 0:                   # 2-D, float64 or float32 on sizes / wavelets without a fused plan
        shape = (int(rs.randint(2, 300)), int(rs.randint(2, 300)))
        nl = int(rs.randint(1, 6))
        X = rs.standard_normal(shape).astype(dt)
        to = o.Transform2d(biort(bn), qshift(qn))
        try:
            want = to.forward(X.astype(np.float64), nlevels=nl)
        except Exception:
            continue
        t = Transform2d(bn, qn)
        p = t.forward(X, nlevels=nl)
        note('2d fwd ' + tag, pyr_err(p, want), ftol, (shape, bn, qn, nl))
        g = rs.uniform(0.2, 1.5, (6, nl))
        note('2d inv ' + tag, rel(t.inverse(p, g), to.inverse(want, g)), itol, (shape, bn, qn, nl))
    elif kind == 1:                 # 1-D: a vector, or k columns (k >= 32 marches, smaller k falls back)
        nn = 2 * int(rs.randint(4, 3000))
        k = int(rs.choice([1, 1, 2, 33, 40, 64, 100]))
        nl = int(rs.randint(1, 7))
        X = rs.standard_normal((nn, k) if k > 1 or rs.randint(2) else (nn,)).astype(dt)
        to = o.Transform1d(biort(bn), qshift(qn))
        try:
            want = to.forward(X.astype(np.float64), nlevels=nl)
        except Exception:
            continue
        t = Transform1d(bn, qn)
        p = t.forward(X, nlevels=nl)
        note('1d fwd ' + tag, pyr_err(p, want), ftol, (X.shape, bn, qn, nl))
        g = rs.uniform(0.2, 1.5, nl)
        note('1d inv ' + tag, rel(t.inverse(p, g), to.inverse(want, g)), itol * 2, (X.shape, bn, qn, nl))
    elif kind == 2:                 # 3-D through the filter-by-filter levels (float64 has no fused kernels)
        ext = int(rs.choice([4, 8]))
        mult = 2 if ext == 4 else 4
        shape = tuple(int(mult * rs.randint(8 // mult, 72 // mult + 1)) for _ in range(3))
        nl = int(rs.randint(1, 4))
        X = rs.standard_normal(shape)
        to = o.Transform3d(biort(bn), qshift(qn), ext_mode=ext)
        try:
            want = to.forward(X, nlevels=nl)
        except Exception:
            continue
        t = Transform3d(bn, qn, ext_mode=ext)
        p = t.forward(X, nlevels=nl)
        note('3d fwd f64', pyr_err(p, want), 1e-12, (shape, bn, qn, nl, ext))
        note('3d inv f64', rel(t.inverse(p), to.inverse(want)), 1e-11, (shape, bn, qn, nl, ext))
    else:                           # larger 2-D float64 (several segments per line, several row groups)
        shape = (int(rs.randint(300, 1400)), int(rs.randint(300, 2600)))
        nl = int(rs.randint(2, 5))
        X = rs.standard_normal(shape)
        to = o.Transform2d(biort(bn), qshift(qn))
        want = to.forward(X, nlevels=nl)
        t = Transform2d(bn, qn)
        p = t.forward(X, nlevels=nl)
        note('2d big fwd f64', pyr_err(p, want), 1e-12, (shape, bn, qn, nl))
        note('2d big inv f64', rel(t.inverse(p), to.inverse(want)), 1e-11, (shape, bn, qn, nl))
print('soak %.0f s:' % (time.time() - t0))
for k in sorted(worst):
    print('  %-16s n=%-5d worst %.2e' % (k, n[k], worst[k]))
