"""A bounded randomised soak in the suite: fixed seed, about 200 transforms of random shapes (2-D up to
2000 on a side), wavelet sets, level counts, ext_modes and gain masks against the oracle.  (tools/soak.py is
the open-ended form of the same loop; the round-1 runs of it are in profiles/r01/soak_*.txt.)"""
import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Transform2d, Transform3d, Transform1d

pytestmark = pytest.mark.gpu

B2 = ['near_sym_a', 'near_sym_b', 'antonini', 'legall', 'near_sym_b_bp']
Q2 = ['qshift_a', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_06', 'qshift_b_bp', 'qshift_32']


def rel(a, b):
    return float(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)).max() / max(np.abs(b).max(), 1e-30))


def test_soak_2d_fixed_seed():
    rs = np.random.RandomState(20260928)
    done = 0
    sizes = [420] * 96 + [1100] * 12 + [2000] * 4           # mostly small, a few large
    for hi in sizes:
        shape = (int(rs.randint(2, hi)), int(rs.randint(2, hi)))
        bn, qn = B2[rs.randint(len(B2))], Q2[rs.randint(len(Q2))]
        nl = int(rs.randint(1, 6))
        X = rs.standard_normal(shape).astype(np.float32)
        to = o.Transform2d(biort(bn), qshift(qn))
        try:
            want = to.forward(X.astype(np.float64), nlevels=nl)
        except Exception:
            continue                                         # shapes the reference itself cannot transform
        t = Transform2d(bn, qn)
        p = t.forward(X, nlevels=nl)
        e = max([rel(p.lowpass, want.lowpass)] + [rel(a, b) for a, b in zip(p.highpasses, want.highpasses)])
        assert e < 2e-6, ('2d fwd', shape, bn, qn, nl, e)
        g = rs.uniform(0.2, 1.5, (6, nl))
        e = rel(t.inverse(p, g), to.inverse(want, g))
        assert e < 5e-6, ('2d inv', shape, bn, qn, nl, e)
        done += 1
    assert done >= 90


def test_soak_3d_1d_fixed_seed():
    rs = np.random.RandomState(20260929)
    done = 0
    for _ in range(60):
        ext = int(rs.choice([4, 8]))
        mult = 2 if ext == 4 else 4
        shape = tuple(int(mult * rs.randint(8 // mult, 72 // mult + 1)) for _ in range(3))
        bn, qn = B2[rs.randint(4)], Q2[rs.randint(5)]
        nl = int(rs.randint(1, 4))
        X = rs.standard_normal(shape).astype(np.float32)
        to = o.Transform3d(biort(bn), qshift(qn), ext_mode=ext)
        try:
            want = to.forward(X.astype(np.float64), nlevels=nl)
        except Exception:
            continue
        t = Transform3d(bn, qn, ext_mode=ext)
        p = t.forward(X, nlevels=nl)
        e = max([rel(p.lowpass, want.lowpass)] + [rel(a, b) for a, b in zip(p.highpasses, want.highpasses)])
        assert e < 2e-6, ('3d fwd', shape, bn, qn, nl, ext, e)
        e = rel(t.inverse(p), to.inverse(want))
        assert e < 1.5e-6, ('3d inv', shape, bn, qn, nl, ext, e)
        done += 1
    assert done >= 40
    for _ in range(40):
        n = 2 * int(rs.randint(8, 3000))
        k = int(rs.choice([1, 1, 3, 32, 40, 130]))
        nl = int(rs.randint(1, 6))
        bn, qn = B2[rs.randint(4)], Q2[rs.randint(5)]
        dt = rs.choice([np.float32, np.float64])
        x = rs.standard_normal((n, k)).astype(dt)
        to = o.Transform1d(biort(bn), qshift(qn))
        try:
            want = to.forward(x.astype(np.float64), nlevels=nl)
        except Exception:
            continue
        t = Transform1d(bn, qn)
        p = t.forward(x, nlevels=nl)
        tol = 2e-6 if dt == np.float32 else 1e-12
        e = max([rel(p.lowpass, want.lowpass)] + [rel(a, b) for a, b in zip(p.highpasses, want.highpasses)])
        assert e < tol, ('1d fwd', n, k, bn, qn, nl, dt, e)
        g = rs.uniform(0.5, 1.5, nl)
        e = rel(np.asarray(t.inverse(p, g)).reshape(n, k), np.asarray(to.inverse(want, g)).reshape(n, k))
        assert e < 10 * tol, ('1d inv', n, k, bn, qn, nl, dt, e)
