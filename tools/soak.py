"""Randomised soak of the fused paths against the oracle (development tool, GPU box):
python tools/soak.py [seconds]  -- 2-D, 3-D, re-sampling; prints the worst relative errors."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Transform2d, Transform3d
from dtcwt_amd import sampling
from oracle import dtcwt_oracle as o
from oracle import sampling_oracle as so

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(os.environ.get('SOAK_SEED', '1')))
B2 = ['near_sym_a', 'near_sym_b', 'antonini', 'legall', 'near_sym_b_bp']
Q2 = ['qshift_a', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_06', 'qshift_b_bp', 'qshift_32']


def rel(a, b):
    return float(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128)).max() / max(np.abs(b).max(), 1e-30))


worst = {'2d fwd': 0, '2d inv': 0, '3d fwd': 0, '3d inv': 0, 'sampling': 0}
n = {k: 0 for k in worst}
t0 = time.time()
while time.time() - t0 < budget:
    kind = rs.randint(3)
    if kind == 0:
        shape = (int(rs.randint(2, 420)), int(rs.randint(2, 420)))
        bn, qn = B2[rs.randint(len(B2))], Q2[rs.randint(len(Q2))]
        nl = int(rs.randint(1, 6))
        X = rs.standard_normal(shape).astype(np.float32)
        try:
            want = o.Transform2d(biort(bn), qshift(qn)).forward(X.astype(np.float64), nlevels=nl)
        except Exception:
            continue
        t = Transform2d(bn, qn)
        p = t.forward(X, nlevels=nl)
        e = max([rel(p.lowpass, want.lowpass)] + [rel(a, b) for a, b in zip(p.highpasses, want.highpasses)])
        worst['2d fwd'] = max(worst['2d fwd'], e); n['2d fwd'] += 1
        assert e < 2e-6, ('2d fwd', shape, bn, qn, nl, e)
        g = rs.uniform(0.2, 1.5, (6, nl))
        e = rel(t.inverse(p, g), o.Transform2d(biort(bn), qshift(qn)).inverse(want, g))
        worst['2d inv'] = max(worst['2d inv'], e); n['2d inv'] += 1
        assert e < 5e-6, ('2d inv', shape, bn, qn, nl, e)
    elif kind == 1:
        ext = int(rs.choice([4, 8]))
        mult = 2 if ext == 4 else 4
        shape = tuple(int(mult * rs.randint(8 // mult, 64 // mult + 1)) for _ in range(3))
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
        worst['3d fwd'] = max(worst['3d fwd'], e); n['3d fwd'] += 1
        assert e < 2e-6, ('3d fwd', shape, bn, qn, nl, ext, e)
        e = rel(t.inverse(p), to.inverse(want))
        worst['3d inv'] = max(worst['3d inv'], e); n['3d inv'] += 1
        assert e < 2e-5, ('3d inv', shape, bn, qn, nl, ext, e)
    else:
        shape = (int(rs.randint(3, 200)), int(rs.randint(3, 200)))
        out = (int(rs.randint(1, 300)), int(rs.randint(1, 300)))
        m = ['nearest', 'bilinear', 'lanczos'][rs.randint(3)]
        hi = (rs.standard_normal(shape + (6,)) + 1j * rs.standard_normal(shape + (6,))).astype(np.complex64)
        e = rel(sampling.rescale_highpass(hi, out, m), so.rescale_highpass(hi, out, m))
        lo = rs.standard_normal(shape)
        xs, ys = rs.uniform(-500, 500, (7, 9)), rs.uniform(-500, 500, (7, 9))
        e = max(e, rel(sampling.sample(lo, xs, ys, m), so.sample(lo, xs, ys, m)) * 1e6 * 1e-6)
        worst['sampling'] = max(worst['sampling'], e); n['sampling'] += 1
        assert e < 5e-6, ('sampling', shape, out, m, e)
print('soak %.0f s:' % (time.time() - t0), ', '.join('%s n=%d worst %.2e' % (k, n[k], worst[k]) for k in worst))
