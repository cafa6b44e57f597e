"""Seeded random 3-D inverses with long rows (axis 2 up to 700 samples: k tiles with halo cells in k_inv3_l1_axis02):
fused against the generic axis passes.  `python tools/soak_inv3_wide.py [trials] [seed]`"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(__file__), '..')))
from dtcwt_amd.hip import Transform3d

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
biorts = ['near_sym_a', 'antonini', 'legall']
worst = 0.0
for t in range(trials):
    shape = (2 * rs.randint(6, 20), 2 * rs.randint(8, 40), 2 * rs.randint(4, 350))
    bn = biorts[rs.randint(3)]
    nl = int(rs.randint(1, 3))
    X = rs.standard_normal(shape).astype(np.float32)
    f, g = Transform3d(bn), Transform3d(bn)
    g.fused = False
    p = f.forward(X, nlevels=nl)
    zf, zg = np.asarray(f.inverse(p)), np.asarray(g.inverse(p))
    e = max(float(np.abs(zf - zg).max()), float(np.abs(zf - X).max())) / float(np.abs(X).max())
    worst = max(worst, e)
    assert e < 3e-5, (shape, bn, nl, e)
print('soak_inv3_wide: %d trials, worst relative error %.2e' % (trials, worst))
