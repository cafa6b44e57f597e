"""Golden vectors of the reference's dtcwt.registration (rjw57/dtcwt @ /root/reference).

BUILD-CONTAINER ONLY:   /opt/conda/bin/python3.9 tests/golden/make_golden_registration.py

Writes tests/golden/registration.npz: two synthetic images and the reference's outputs
(data only).  NumPy-compat shims as in oracle/check_registration_against_reference.py.
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

import dtcwt                                    # noqa: E402
import dtcwt.registration as R                  # noqa: E402


class _ListIndexable(np.ndarray):
    def __getitem__(self, key):
        return super().__getitem__(tuple(key) if isinstance(key, list) else key)


_ref_boxfilter = R._boxfilter
R._boxfilter = lambda X, k: np.asarray(_ref_boxfilter(np.asarray(X).view(_ListIndexable), k))


def scene(n, warp, seed):
    yy, xx = np.mgrid[0:n, 0:n] / float(n)
    x2, y2 = xx * (1 + warp[2]) + warp[0], yy * (1 + warp[3]) + warp[1]
    out = []
    for gx, gy in ((xx, yy), (x2, y2)):
        rs = np.random.RandomState(seed)
        im = np.zeros((n, n))
        for _ in range(24):
            fx, fy, ph = rs.uniform(1, 9), rs.uniform(1, 9), rs.uniform(0, 6.28)
            im += rs.uniform(0.2, 1) * np.cos(6.283 * (fx * gx + fy * gy) + ph)
        out.append(im)
    return out


im1, im2 = scene(128, (0.012, -0.008, 0.01, -0.006), 7)
t = dtcwt.Transform2d()
p1, p2 = t.forward(im1, nlevels=5), t.forward(im2, nlevels=5)
st = {'im1': im1, 'im2': im2}
for l, q in zip((2, 4), R.qtildematrices(p1, p2, [2, 4])):
    st['qtilde/%d' % l] = q
av = R.estimatereg(p1, p2)
st['estimatereg'] = av
st['estimatereg_custom'] = R.estimatereg(p1, p2, regshape=(5, 7), levels=[[4, 3], [3, 2]])
vx, vy = R.velocityfield(av, (32, 32), 'bilinear')
st['velocity_x'], st['velocity_y'] = vx, vy
st['warp'] = R.warp(im1, av, 'bilinear')
st['warphighpass'] = R.warphighpass(p1.highpasses[2], av, 'bilinear')
st['solve_in'] = np.random.RandomState(1).standard_normal((3, 4, 27))
st['solve_out'] = R.solvetransform(st['solve_in'])
np.savez_compressed(os.path.join(HERE, 'registration.npz'), **st)
print('registration.npz: %d arrays, %d bytes' % (len(st), os.path.getsize(os.path.join(HERE, 'registration.npz'))))
