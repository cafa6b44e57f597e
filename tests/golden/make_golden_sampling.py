"""Golden vectors of the reference's dtcwt.sampling (rjw57/dtcwt @ /root/reference).

BUILD-CONTAINER ONLY:   /opt/conda/bin/python3.9 tests/golden/make_golden_sampling.py

Writes tests/golden/sampling.npz: seeded inputs and the reference's outputs (data only).
"""
import os
import sys
import warnings

import numpy as np

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

import dtcwt.sampling as S                       # noqa: E402  (the reference)

rs = np.random.RandomState(77)
st = {}
lo = rs.standard_normal((21, 26))
lo3 = rs.standard_normal((12, 10, 3))
hi = rs.standard_normal((14, 11, 6)) + 1j * rs.standard_normal((14, 11, 6))
xs = rs.uniform(-30, 60, (6, 9))
ys = rs.uniform(-25, 50, (6, 9))
xs[0, :4] = (-0.5, 25.5, 0.0, 3.5)
ys[0, :4] = (20.5, -0.5, 7.0, 2.5)
st.update(lo=lo, lo3=lo3, hi=hi, xs=xs, ys=ys)
cases = []
for m in ('nearest', 'bilinear', 'lanczos'):
    for dt in ('float64', 'float32'):
        key = '%s/%s' % (m, dt)
        cases.append(key)
        st[key + '/sample'] = S.sample(lo.astype(dt), xs, ys, m)
        st[key + '/sample3'] = S.sample(lo3.astype(dt), xs, ys, m)
        st[key + '/rescale_up'] = S.rescale(lo.astype(dt), (40, 33), m)
        st[key + '/rescale_down'] = S.rescale(lo3.astype(dt), (5, 7), m)
        st[key + '/upsample'] = S.upsample(lo3.astype(dt), m)
        cdt = 'complex128' if dt == 'float64' else 'complex64'
        st[key + '/sample_highpass'] = S.sample_highpass(hi.astype(cdt), xs, ys, m)
        st[key + '/sample_highpass_sbs'] = S.sample_highpass(hi.astype(cdt), xs, ys, m, np.array([0, 2, 3, 5]))
        st[key + '/rescale_highpass'] = S.rescale_highpass(hi.astype(cdt), (20, 30), m)
        st[key + '/rescale_highpass_sbs'] = S.rescale_highpass(hi.astype(cdt), (9, 8), m, np.array([4, 1]))
        st[key + '/upsample_highpass'] = S.upsample_highpass(hi.astype(cdt), m)
st['cases'] = np.array(cases)
np.savez_compressed(os.path.join(HERE, 'sampling.npz'), **st)
print('sampling.npz: %d arrays, %d bytes' % (len(st), os.path.getsize(os.path.join(HERE, 'sampling.npz'))))
