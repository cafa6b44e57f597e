"""Build dtcwt_amd/data/wavelets.npz: the published DT-CWT tap tables as DATA.

Run in the build container only (needs /root/reference/dtcwt/data/*.npz, the reference's
own data files; dtcwt/coeffs.py:13-25 loads the same files).  The output holds every
vector of every wavelet file, float64, key '<wavelet>/<vector>'.
"""
import glob
import os
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/dtcwt/data'
out = os.path.join(os.path.dirname(__file__), '..', 'dtcwt_amd', 'data', 'wavelets.npz')
table = {}
for f in sorted(glob.glob(os.path.join(src, '*.npz'))):
    name = os.path.splitext(os.path.basename(f))[0]
    d = np.load(f)
    for k in d.files:
        if k.startswith('__') or d[k].dtype.kind not in 'fiu':
            continue                      # MATLAB header strings
        table['%s/%s' % (name, k)] = np.asarray(d[k], dtype=np.float64).reshape(-1)
np.savez_compressed(out, **table)
print('wrote', os.path.abspath(out), len(table), 'vectors')
for k in sorted(table):
    print(k, table[k].shape[0])
