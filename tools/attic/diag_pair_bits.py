"""Diagnostic: where do k_inv21m and the same macro-steps as a marching pair (k_inv21p<7, 5, 10>) differ in the last bit?"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from dtcwt_amd.hip import Transform2d, Pyramid
os.environ['DTCWT_HIP_MARCH'] = '1'; os.environ['DTCWT_HIP_MARCH_INV'] = '1'
rs = np.random.RandomState(29)
X = rs.standard_normal((256, 320)).astype(np.float32)
t = Transform2d('near_sym_a', 'qshift_a')
p = t.forward(X, nlevels=2)
def both(pyr):
    out = {}
    for arm in ('0', '1'):
        os.environ['DTCWT_HIP_INV21_PAIR'] = arm
        out[arm] = np.array(t.inverse(pyr))
    return out['0'], out['1']
lo0 = np.zeros_like(np.array(p.lowpass)); h0 = np.zeros_like(np.array(p.highpasses[0]))
for dset in ((0, 5), (1, 4), (2, 3), (0, 1), (0, 2), (0, 4), (5, 4), (0,), (0, 1, 2, 3, 4, 5)):
    for part in ('re', 'im', 'both'):
        h1 = np.zeros_like(np.array(p.highpasses[1]))
        for d in dset:
            src = np.array(p.highpasses[1])[:, :, d]
            h1[:, :, d] = src.real if part == 're' else (1j * src.imag if part == 'im' else src)
        a, b = both(Pyramid(lo0, (h0, h1)))
        df = a != b
        r, c = np.where(df)
        print('subbands', dset, part, 'differing', int(df.sum()), 'row%4', np.bincount(r % 4, minlength=4) if len(r) else None, 'col%4', np.bincount(c % 4, minlength=4) if len(c) else None)
