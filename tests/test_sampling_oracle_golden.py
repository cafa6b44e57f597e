"""The sampling oracle (oracle/sampling_oracle.py) against vectors generated from the
reference's dtcwt.sampling (tests/golden/sampling.npz, tests/golden/make_golden_sampling.py)."""
import os

import numpy as np
import pytest

from oracle import sampling_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = ('sample', 'sample3', 'rescale_up', 'rescale_down', 'upsample', 'sample_highpass', 'sample_highpass_sbs',
         'rescale_highpass', 'rescale_highpass_sbs', 'upsample_highpass')


def golden():
    return np.load(os.path.join(HERE, 'golden', 'sampling.npz'))


def compute(mod, g, case, kind, **kw):
    """One golden case through module *mod* (the oracle here, the hip backend in the GPU suite)."""
    m, dt = case.split('/')
    cdt = 'complex128' if dt == 'float64' else 'complex64'
    lo, lo3, hi, xs, ys = g['lo'].astype(dt), g['lo3'].astype(dt), g['hi'].astype(cdt), g['xs'], g['ys']
    if kind == 'sample':
        return mod.sample(lo, xs, ys, m, **kw)
    if kind == 'sample3':
        return mod.sample(lo3, xs, ys, m, **kw)
    if kind == 'rescale_up':
        return mod.rescale(lo, (40, 33), m, **kw)
    if kind == 'rescale_down':
        return mod.rescale(lo3, (5, 7), m, **kw)
    if kind == 'upsample':
        return mod.upsample(lo3, m, **kw)
    if kind == 'sample_highpass':
        return mod.sample_highpass(hi, xs, ys, m, **kw)
    if kind == 'sample_highpass_sbs':
        return mod.sample_highpass(hi, xs, ys, m, np.array([0, 2, 3, 5]), **kw)
    if kind == 'rescale_highpass':
        return mod.rescale_highpass(hi, (20, 30), m, **kw)
    if kind == 'rescale_highpass_sbs':
        return mod.rescale_highpass(hi, (9, 8), m, np.array([4, 1]), **kw)
    if kind == 'upsample_highpass':
        return mod.upsample_highpass(hi, m, **kw)
    raise KeyError(kind)


def close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    err = np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() / max(np.abs(b).max(), 1e-30)
    assert err <= tol, err


@pytest.mark.parametrize('kind', KINDS)
def test_oracle_matches_reference_vectors(kind):
    g = golden()
    for case in g['cases']:
        case = str(case)
        want = g[case + '/' + kind]
        got = compute(so, g, case, kind)
        assert got.dtype == want.dtype, (case, kind)
        close(got, want, 4e-7 if want.dtype in (np.float32, np.complex64) else 1e-13)


def test_oracle_argument_errors():
    with pytest.raises(ValueError):
        so.sample(np.zeros((4, 4)), np.zeros((2, 3)), np.zeros((3, 2)), 'bilinear')
    with pytest.raises(NotImplementedError):
        so.sample(np.zeros((4, 4)), np.zeros((2, 2)), np.zeros((2, 2)), 'cubic')
    with pytest.raises(ValueError):
        so.upsample(np.zeros((4, 4)), 'cubic')
