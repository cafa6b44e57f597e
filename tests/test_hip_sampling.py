"""dtcwt_amd.sampling (hip kernels) against the reference's golden vectors, the oracle, and
the reference's own test patterns (tests/test_sampling.py of the reference)."""
import numpy as np
import pytest

import dtcwt_amd
from dtcwt_amd import sampling
from dtcwt_amd.hip import Transform2d
from dtcwt_amd.hip._lib import DeviceArray
from oracle import sampling_oracle as so
from tests.test_sampling_oracle_golden import KINDS, golden, compute, close

pytestmark = pytest.mark.gpu

F32_TOL = 2e-6      # float32 interpolation against the reference's float64-promoted result
F64_TOL = 1e-12


@pytest.mark.parametrize('kind', KINDS)
def test_golden_vectors(kind):
    g = golden()
    for case in g['cases']:
        case = str(case)
        want = g[case + '/' + kind]
        got = compute(sampling, g, case, kind)
        assert got.dtype == want.dtype, (case, kind, got.dtype, want.dtype)
        close(got, want, F32_TOL if case.endswith('float32') else F64_TOL)


@pytest.mark.parametrize('method', ['nearest', 'bilinear', 'lanczos'])
def test_vs_oracle_far_outside_and_large(method):
    rs = np.random.RandomState(3)
    im = rs.standard_normal((300, 517, 2))
    xs = rs.uniform(-3000, 3000, (64, 33))
    ys = rs.uniform(-2000, 2000, (64, 33))
    close(sampling.sample(im, xs, ys, method), so.sample(im, xs, ys, method), F64_TOL)
    close(sampling.sample(im.astype(np.float32), xs, ys, method), so.sample(im.astype(np.float32), xs, ys, method),
          F32_TOL)
    hi = (rs.standard_normal((96, 128, 6)) + 1j * rs.standard_normal((96, 128, 6)))
    close(sampling.rescale_highpass(hi, (150, 100), method), so.rescale_highpass(hi, (150, 100), method), 1e-11)
    close(sampling.upsample_highpass(hi.astype(np.complex64), method),
          so.upsample_highpass(hi.astype(np.complex64), method), F32_TOL)


def test_reference_rescale_round_trips():
    """tests/test_sampling.py of the reference: up and down again, per method."""
    X = np.random.RandomState(0).rand(100, 120)
    for method, up, tol in (('lanczos', (300, 210), 5e-2), ('bilinear', (300, 210), 3e-1), ('nearest', (200, 240), 1e-2)):
        Xrs = sampling.rescale(X, up, method)
        assert Xrs.shape == up
        Xrecon = sampling.rescale(Xrs, X.shape, method)
        assert Xrecon.shape == X.shape
        assert np.all(np.abs(X - Xrecon) < tol)


def test_integer_image_and_errors():
    im = np.random.RandomState(1).randint(0, 255, (10, 12))
    for m in ('nearest', 'bilinear', 'lanczos'):
        a, b = sampling.rescale(im, (20, 30), m), so.rescale(im, (20, 30), m)
        assert a.dtype == b.dtype
        if m == 'bilinear':         # truncation back to integers: at most one count apart
            assert np.abs(a - b).max() <= 1
        else:
            close(a, b, F64_TOL)
        assert sampling.upsample(im, m).dtype == np.float64
    with pytest.raises(ValueError):
        sampling.sample(np.zeros((4, 4)), np.zeros((2, 3)), np.zeros((3, 2)))
    with pytest.raises(NotImplementedError):
        sampling.sample(np.zeros((4, 4)), np.zeros((2, 2)), np.zeros((2, 2)), 'cubic')
    with pytest.raises(ValueError):
        sampling.upsample(np.zeros((4, 4)), 'cubic')


def test_device_resident_pyramid_is_resampled_in_place():
    """The use the reference's registration makes of it: highpass subbands of a pyramid,
    re-sampled without a round trip through the host."""
    X = np.random.RandomState(5).standard_normal((128, 160)).astype(np.float32)
    p = Transform2d().forward(X, nlevels=3)
    d = p.hip_highpasses[1]
    assert isinstance(d, DeviceArray)
    out = sampling.rescale_highpass(d, (64, 80), 'bilinear', device_output=True)
    assert isinstance(out, DeviceArray) and out.shape == (64, 80, 6) and out.dtype == np.complex64
    want = so.rescale_highpass(p.highpasses[1], (64, 80), 'bilinear')
    close(out.get(), want, F32_TOL)
    lo = sampling.upsample(p.hip_lowpass, 'lanczos', device_output=True)
    close(lo.get(), so.upsample(p.lowpass, 'lanczos'), F32_TOL)
    assert dtcwt_amd.sampling.sample is sampling.sample


def test_nearest_at_exact_ties_follows_numpy_rounding():
    """Output pixel centres that fall exactly half-way between two source pixels (4 rows -> 210: row 52 sits at
    y = 0.5) must round like NumPy does on the coordinate NumPy computes: scale * (i + 1/2) - 1/2 in two rounded
    steps, half to even -- a fused multiply-add differs in the last bit and picks the other row (found by
    tools/soak.py, seed 11)."""
    rs = np.random.RandomState(3)
    for shape, out in (((4, 106), (210, 31)), ((6, 10), (300, 25)), ((3, 3), (6, 6)), ((5, 7), (50, 70))):
        im = rs.standard_normal(shape)
        assert np.array_equal(sampling.rescale(im, out, 'nearest'), so.rescale(im, out, 'nearest')), (shape, out)
        hi = (rs.standard_normal(shape + (6,)) + 1j * rs.standard_normal(shape + (6,))).astype(np.complex64)
        close(sampling.rescale_highpass(hi, out, 'nearest'), so.rescale_highpass(hi, out, 'nearest'), 5e-6)
