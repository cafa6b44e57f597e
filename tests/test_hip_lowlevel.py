"""colfilter / coldfilt / colifilt of the hip backend.

Shape and ValueError behaviour mirrors the reference's tests/test_colfilter.py:19-50,
test_coldfilt.py:20-40, test_colifilt.py:20-53; numerics mirror the accelerated-vs-gold
pattern of tests/test_openclcolfilter.py:22-84 etc. with the oracle (and the golden vectors
from the reference itself) as gold."""
import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip.lowlevel import colfilter, coldfilt, colifilt
from tests import _golden as G
from tests._hip import assert_close, LOW_TOL, F64_TOL


# ---- argument checking happens on the host, before any device call: runs anywhere ----
def test_coldfilt_argument_errors():
    X = np.zeros((512, 512), np.float32)
    with pytest.raises(ValueError):
        coldfilt(X, (-1, 1), (1, -1, 0))          # different size
    with pytest.raises(ValueError):
        coldfilt(X, (-1, 2, 1), (1, 2, -1))       # odd filter
    with pytest.raises(ValueError):
        coldfilt(X[:511, :], (-1, 1), (1, -1))    # bad input size


def test_colifilt_argument_errors():
    X = np.zeros((512, 512), np.float32)
    with pytest.raises(ValueError):
        colifilt(X, (-1, 1), (1, -1, 0))
    with pytest.raises(ValueError):
        colifilt(X, (-1, 2, 1), (1, 2, -1))
    with pytest.raises(ValueError):
        colifilt(X[:511, :], (-1, 1), (1, -1))


pytestmark_gpu = pytest.mark.gpu


def _mandrill():
    return G.load('mandrill')['mandrill']


@pytest.mark.gpu
def test_colfilter_shapes_and_zero():
    m = _mandrill()
    assert colfilter(m, (-1, 2, -1)).shape == m.shape            # odd
    assert colfilter(m, (-1, 1)).shape == (m.shape[0] + 1, m.shape[1])   # even
    assert colfilter(m[:, :481], (-1, 2, -1)).shape == (512, 481)
    z = colfilter(np.zeros_like(m), biort('antonini')[0])
    assert z.shape == m.shape and not np.any(z)
    assert colfilter(m, biort('antonini')[0]).dtype == np.float32
    assert coldfilt(m, (-1, 1), (1, -1)).shape == (256, 512)
    assert colifilt(m, (-1, 1), (1, -1)).shape == (1024, 512)
    assert not np.any(colifilt(np.zeros_like(m), (-1, 1), (1, -1)))
    assert colfilter(m.tolist()[:8], (1, 2, 1)).dtype == np.float64     # list input -> float64


@pytest.mark.gpu
@pytest.mark.parametrize('dt', ['float32', 'float64'])
def test_lowlevel_vs_oracle_mandrill(dt):
    m = _mandrill().astype(dt)
    tol = LOW_TOL if dt == 'float32' else F64_TOL
    filt = [biort('near_sym_a')[0], biort('near_sym_a')[2], biort('near_sym_b')[2], (-1, 1),
            qshift('qshift_a')[0], (1, 2, 3, 4, 5, 6)]
    for h in filt:
        got = colfilter(m, h)
        assert got.dtype == m.dtype
        assert_close(got, o.colfilter(m, h), tol, 'colfilter m=%d' % len(np.ravel(h)))
    d = qshift('qshift_d')
    pairs = []
    for q in ('qshift_a', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_32'):
        t = qshift(q)
        pairs += [(t[1], t[0]), (t[0], t[1]), (t[5], t[4]), (t[7], t[6])]
    pairs += [((-1, 1), (1, -1)), ((1, 1), (1, 1)), ((-1, 0, 0, 1), (1, 0, 0, -1)),
              (d[4][1:-1], d[5][1:-1])]                           # tests/test_openclcolifilt.py:97-109
    for ha, hb in pairs:
        assert_close(coldfilt(m, ha, hb), o.coldfilt(m, ha, hb), tol, 'coldfilt')
        assert_close(colifilt(m, ha, hb), o.colifilt(m, ha, hb), tol, 'colifilt')


@pytest.mark.gpu
def test_lowlevel_golden():
    s = G.load('lowlevel')
    X = s['X']
    for k in s:
        parts = k.split('/')
        if parts[0] == 'colfilter' and parts[2] in ('float64', 'float32'):
            got = colfilter(X.astype(parts[2]), s['colfilter/%s/h' % parts[1]])
        elif parts[0] == 'coldfilt':
            got = coldfilt(X.astype(parts[2]), s['pair/%s/ha' % parts[1]], s['pair/%s/hb' % parts[1]])
        elif parts[0] == 'colifilt':
            got = colifilt(X.astype(parts[2]), s['pair/%s/ha' % parts[1]], s['pair/%s/hb' % parts[1]])
        else:
            continue
        assert got.dtype == s[k].dtype
        assert_close(got, s[k], LOW_TOL if parts[2] == 'float32' else F64_TOL, k)
    d = qshift('qshift_d')
    for r in (2, 4, 8):                                            # multi-bounce reflection
        Xs = s['tiny/X%d' % r]
        assert_close(colfilter(Xs, biort('near_sym_b')[2]), s['tiny/colfilter%d' % r], F64_TOL)
        if r % 4 == 0:
            assert_close(coldfilt(Xs, d[1], d[0]), s['tiny/coldfilt%d' % r], F64_TOL)
        assert_close(colifilt(Xs, d[1], d[0]), s['tiny/colifilt%d' % r], F64_TOL)


@pytest.mark.gpu
def test_lowlevel_ragged_shapes():
    rs = np.random.RandomState(9)
    q = qshift('qshift_b')
    for shape in ((4, 1), (8, 3), (12, 129), (100, 7), (4, 1000)):
        X = rs.standard_normal(shape).astype(np.float32)
        assert_close(colfilter(X, biort('near_sym_b')[0]), o.colfilter(X, biort('near_sym_b')[0]), LOW_TOL)
        assert_close(coldfilt(X, q[1], q[0]), o.coldfilt(X, q[1], q[0]), LOW_TOL)
        assert_close(colifilt(X, q[3], q[2]), o.colifilt(X, q[3], q[2]), LOW_TOL)


@pytest.mark.gpu
def test_device_array_in_device_array_out():
    from dtcwt_amd.hip import default_context, DeviceArray
    ctx = default_context()
    m = _mandrill()
    d = ctx.to_device(m)
    y = colfilter(d, biort('near_sym_a')[0])
    assert isinstance(y, DeviceArray)
    assert_close(y.get(), o.colfilter(m, biort('near_sym_a')[0]), LOW_TOL)


@pytest.mark.gpu
@pytest.mark.parametrize('rows', [24, 40, 132, 256])
def test_marching_kernels_all_tap_buckets(rows):
    """The register-window variants (lanes along a contiguous inner dimension of >= 64, window
    reach shorter than the signal) and the fall-back to one output per thread, for every tap
    bucket and for lengths outside them, float32 and float64."""
    rs = np.random.RandomState(rows)
    for dt, tol in ((np.float32, LOW_TOL), (np.float64, F64_TOL)):
        X = rs.standard_normal((rows, 192)).astype(dt)
        for m in (1, 3, 5, 7, 8, 9, 13, 19, 20, 21, 32):
            h = rs.standard_normal(m)
            assert_close(colfilter(X, h), o.colfilter(X, h), tol, 'colfilter m=%d' % m)
        for m in (2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 32):
            ha, hb = rs.standard_normal(m), rs.standard_normal(m)
            if rows % 4 == 0:
                assert_close(coldfilt(X, ha, hb), o.coldfilt(X, ha, hb), tol, 'coldfilt m=%d' % m)
                assert_close(coldfilt(X, ha, -hb), o.coldfilt(X, ha, -hb), tol, 'coldfilt m=%d flipped' % m)
            assert_close(colifilt(X, ha, hb), o.colifilt(X, ha, hb), tol, 'colifilt m=%d' % m)
            assert_close(colifilt(X, ha, -hb), o.colifilt(X, ha, -hb), tol, 'colifilt m=%d flipped' % m)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [256, 260, 1000, 1024, 1028, 2052, 4096])
def test_row_kernels_filter_axis_contiguous(n):
    """The LDS-row kernels (filters.hip: k_colfilter_rows / k_coldfilt_rows / k_colifilt_rows): the filter axis is the
    contiguous one -- what the transforms' row passes and axis_*(X, ..., axis=1) run.  Every tap bucket and lengths outside
    them, float32 and float64, rows that are whole segments of 1024 outputs, partial ones and one sample over; edge padding
    and crops (in fours: the 16-byte-store path; otherwise the scalar one) and accumulation into an existing array."""
    from dtcwt_amd.hip import default_context
    from dtcwt_amd.hip import lowlevel as ll
    ctx = default_context()
    rs = np.random.RandomState(n)

    def ref(fn, X, pad, crop, *taps):
        Xp = np.pad(X, ((0, 0), tuple(pad)), mode='edge')
        Y = fn(np.ascontiguousarray(Xp.T), *taps).T
        return Y[:, crop[0]:Y.shape[1] - crop[1]]
    for dt, tol in ((np.float32, LOW_TOL), (np.float64, F64_TOL)):
        X = rs.standard_normal((5, n)).astype(dt)
        d = ctx.to_device(X)
        for m in (1, 5, 7, 8, 13, 19, 20, 21):
            h = rs.standard_normal(m)
            assert_close(ll.axis_colfilter(d, h, axis=1).get(), ref(o.colfilter, X, (0, 0), (0, 0), h), tol, 'colfilter rows m=%d' % m)
        h = rs.standard_normal(7)
        for pad, crop in (((0, 0), (4, 8)), ((1, 1), (1, 1)), ((2, 2), (0, 4)), ((0, 0), (3, 0))):
            got = ll.axis_colfilter(d, h, axis=1, pad=pad, crop=crop).get()
            assert_close(got, ref(o.colfilter, X, pad, crop, h), tol, 'colfilter rows pad %r crop %r' % (pad, crop))
        acc = ctx.to_device(np.ones((5, n), dt))
        ll.axis_colfilter(d, h, axis=1, out=acc, accumulate=True)
        assert_close(acc.get(), 1 + ref(o.colfilter, X, (0, 0), (0, 0), h), tol, 'colfilter rows accumulate')
        for m in (2, 6, 10, 14, 18, 20, 22):
            ha, hb = rs.standard_normal(m), rs.standard_normal(m)
            for sgn in (1, -1):
                assert_close(ll.axis_coldfilt(d, ha, sgn * hb, axis=1).get(), ref(o.coldfilt, X, (0, 0), (0, 0), ha, sgn * hb), tol, 'coldfilt rows m=%d' % m)
                assert_close(ll.axis_colifilt(d, ha, sgn * hb, axis=1).get(), ref(o.colifilt, X, (0, 0), (0, 0), ha, sgn * hb), tol, 'colifilt rows m=%d' % m)
        q = qshift('qshift_b')
        for pad, crop in (((2, 2), (0, 0)), ((0, 0), (2, 2)), ((0, 0), (4, 4))):
            if (n + pad[0] + pad[1]) % 4 == 0:
                assert_close(ll.axis_coldfilt(d, q[1], q[0], axis=1, pad=pad, crop=crop).get(), ref(o.coldfilt, X, pad, crop, q[1], q[0]), tol, 'coldfilt rows pad/crop')
            assert_close(ll.axis_colifilt(d, q[3], q[2], axis=1, pad=pad, crop=crop).get(), ref(o.colifilt, X, pad, crop, q[3], q[2]), tol, 'colifilt rows pad/crop')
