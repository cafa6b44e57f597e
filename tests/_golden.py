"""Readers for tests/golden/*.npz (written by tests/golden/make_golden.py from the
reference itself) and the comparison helpers shared by the CPU and GPU suites."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_cache = {}


def load(name):
    if name not in _cache:
        with np.load(os.path.join(GOLDEN, name + '.npz')) as f:
            _cache[name] = {k: f[k] for k in f.files}
    return _cache[name]


def proj(a):
    """Same projection as make_golden.proj."""
    a = np.asarray(a)
    flat = a.reshape(-1).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    rs = np.random.RandomState(flat.size % 100003)
    out = []
    for _ in range(8):
        w = rs.randint(0, 2, size=flat.size) * 2.0 - 1.0
        out.append(np.dot(flat, w))
    return np.array(out)


def rel_err(a, b):
    """max|a-b| / max|b| (the parity measure of SURVEY.md section 7.3 item 5)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max()) / scale


def check_stored(store, key, a, tol, check_dtype=True):
    """Compare array *a* with the fixture entry *key* (whole data or projections)."""
    a = np.asarray(a)
    shape = tuple(int(x) for x in store[key + '/shape'])
    assert a.shape == shape, '%s: shape %s, golden %s' % (key, a.shape, shape)
    if check_dtype:
        assert str(a.dtype) == str(store[key + '/dtype']), '%s: dtype %s, golden %s' % (
            key, a.dtype, store[key + '/dtype'])
    if key + '/data' in store:
        e = rel_err(a, store[key + '/data'])
        assert e <= tol, '%s: rel err %.3e > %g' % (key, e, tol)
    else:
        want = store[key + '/proj']
        got = proj(a)
        # a projection sums n terms of size <= absmax with random signs
        scale = float(store[key + '/absmax']) * np.sqrt(max(a.size, 1)) + 1e-30
        e = float(np.abs(got - want).max()) / scale
        assert e <= tol, '%s: projection err %.3e > %g' % (key, e, tol)


def check_pyramid(store, key, p, tol, check_dtype=True):
    check_stored(store, key + '/Yl', p.lowpass, tol, check_dtype)
    nl = int(store[key + '/nlevels'])
    assert len(p.highpasses) == nl
    for l in range(nl):
        if key + '/Yh%d/none' % l in store:
            assert p.highpasses[l] is None
        else:
            check_stored(store, key + '/Yh%d' % l, p.highpasses[l], tol, check_dtype)
    if p.scales is not None:
        for l in range(nl):
            check_stored(store, key + '/Ys%d' % l, p.scales[l], tol, check_dtype)


class StoredPyramid(object):
    """Rebuild a Pyramid-like object from whole-data fixture entries."""

    def __init__(self, store, key):
        self.lowpass = store[key + '/Yl/data']
        nl = int(store[key + '/nlevels'])
        self.highpasses = tuple(None if key + '/Yh%d/none' % l in store else store[key + '/Yh%d/data' % l]
                                for l in range(nl))
        self.scales = None


def summarise_mat(M, apron=8):
    """The reference tests' reduction (tests/util.py:46-60 there), restated."""
    def mean(a, axis):
        return np.expand_dims(np.mean(a, axis=axis), axis)
    centre = mean(mean(M[apron:-apron, apron:-apron, ...], 0), 1)
    return np.vstack((
        np.hstack((M[:apron, :apron, ...], mean(M[:apron, apron:-apron, ...], 1), M[:apron, -apron:, ...])),
        np.hstack((mean(M[apron:-apron, :apron, ...], 0), centre, mean(M[apron:-apron, -apron:, ...], 0))),
        np.hstack((M[-apron:, :apron, ...], mean(M[-apron:, apron:-apron, ...], 1), M[-apron:, -apron:, ...])),
    ))


def parse_case(name):
    return name.split('-')
