"""CPU oracle for the DT-CWT filter-bank hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the algorithm of rjw57/dtcwt's ``dtcwt.numpy``
backend for the path SURVEY.md section 8 names (colfilter / coldfilt / colifilt, the
quad/cube <-> complex packings and the 1-D/2-D/3-D level loops).  It exists so that the
HIP kernels can be checked against an independent statement of the same arithmetic.

* It is NOT part of the product.  Nothing under ``dtcwt_amd/`` imports it.  Only
  ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
  may import it, and only as the checker / the timed CPU baseline.
* Parity is PINNED: ``oracle/check_against_reference.py`` (run in the build container,
  where the reference is importable from /root/reference) compares every function here
  with the reference on random and adversarial inputs (<= 1e-12 in float64, bit-exact
  for most), and ``tests/golden/*.npz`` holds outputs of the reference itself that the
  CPU test-suite replays against this file on every run.
* MATLAB parity (``tests/verification.npz`` of the reference) is unpinned: that file is
  absent from the reference mount; it is inherited transitively
  through ``dtcwt.numpy``.

Every function cites the reference lines it restates (paths relative to
/root/reference).  The formulation is index-algebraic (explicit reflected gather
indices) rather than the reference's extend-then-convolve, but the
floating-point operation order per output sample (tap-serial multiply then add, taps cast
to the signal dtype) is the reference's, so float32 results agree to the last bit in
almost all cases.
"""
import logging

import numpy as np

__all__ = [
    'reflect_index', 'colfilter', 'coldfilt', 'colifilt', 'q2c', 'c2q', 'cube2c',
    'c2cube', 'c2q1d', 'Pyramid', 'Transform1d', 'Transform2d', 'Transform3d',
    'asfarray', 'complex_dtype_for',
]


# --------------------------------------------------------------------------- helpers
def asfarray(X):
    """float32/float64 pass through, anything else becomes float64.
    Restates dtcwt/utils.py:98-105."""
    X = np.asanyarray(X)
    if X.dtype in (np.float32, np.float64):
        return X
    if np.issubdtype(X.dtype, np.complexfloating):
        return X
    return X.astype(np.float64)


def complex_dtype_for(X):
    """float32 -> complex64, float64 -> complex128 (dtcwt/utils.py:107-124)."""
    dt = asfarray(X).dtype
    if dt in (np.complex64, np.complex128):
        return dt
    return np.complex64 if dt == np.float32 else np.complex128


def _taps(h, dtype):
    """Filter as a flat vector in the signal's dtype (dtcwt/numpy/lowlevel.py:33)."""
    return np.asarray(h, dtype=np.float64).reshape(-1).astype(dtype)


def reflect_index(i, n):
    """Half-sample symmetric reflection of integer indices into [0, n).

    Equals ``reflect(i, -0.5, n-0.5)`` of dtcwt/utils.py:136-153 for integer ``i``:
    ... 1 0 | 0 1 ... n-1 | n-1 n-2 ...   Multi-bounce safe through the modulus."""
    j = np.mod(np.asarray(i, dtype=np.int64), 2 * n)
    return np.where(j < n, j, 2 * n - 1 - j)


def _gather_fir(X, taps, index_of_tap, nout):
    """acc[i] = sum_k taps[k] * X[index_of_tap(k)[i]] accumulated tap-serially in X's
    dtype: the hot loop of dtcwt/numpy/lowlevel.py:40-41 expressed on gathered rows."""
    acc = np.zeros((nout,) + X.shape[1:], dtype=X.dtype)
    for k in range(len(taps)):
        acc += X[index_of_tap(k)] * taps[k]
    return acc


# --------------------------------------------------------------------------- filters
def colfilter(X, h):
    """Undecimated symmetric-extension FIR down axis 0.
    Restates dtcwt/numpy/lowlevel.py:47-80 (and _column_convolve :23-45).

    Y[i] = sum_k h[k] * X[rho_r(i + m - 1 - k - m//2)],  i in [0, r) for odd m,
    [0, r] for even m."""
    X = asfarray(X)
    h = _taps(h, X.dtype)
    r = X.shape[0]
    m = h.shape[0]
    m2 = m // 2
    nout = r if (m % 2) else r + 1
    i = np.arange(nout)
    return _gather_fir(X, h, lambda k: reflect_index(i + (m - 1 - k - m2), r), nout)


def _check_dual(X, ha, hb, mult, what):
    if X.shape[0] % mult != 0:
        raise ValueError('No. of rows in X must be a multiple of %d' % mult)
    if np.shape(ha) != np.shape(hb):
        raise ValueError('Shapes of ha and hb must be the same')
    if np.asarray(ha).reshape(-1).shape[0] % 2 != 0:
        raise ValueError('Lengths of ha and hb must be even')


def coldfilt(X, ha, hb):
    """Dual-tree decimating filter down axis 0 (rows % 4 == 0, even-length taps).
    Restates dtcwt/numpy/lowlevel.py:82-154.

    With p = m/2 and b_k = 4(i+p-1-k) - m, i in [0, r/4):
      A[i] = sum_k ha[2k] X[rho(b_k+4)] + ha[2k+1] X[rho(b_k+2)]
      B[i] = sum_k hb[2k] X[rho(b_k+5)] + hb[2k+1] X[rho(b_k+3)]
    and (Y[2i], Y[2i+1]) = (A, B) if sum(ha*hb) > 0 else (B, A)."""
    X = asfarray(X)
    _check_dual(X, ha, hb, 4, 'coldfilt')
    ha64 = np.asarray(ha, dtype=np.float64).reshape(-1)
    hb64 = np.asarray(hb, dtype=np.float64).reshape(-1)
    ha_t = ha64.astype(X.dtype)
    hb_t = hb64.astype(X.dtype)
    r = X.shape[0]
    m = ha_t.shape[0]
    p = m // 2
    q = r // 4
    i = np.arange(q)

    def base(k):
        return 4 * (i + p - 1 - k) - m

    # The reference adds two separately accumulated half-length convolutions
    # (lowlevel.py:151-152); keep that association for float32 bit-parity.
    A = (_gather_fir(X, ha_t[0::2], lambda k: reflect_index(base(k) + 4, r), q) +
         _gather_fir(X, ha_t[1::2], lambda k: reflect_index(base(k) + 2, r), q))
    B = (_gather_fir(X, hb_t[0::2], lambda k: reflect_index(base(k) + 5, r), q) +
         _gather_fir(X, hb_t[1::2], lambda k: reflect_index(base(k) + 3, r), q))

    Y = np.zeros((r // 2,) + X.shape[1:], dtype=X.dtype)
    if np.sum(ha64 * hb64) > 0:
        Y[0::2], Y[1::2] = A, B
    else:
        Y[0::2], Y[1::2] = B, A
    return Y


def colifilt(X, ha, hb, mimic_row0_quirk=False):
    """Dual-tree interpolating filter down axis 0 (rows % 2 == 0, even-length taps).
    Restates dtcwt/numpy/lowlevel.py:156-260.

    n = m/2 taps per polyphase component; for j in [0, r/2), k in [0, n), jj = j+n-1-k:
      m/2 even: t = 3+2jj; (ta, tb) = (t, t-1) if sum(ha*hb) > 0 else (t-1, t)
          Y[4j+0] += hae[k] X[rho(tb-2-m2)]   Y[4j+1] += hbe[k] X[rho(ta-2-m2)]
          Y[4j+2] += hao[k] X[rho(tb-m2)]     Y[4j+3] += hbo[k] X[rho(ta-m2)]
      m/2 odd:  t = 2+2jj
          Y[4j+0] += hao[k] X[rho(tb-m2)]     Y[4j+1] += hbo[k] X[rho(ta-m2)]
          Y[4j+2] += hae[k] X[rho(tb-m2)]     Y[4j+3] += hbe[k] X[rho(ta-m2)]

    ``mimic_row0_quirk`` reproduces lowlevel.py:202, which returns zeros whenever all
    non-zeros of X lie in row 0 (it tests the row *indices* of the non-zeros)."""
    X = asfarray(X)
    _check_dual(X, ha, hb, 2, 'colifilt')
    ha64 = np.asarray(ha, dtype=np.float64).reshape(-1)
    hb64 = np.asarray(hb, dtype=np.float64).reshape(-1)
    ha_t = ha64.astype(X.dtype)
    hb_t = hb64.astype(X.dtype)
    r = X.shape[0]
    m = ha_t.shape[0]
    m2 = m // 2
    n = m2
    Y = np.zeros((2 * r,) + X.shape[1:], dtype=X.dtype)
    if mimic_row0_quirk and not np.any(np.nonzero(X)[0]):
        return Y
    hao, hae = ha_t[0::2], ha_t[1::2]
    hbo, hbe = hb_t[0::2], hb_t[1::2]
    q = r // 2
    j = np.arange(q)
    pos = np.sum(ha64 * hb64) > 0

    def tt(k, t0):
        t = t0 + 2 * (j + n - 1 - k)
        return (t, t - 1) if pos else (t - 1, t)   # (ta, tb)

    if m2 % 2 == 0:
        Y[0::4] = _gather_fir(X, hae, lambda k: reflect_index(tt(k, 3)[1] - 2 - m2, r), q)
        Y[1::4] = _gather_fir(X, hbe, lambda k: reflect_index(tt(k, 3)[0] - 2 - m2, r), q)
        Y[2::4] = _gather_fir(X, hao, lambda k: reflect_index(tt(k, 3)[1] - m2, r), q)
        Y[3::4] = _gather_fir(X, hbo, lambda k: reflect_index(tt(k, 3)[0] - m2, r), q)
    else:
        Y[0::4] = _gather_fir(X, hao, lambda k: reflect_index(tt(k, 2)[1] - m2, r), q)
        Y[1::4] = _gather_fir(X, hbo, lambda k: reflect_index(tt(k, 2)[0] - m2, r), q)
        Y[2::4] = _gather_fir(X, hae, lambda k: reflect_index(tt(k, 2)[1] - m2, r), q)
        Y[3::4] = _gather_fir(X, hbe, lambda k: reflect_index(tt(k, 2)[0] - m2, r), q)
    return Y


# ------------------------------------------------------------------ quad/cube packing
def q2c(y):
    """2x2 real quads -> two complex subbands (dtcwt/numpy/transform2d.py:301-322).
    Quad (a b / c d): p = (a + jb)/sqrt2, q = (d - jc)/sqrt2, out = (p - q, p + q)."""
    y = asfarray(y)
    cdt = complex_dtype_for(y)
    s = np.sqrt(0.5)
    j2 = np.array([s, 1j * s]).astype(cdt)
    p = y[0::2, 0::2] * j2[0] + y[0::2, 1::2] * j2[1]
    q = y[1::2, 1::2] * j2[0] - y[1::2, 0::2] * j2[1]
    return np.dstack((p - q, p + q))


def c2q(w, gain):
    """Two complex subbands -> 2x2 real quads, scaled by gain*sqrt(1/2)
    (dtcwt/numpy/transform2d.py:324-350).  Stays in the subbands' precision, as the
    reference does under NumPy 1.x."""
    w = np.asanyarray(w)
    rdt = w.real.dtype
    x = np.zeros((w.shape[0] << 1, w.shape[1] << 1), dtype=rdt)
    sc = (np.sqrt(0.5) * np.asarray(gain, dtype=np.float64)).astype(rdt)
    P = w[:, :, 0] * sc[0] + w[:, :, 1] * sc[1]
    Q = w[:, :, 0] * sc[0] - w[:, :, 1] * sc[1]
    x[0::2, 0::2] = P.real
    x[0::2, 1::2] = P.imag
    x[1::2, 0::2] = Q.imag
    x[1::2, 1::2] = -Q.real
    return x


def c2q1d(x):
    """Complex rows -> interleaved real/imag rows (dtcwt/numpy/transform1d.py:186-196)."""
    x = np.asanyarray(x)
    z = np.zeros((x.shape[0] * 2, x.shape[1]), dtype=x.real.dtype)
    z[0::2] = x.real
    z[1::2] = x.imag
    return z


def cube2c(y):
    """2x2x2 real octets -> four complex numbers (dtcwt/numpy/transform3d.py:532-579)."""
    y = asfarray(y)
    cdt = complex_dtype_for(y)
    half = y.dtype.type(0.5)
    A = y[0::2, 0::2, 0::2]; B = y[0::2, 1::2, 0::2]
    C = y[1::2, 0::2, 0::2]; D = y[1::2, 1::2, 0::2]
    E = y[0::2, 0::2, 1::2]; F = y[0::2, 1::2, 1::2]
    G = y[1::2, 0::2, 1::2]; H = y[1::2, 1::2, 1::2]
    z = np.empty(A.shape + (4,), dtype=cdt)
    z[..., 0].real = (A - G - D - F) * half; z[..., 0].imag = (B - H + C + E) * half
    z[..., 1].real = (A - G + D + F) * half; z[..., 1].imag = (-B + H + C + E) * half
    z[..., 2].real = (A + G + D - F) * half; z[..., 2].imag = (B + H - C + E) * half
    z[..., 3].real = (A + G - D + F) * half; z[..., 3].imag = (-B - H - C + E) * half
    return z


def c2cube(z):
    """Inverse of cube2c (dtcwt/numpy/transform3d.py:581-619)."""
    z = np.asanyarray(z)
    rdt = z.real.dtype
    pr, pi = z[..., 0].real, z[..., 0].imag
    qr, qi = z[..., 1].real, z[..., 1].imag
    rr, ri = z[..., 2].real, z[..., 2].imag
    sr, si = z[..., 3].real, z[..., 3].imag
    y = np.zeros(tuple(2 * s for s in z.shape[:3]), dtype=rdt)
    y[0::2, 0::2, 0::2] = (pr + qr + rr + sr)
    y[1::2, 0::2, 1::2] = (-pr - qr + rr + sr)
    y[1::2, 1::2, 0::2] = (-pr + qr + rr - sr)
    y[0::2, 1::2, 1::2] = (-pr + qr - rr + sr)
    y[0::2, 1::2, 0::2] = (pi - qi + ri - si)
    y[1::2, 1::2, 1::2] = (-pi + qi + ri - si)
    y[1::2, 0::2, 0::2] = (pi + qi - ri - si)
    y[0::2, 0::2, 1::2] = (pi + qi + ri + si)
    return y * rdt.type(0.5)


# ---------------------------------------------------------------------------- Pyramid
class Pyramid(object):
    """lowpass / highpasses / scales holder (dtcwt/numpy/common.py:5-32)."""

    def __init__(self, lowpass, highpasses, scales=None):
        self.lowpass = asfarray(lowpass)
        self.highpasses = tuple(asfarray(x) if x is not None else None for x in highpasses)
        self.scales = tuple(asfarray(x) for x in scales) if scales is not None else None


def _unpack_biort(b):
    if len(b) == 4:
        h0o, g0o, h1o, g1o = b
        return h0o, g0o, h1o, g1o, None, None
    if len(b) == 6:
        return tuple(b)
    raise ValueError('Biort wavelet must have 6 or 4 components.')


def _unpack_qshift(q):
    if len(q) == 8:
        return tuple(q) + (None,) * 4
    if len(q) == 12:
        return tuple(q)
    raise ValueError('Qshift wavelet must have 12 or 8 components.')


def _edge_pad(a, axis, n=1):
    """Replicate the first and last hyperplane n times along axis
    (transform2d.py:134-140, transform3d.py:322-335)."""
    first = np.take(a, [0] * n, axis=axis)
    last = np.take(a, [a.shape[axis] - 1] * n, axis=axis)
    return np.concatenate((first, a, last), axis=axis)


# ------------------------------------------------------------------------ 2-D driver
class Transform2d(object):
    """Level loop of the 2-D DT-CWT (dtcwt/numpy/transform2d.py:18-295).

    ``biort``/``qshift`` are tuples of tap vectors (the oracle has no wavelet file
    loader of its own; tests pass tables from dtcwt_amd.coeffs)."""

    def __init__(self, biort, qshift):
        self.biort = biort
        self.qshift = qshift

    # filtering along axis 1 == the reference's  colfilter(A.T, h).T
    @staticmethod
    def _rows(fn, A, *h):
        return fn(A.T, *h).T

    def forward(self, X, nlevels=3, include_scale=False):
        """transform2d.py:40-188."""
        h0o, g0o, h1o, g1o, h2o, g2o = _unpack_biort(self.biort)
        (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b, h2a, h2b, g2a, g2b) = \
            _unpack_qshift(self.qshift)
        bp1 = len(self.biort) >= 6
        bp2 = len(self.qshift) >= 12

        X = np.atleast_2d(asfarray(X))
        if X.ndim >= 3:
            raise ValueError('The entered image is {0}, which is invalid for the 2D '
                             'transform.'.format('x'.join(str(s) for s in X.shape)))
        original_size = X.shape
        if X.shape[0] % 2:                                   # :86-94
            X = np.concatenate((X, X[-1:, :]), axis=0)
        if X.shape[1] % 2:
            X = np.concatenate((X, X[:, -1:]), axis=1)
        extended_size = X.shape

        if nlevels == 0:
            return Pyramid(X, (), ()) if include_scale else Pyramid(X, ())

        Yh = [None] * nlevels
        Yscale = [None] * nlevels
        cdt = complex_dtype_for(X)
        rows = self._rows

        # level 1, :112-130
        Lo = colfilter(X, h0o)
        Hi = colfilter(X, h1o)
        LoLo = rows(colfilter, Lo, h0o)
        Yh[0] = np.zeros((LoLo.shape[0] >> 1, LoLo.shape[1] >> 1, 6), dtype=cdt)
        Yh[0][:, :, 0:6:5] = q2c(rows(colfilter, Hi, h0o))
        Yh[0][:, :, 2:4:1] = q2c(rows(colfilter, Lo, h1o))
        if bp1:
            Ba = colfilter(X, h2o)
            Yh[0][:, :, 1:5:3] = q2c(rows(colfilter, Ba, h2o))
        else:
            Yh[0][:, :, 1:5:3] = q2c(rows(colfilter, Hi, h1o))
        Yscale[0] = LoLo

        # levels >= 2, :132-160
        for level in range(1, nlevels):
            if LoLo.shape[0] % 4:
                LoLo = _edge_pad(LoLo, 0)
            if LoLo.shape[1] % 4:
                LoLo = _edge_pad(LoLo, 1)
            Lo = coldfilt(LoLo, h0b, h0a)
            Hi = coldfilt(LoLo, h1b, h1a)
            if bp2:
                Ba = coldfilt(LoLo, h2b, h2a)
            LoLo = rows(coldfilt, Lo, h0b, h0a)
            Yh[level] = np.zeros((LoLo.shape[0] >> 1, LoLo.shape[1] >> 1, 6), dtype=cdt)
            Yh[level][:, :, 0:6:5] = q2c(rows(coldfilt, Hi, h0b, h0a))
            Yh[level][:, :, 2:4:1] = q2c(rows(coldfilt, Lo, h1b, h1a))
            if bp2:
                Yh[level][:, :, 1:5:3] = q2c(rows(coldfilt, Ba, h2b, h2a))
            else:
                Yh[level][:, :, 1:5:3] = q2c(rows(coldfilt, Hi, h1b, h1a))
            Yscale[level] = LoLo

        if extended_size != original_size:                   # :164-183
            logging.warning('The image entered is now a {0} NOT a {1}.'.format(
                'x'.join(str(s) for s in extended_size),
                'x'.join(str(s) for s in original_size)))
        if include_scale:
            return Pyramid(LoLo, tuple(Yh), tuple(Yscale))
        return Pyramid(LoLo, tuple(Yh))

    def inverse(self, pyramid, gain_mask=None):
        """transform2d.py:190-295."""
        h0o, g0o, h1o, g1o, h2o, g2o = _unpack_biort(self.biort)
        (h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b, h2a, h2b, g2a, g2b) = \
            _unpack_qshift(self.qshift)
        bp1 = len(self.biort) >= 6
        bp2 = len(self.qshift) >= 12
        Z = pyramid.lowpass
        Yh = pyramid.highpasses
        a = len(Yh)
        gain_mask = np.ones((6, a)) if gain_mask is None else np.array(gain_mask)
        rows = self._rows

        level = a
        while level >= 2:                                    # :242-273
            w = Yh[level - 1]
            g = gain_mask[:, level - 1]
            lh = c2q(w[:, :, [0, 5]], g[[0, 5]])
            hl = c2q(w[:, :, [2, 3]], g[[2, 3]])
            hh = c2q(w[:, :, [1, 4]], g[[1, 4]])
            y1 = colifilt(Z, g0b, g0a) + colifilt(lh, g1b, g1a)
            if bp2:
                y2 = colifilt(hl, g0b, g0a)
                y2bp = colifilt(hh, g2b, g2a)
                Z = (rows(colifilt, y1, g0b, g0a) + rows(colifilt, y2, g1b, g1a) +
                     rows(colifilt, y2bp, g2b, g2a))
            else:
                y2 = colifilt(hl, g0b, g0a) + colifilt(hh, g1b, g1a)
                Z = rows(colifilt, y1, g0b, g0a) + rows(colifilt, y2, g1b, g1a)
            S = 2 * np.array(Yh[level - 2].shape[:2])
            if Z.shape[0] != S[0]:
                Z = Z[1:-1, :]
            if Z.shape[1] != S[1]:
                Z = Z[:, 1:-1]
            if np.any(np.array(Z.shape) != S):
                raise ValueError('Sizes of highpasses are not valid for DTWAVEIFM2')
            level -= 1

        if level == 1:                                       # :275-293
            w = Yh[0]
            g = gain_mask[:, 0]
            lh = c2q(w[:, :, [0, 5]], g[[0, 5]])
            hl = c2q(w[:, :, [2, 3]], g[[2, 3]])
            hh = c2q(w[:, :, [1, 4]], g[[1, 4]])
            y1 = colfilter(Z, g0o) + colfilter(lh, g1o)
            if bp1:
                y2 = colfilter(hl, g0o)
                y2bp = colfilter(hh, g2o)
                Z = (rows(colfilter, y1, g0o) + rows(colfilter, y2, g1o) +
                     rows(colfilter, y2bp, g2o))
            else:
                y2 = colfilter(hl, g0o) + colfilter(hh, g1o)
                Z = rows(colfilter, y1, g0o) + rows(colfilter, y2, g1o)
        return Z


# ------------------------------------------------------------------------ 1-D driver
class Transform1d(object):
    """Level loop of the 1-D DT-CWT (dtcwt/numpy/transform1d.py:14-180)."""

    def __init__(self, biort, qshift):
        self.biort = biort
        self.qshift = qshift

    def forward(self, X, nlevels=3, include_scale=False):
        """transform1d.py:26-110."""
        h0o, g0o, h1o, g1o = self.biort[:4]
        h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = self.qshift[:8]
        X = asfarray(X)
        if X.ndim == 1:
            X = X[:, None]
        if X.shape[0] % 2:
            raise ValueError('Size of input X must be a multiple of 2')
        if nlevels == 0:
            return Pyramid(X, (), ()) if include_scale else Pyramid(X, ())
        Yh = [None] * nlevels
        Yscale = [None] * nlevels
        Hi = colfilter(X, h1o)
        Lo = colfilter(X, h0o)
        Yh[0] = Hi[0::2] + 1j * Hi[1::2]
        Yscale[0] = Lo
        for level in range(1, nlevels):
            if Lo.shape[0] % 4:
                Lo = _edge_pad(Lo, 0)
            Hi = coldfilt(Lo, h1b, h1a)
            Lo = coldfilt(Lo, h0b, h0a)
            Yh[level] = Hi[0::2] + 1j * Hi[1::2]
            Yscale[level] = Lo
        cdt = complex_dtype_for(X)
        Yh = [y.astype(cdt) for y in Yh]
        if include_scale:
            return Pyramid(Lo, Yh, Yscale)
        return Pyramid(Lo, Yh)

    def inverse(self, pyramid, gain_mask=None):
        """transform1d.py:112-180."""
        h0o, g0o, h1o, g1o = self.biort[:4]
        h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = self.qshift[:8]
        Lo = pyramid.lowpass
        Yh = pyramid.highpasses
        a = len(Yh)
        gain_mask = np.ones(a) if gain_mask is None else np.asarray(gain_mask)
        level = a - 1
        if level < 0:
            return Lo
        while level >= 1:
            Hi = c2q1d(Yh[level] * Yh[level].real.dtype.type(gain_mask[level]))
            Lo = colifilt(Lo, g0b, g0a) + colifilt(Hi, g1b, g1a)
            if Lo.shape[0] != 2 * Yh[level - 1].shape[0]:
                Lo = Lo[1:-1]
            if Lo.shape[0] != 2 * Yh[level - 1].shape[0] or Lo.shape[1] != Yh[level - 1].shape[1]:
                raise ValueError('Yh sizes are not valid for DTWAVEIFM')
            level -= 1
        Hi = c2q1d(Yh[0] * Yh[0].real.dtype.type(gain_mask[0]))
        Z = colfilter(Lo, g0o) + colfilter(Hi, g1o)
        return Z.flatten() if Z.shape[1] == 1 else Z


# ------------------------------------------------------------------------ 3-D driver
def _along(fn, V, axis, *h):
    """Apply an axis-0 filter along ``axis`` of a volume."""
    return np.moveaxis(fn(np.moveaxis(V, axis, 0), *h), 0, axis)


_OCTANTS = ((0, 1, 0), (1, 0, 0), (1, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 1))
"""(hi on axis0, hi on axis1, hi on axis2) of the seven highpass octants, in the order
the reference concatenates them (transform3d.py:278-289 / :372-383)."""


class Transform3d(object):
    """Level loops of the 3-D DT-CWT (dtcwt/numpy/transform3d.py:14-526), restated
    axis-wise: filter the whole volume along axis 2, then 1, then 0, then pack the seven
    highpass octants with cube2c.  SURVEY.md section 3.3 records that this ordering
    reproduces the reference's slice loops exactly."""

    def __init__(self, biort, qshift, ext_mode=4, mimic_ifm_no_highpass_quirk=False):
        self.biort = biort
        self.qshift = qshift
        self.ext_mode = ext_mode
        # transform3d.py:454-456 assigns the axis-2 result of the highpass-free level-1
        # inverse without transposing it back: cubic volumes come out with axes 0 and 2
        # swapped and non-cubic ones raise.  The oracle (and the HIP path) compute the
        # intended result; the flag reproduces the swap for diffing against the reference.
        self.mimic_ifm_no_highpass_quirk = mimic_ifm_no_highpass_quirk

    def forward(self, X, nlevels=3, include_scale=False, discard_level_1=False):
        """transform3d.py:37-131."""
        X = np.atleast_3d(asfarray(X))
        h0o, g0o, h1o, g1o = self.biort[:4]
        if len(self.biort) not in (4, 6):
            raise ValueError('Biort wavelet must have 6 or 4 components.')
        if len(self.qshift) not in (8, 12):
            raise ValueError('Qshift wavelet must have 12 or 8 components.')
        h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = self.qshift[:8]
        if self.ext_mode not in (4, 8):
            raise ValueError('ext_mode must be one of 4 or 8')
        Yl = X
        Yh = [None] * nlevels
        Yscale = [None] * nlevels
        for level in range(nlevels):
            if level == 0:
                Yl, Yh[0] = self._level1_xfm(Yl, h0o, h1o, not discard_level_1)
            else:
                Yl, Yh[level] = self._level2_xfm(Yl, h0a, h0b, h1a, h1b)
            Yscale[level] = Yl.copy()
        if include_scale:
            return Pyramid(Yl, tuple(Yh), tuple(Yscale))
        return Pyramid(Yl, tuple(Yh))

    def _split(self, V, fn, lo, hi):
        """One volume -> 8 octants keyed (a0, a1, a2); axis order 2, 1, 0."""
        parts = {(): V}
        for axis in (2, 1, 0):
            nxt = {}
            for key, vol in parts.items():
                nxt[(0,) + key] = _along(fn, vol, axis, *lo)
                nxt[(1,) + key] = _along(fn, vol, axis, *hi)
            parts = nxt
        return parts

    def _level1_xfm(self, X, h0o, h1o, want_highpass):
        """transform3d.py:208-289 and :291-315."""
        mult = 2 if self.ext_mode == 4 else 4
        if any(s % mult for s in X.shape):
            raise ValueError('Input shape should be a multiple of %d in each direction '
                             'when self.ext_mode == %d' % (mult, self.ext_mode))
        if not want_highpass:
            L = X
            for axis in (2, 1, 0):
                L = _along(colfilter, L, axis, h0o)
            return L, None
        # Even-length taps (:223-251, e.g. Haar): the reference allocates (N+1)-long
        # halves and writes replicated planes that its own loops then overwrite; the
        # net effect is N -> N+1 samples per axis, Yl = the whole (N+1)^3 LLL block and
        # the highpass octants packed from their first N samples per axis (x0a/x1b/...).
        n0, n1, n2 = X.shape
        parts = self._split(X, colfilter, (h0o,), (h1o,))
        Yl = parts[(0, 0, 0)]
        Yh = np.concatenate([cube2c(parts[o][:n0, :n1, :n2]) for o in _OCTANTS], axis=3)
        return Yl, Yh

    def _level2_xfm(self, X, h0a, h0b, h1a, h1b):
        """transform3d.py:317-383."""
        mult, npad = (4, 1) if self.ext_mode == 4 else (8, 2)
        for axis in range(3):
            if X.shape[axis] % mult:
                X = _edge_pad(X, axis, npad)
        parts = self._split(X, coldfilt, (h0b, h0a), (h1b, h1a))
        Yl = parts[(0, 0, 0)]
        Yh = np.concatenate([cube2c(parts[o]) for o in _OCTANTS], axis=3)
        return Yl, Yh

    def inverse(self, pyramid):
        """transform3d.py:133-206."""
        Yl = pyramid.lowpass
        Yh = pyramid.highpasses
        h0o, g0o, h1o, g1o = self.biort[:4]
        h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = self.qshift[:8]
        nlevels = len(Yh)
        for level in range(nlevels):
            if level == nlevels - 1:
                if Yh[-level - 1] is None:
                    for axis in (1, 0, 2):                  # :442-458
                        Yl = _along(colfilter, Yl, axis, g0o)
                    if self.mimic_ifm_no_highpass_quirk:
                        Yl = Yl.transpose(2, 1, 0)
                elif np.asarray(g0o).reshape(-1).shape[0] % 2 == 0:
                    # even-length taps (:394-398, :437-438): use the first N samples
                    # of the lowpass block, N -> N+1 per axis, drop sample 0 per axis.
                    n0, n1, n2 = (2 * s for s in Yh[-level - 1].shape[:3])
                    Yl = self._merge(Yl[:n0, :n1, :n2], Yh[-level - 1], colfilter,
                                     (g0o,), (g1o,))[1:, 1:, 1:]
                else:
                    Yl = self._merge(Yl, Yh[-level - 1], colfilter, (g0o,), (g1o,))
            else:
                if Yh[-level - 2] is not None:
                    prev = np.array(Yh[-level - 2].shape[:3])
                else:
                    prev = np.array(Yh[-level - 1].shape[:3]) * 2
                Yl = self._merge(Yl, Yh[-level - 1], colifilt, (g0b, g0a), (g1b, g1a))
                cur = np.array(Yh[-level - 1].shape[:3])
                c = 1 if self.ext_mode == 4 else 2          # :505-524
                sl = [slice(None)] * 3
                for axis in range(3):
                    if cur[axis] * 2 != prev[axis]:
                        sl[axis] = slice(c, -c)
                Yl = Yl[tuple(sl)]
        return Yl

    @staticmethod
    def _merge(Yl, Yh, fn, lo, hi):
        """8 octants -> one volume; axis order 1, 0, 2 (transform3d.py:425-435,
        :485-495).  The sums are lo-branch + hi-branch as in the reference."""
        parts = {(0, 0, 0): Yl}
        for n, o in enumerate(_OCTANTS):
            parts[o] = c2cube(Yh[:, :, :, 4 * n:4 * n + 4])
        # axis 1
        p1 = {}
        for a0 in (0, 1):
            for a2 in (0, 1):
                p1[(a0, a2)] = (_along(fn, parts[(a0, 0, a2)], 1, *lo) +
                                _along(fn, parts[(a0, 1, a2)], 1, *hi))
        # axis 0
        p0 = {}
        for a2 in (0, 1):
            p0[a2] = _along(fn, p1[(0, a2)], 0, *lo) + _along(fn, p1[(1, a2)], 0, *hi)
        # axis 2
        return _along(fn, p0[0], 2, *lo) + _along(fn, p0[1], 2, *hi)
