"""CPU restatement of the reference's DT-CWT image registration (rjw57/dtcwt,
dtcwt/registration.py), the consumer of pyramids that SURVEY.md section 8(f) row 2 names.

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by dtcwt_amd.  Parity PINNED:
oracle/check_registration_against_reference.py compares every function here with the
imported reference, and tests/golden/registration.npz holds outputs of the reference itself.

Written per pixel in index form (what the device kernels compute), not as the reference's
whole-array slicing:

  phase gradients (registration.py:31-75), for w = (wx, wy) the subband's expected shift
      Sx[y, x] = (a[y, x+1] conj a[y, x] + b[y, x+1] conj b[y, x]) exp(-j wx),  0 <= x < W-1
      dx[y, x] = wx + angle(Sx[y, 0])                    x = 0
                 wx + angle((Sx[y, x-1] + Sx[y, x]) / 2)  0 < x < W-1
                 wx + angle(Sx[y, W-2])                  x = W-1          (dy alike down the rows)
      dt       = angle(b conj a)
  confidence (:83-137): the four diagonal neighbours with edge replication
      C = |sum conj(a_n) b_n|^2 / (eps + sum |a_n|^3 + |b_n|^3)
  Q-tilde (:140-214): t = (W dx, H dy, x' W dx, x' H dy, y' W dx, y' H dy, -dt), x' = x (1/W),
      27 values: t_r t_c over the upper triangle of 6x6 (row major), then t_r t_6; times C^2,
      summed over the six subbands
  solve (:216-250): only the upper triangle of Q is filled (:231-232), so Q a = -q is a back
      substitution.
"""
import numpy as np

from oracle.dtcwt_oracle import Pyramid, reflect_index
from oracle import sampling_oracle as so

#: expected horizontal / vertical phase shift of each subband (registration.py:29)
EXPECTED_SHIFTS = np.array(((-1, -3), (-3, -3), (-3, -1), (-3, 1), (-3, 3), (-1, 3))) * np.pi / 2.15

_TRIU = list(zip(*np.triu_indices(6)))


def phasegradient(sb1, sb2, w=None):
    """-> (dy, dx, dt)   (registration.py:31-75)"""
    wx, wy = (0, 0) if w is None else w
    if sb1.size != sb2.size:
        raise ValueError('Subbands should have identical size')

    def along(a, b, wv):            # gradient along the last axis
        S = (a[..., 1:] * np.conj(a[..., :-1]) + b[..., 1:] * np.conj(b[..., :-1])) * np.exp(-1j * wv)
        mid = 0.5 * (S[..., :-1] + S[..., 1:])
        return np.angle(np.concatenate((S[..., :1], mid, S[..., -1:]), axis=-1)) + wv

    dx = along(sb1, sb2, wx)
    dy = along(sb1.T, sb2.T, wy).T
    dt = np.angle(sb2 * np.conj(sb1))
    return dy, dx, dt


def confidence(sb1, sb2, epsilon=1e-6):
    """registration.py:83-137"""
    if sb1.size != sb2.size:
        raise ValueError('Subbands should have identical size')
    h, w = sb1.shape
    num, den = 0.0, epsilon
    for oy in (-1, 1):
        yi = np.clip(np.arange(h) + oy, 0, h - 1)
        for ox in (-1, 1):
            xi = np.clip(np.arange(w) + ox, 0, w - 1)
            u, v = sb1[np.ix_(yi, xi)], sb2[np.ix_(yi, xi)]
            num = num + np.conj(u) * v
            au, av = np.abs(u), np.abs(v)
            den = den + (au * au * au + av * av * av)
    an = np.abs(num)
    return an * an / den


def qtilde_level(h1, h2):
    """One level of `qtildematrices` (registration.py:166-212): (H, W, 27)."""
    H, W = h1.shape[:2]
    xs, ys = np.meshgrid(np.arange(W) * (1.0 / W), np.arange(H) * (1.0 / H))
    total = None
    for sb in range(h1.shape[2]):
        a, b = h1[:, :, sb], h2[:, :, sb]
        C = confidence(a, b)
        dy, dx, dt = phasegradient(a, b, EXPECTED_SHIFTS[sb, :])
        dx = dx * W
        dy = dy * H
        t = (dx, dy, xs * dx, xs * dy, ys * dx, ys * dy, -dt)
        Qt = np.zeros(dx.shape[:2] + (27,))
        for e, (r, c) in enumerate(_TRIU):
            Qt[:, :, e] = t[r] * t[c]
        for r in range(6):
            Qt[:, :, 21 + r] = t[r] * t[6]
        Qt *= (C ** 2)[:, :, None]
        total = Qt if total is None else total + Qt
    return total


def qtildematrices(t_ref, t_target, levels):
    """registration.py:140-214"""
    return [qtilde_level(t_ref.highpasses[l], t_target.highpasses[l]) for l in levels]


def solvetransform(Qt):
    """a = -Q^{-1} q with the Q the reference builds: upper triangle only (registration.py:216-250)."""
    Qt = np.asarray(Qt, dtype=np.float64)
    a = np.zeros(Qt.shape[:-1] + (6,))
    U = np.zeros(Qt.shape[:-1] + (6, 6))
    for e, (r, c) in enumerate(_TRIU):
        U[..., r, c] = Qt[..., e]
    q = Qt[..., 21:]
    for r in range(5, -1, -1):
        acc = -q[..., r]
        for c in range(r + 1, 6):
            acc = acc - U[..., r, c] * a[..., c]
        a[..., r] = acc / U[..., r, r]
    return a


def boxfilter(X, kernel_size):
    """registration.py:417-446: running sum with symmetric extension, axis 0 then axis 1."""
    if kernel_size % 2 == 0:
        raise ValueError('Kernel size must be odd')
    for axis in range(2):
        n = X.shape[axis]
        out = X
        for d in range(1, 1 + (kernel_size - 1) // 2):
            out = out + np.take(X, reflect_index(np.arange(n) + d, n), axis=axis)
            out = out + np.take(X, reflect_index(np.arange(n) - d, n), axis=axis)
        X = out / kernel_size
    return X


def _unit_grid(h, w):
    """float32 pixel coordinates in units of the image size (:378-379, :401-402, :410-411)."""
    return np.meshgrid(np.arange(0, w, dtype=np.float32) / w, np.arange(0, h, dtype=np.float32) / h)


def velocityfield(avecs, shape, method=None):
    """registration.py:374-395"""
    h, w = avecs.shape[:2]
    pxs, pys = _unit_grid(h, w)
    vxs = avecs[:, :, 0] + avecs[:, :, 2] * pxs + avecs[:, :, 4] * pys
    vys = avecs[:, :, 1] + avecs[:, :, 3] * pxs + avecs[:, :, 5] * pys
    return so.rescale(vxs, shape, method), so.rescale(vys, shape, method)


def warphighpass(Yh, avecs, method=None):
    """registration.py:397-408"""
    X, Y = _unit_grid(Yh.shape[0], Yh.shape[1])
    vxs, vys = velocityfield(avecs, Yh.shape, method)
    return so.sample_highpass(Yh, (X + vxs) * Yh.shape[1], (Y + vys) * Yh.shape[0], method)


def warp(I, avecs, method=None):
    """registration.py:410-415"""
    X, Y = _unit_grid(I.shape[0], I.shape[1])
    vxs, vys = velocityfield(avecs, I.shape, method)
    return so.sample(I, (X + vxs) * I.shape[1], (Y + vys) * I.shape[0], method)


def warptransform(t, avecs, levels, method=None):
    """registration.py:274-299"""
    hp = list(t.highpasses)
    for l in levels:
        hp[l] = warphighpass(hp[l], avecs, method)
    return Pyramid(t.lowpass, tuple(hp), t.scales)


def default_levels(nlevels):
    """The level schedule of `estimatereg` (registration.py:330-337)."""
    levels = [[x for x in range(nlevels - 1, nlevels - 3, -1) if x >= 0]]
    for s in np.arange(nlevels - 1, 0, -0.5):
        ref = [int(np.floor(s)) - x for x in range(2) if s - x >= 2]
        if len(ref) >= 2:
            levels.append(ref)
    return levels


def estimatereg(source, reference, regshape=None, levels=None):
    """registration.py:301-372"""
    nlevels = len(source.highpasses)
    shape = (source.highpasses[3].shape[:2] if regshape is None else tuple(regshape[:2])) + (6,)
    avecs = np.zeros(shape)
    if levels is None:
        levels = default_levels(nlevels)
    Qt = np.sum([np.sum(np.sum(x, axis=0), axis=0) for x in qtildematrices(source, reference, levels[0])], axis=0)
    avecs[:, :, :] = solvetransform(Qt)
    for est in levels[1:]:
        warped = warptransform(source, avecs, est, 'bilinear')
        all_qts = qtildematrices(warped, reference, est)
        if len(all_qts) < 1:
            continue
        qts = np.zeros(avecs.shape[:2] + all_qts[0].shape[2:])
        for x in all_qts:
            qts += so.rescale(boxfilter(x, 3), avecs.shape[:2], 'bilinear')
        avecs += solvetransform(qts)
    return avecs
