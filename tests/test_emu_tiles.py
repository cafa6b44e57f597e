"""Index algebra of the fused gfx950 tile programs, stepped on the CPU.

tests/emu/emu.hip compiles the same __host__ __device__ phase functions the kernels in
dtcwt_amd/csrc/fused2d.hip call and walks them workgroup by workgroup on the host; here
each level kernel is compared with the oracle's statement of that level.  (The GPU suite
repeats the comparison on the real kernels through the C ABI.)
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_SRC = os.path.join(HERE, 'emu', 'emu.hip')
EMU_LIB = os.path.join(HERE, 'emu', 'libdtcwt_emu.so')
HIPCC = '/opt/rocm/bin/hipcc'

TOL = 2e-6


def _build():
    deps = [EMU_SRC] + [
        os.path.join(ROOT, 'dtcwt_amd', 'csrc', f) for f in
        ('fused2d_tiles.hpp', 'fused2d_tiles_v2.hpp', 'fused2d_table.hpp', 'fused3d_tiles.hpp', 'fused3d_inv_tiles.hpp',
         'march2d.hpp')]
    if os.path.exists(EMU_LIB) and all(os.path.getmtime(EMU_LIB) >= os.path.getmtime(d) for d in deps):
        return
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available to build the emulator')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O2', '-std=c++17', '-fPIC', '-shared',
                           '-I' + os.path.join(ROOT, 'dtcwt_amd', 'csrc'), '-I' + os.path.join(ROOT, 'include'),
                           EMU_SRC, '-o', EMU_LIB])


@pytest.fixture(scope='module')
def emu():
    _build()
    return ctypes.CDLL(EMU_LIB)


def _d(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def rel(a, b):
    return np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max() / max(np.abs(b).max(), 1e-30)


def emu_fwd1(emu, X, h0o, h1o):
    B, r, c = X.shape
    R, C = r + (r & 1), c + (c & 1)
    lolo = np.full((B, R, C), np.nan, np.float32)
    yh = np.full((B, R // 2, C // 2, 12), np.nan, np.float32)
    h0, p0 = _d(h0o); h1, p1 = _d(h1o)
    rc = emu.emu_fwd1(len(h0), len(h1), _f(X), _f(lolo), _f(yh), B, r, c, p0, p1)
    assert rc == 0
    return lolo, yh.view(np.complex64)


def emu_fwd2(emu, X, q):
    B, r, c = X.shape
    LR, LC = r + (2 if r % 4 else 0), c + (2 if c % 4 else 0)
    lolo = np.full((B, LR // 2, LC // 2), np.nan, np.float32)
    yh = np.full((B, LR // 4, LC // 4, 12), np.nan, np.float32)
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = q[:8]
    la, pla = _d(h0b); lb, plb = _d(h0a); ha, pha = _d(h1b); hb, phb = _d(h1a)
    rc = emu.emu_fwd2(len(la), _f(X), _f(lolo), _f(yh), B, r, c, pla, plb, pha, phb)
    assert rc == 0
    return lolo, yh.view(np.complex64)


def emu_inv1(emu, Z, Yh, g0o, g1o, gain):
    B, R, C = Z.shape
    X = np.full((B, R, C), np.nan, np.float32)
    g0, p0 = _d(g0o); g1, p1 = _d(g1o); gn, pg = _d(gain)
    yh = np.ascontiguousarray(Yh).view(np.float32)
    rc = emu.emu_inv1(len(g0), len(g1), _f(Z), _f(yh), _f(X), B, R, C, pg, p0, p1)
    assert rc == 0
    return X


def emu_inv2(emu, Z, Yh, q, gain, cropR, cropC, large=False):
    B, zr, zc = Z.shape
    out = np.full((B, 2 * zr - 2 * cropR, 2 * zc - 2 * cropC), np.nan, np.float32)
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = q[:8]
    la, pla = _d(g0b); lb, plb = _d(g0a); ha, pha = _d(g1b); hb, phb = _d(g1a)
    gn, pg = _d(gain)
    yh = np.ascontiguousarray(Yh).view(np.float32)
    fn = emu.emu_inv2_large if large else emu.emu_inv2          # default: the tile choice of the library (small here)
    rc = fn(len(la), _f(Z), _f(yh), _f(out), B, zr, zc, cropR, cropC, pg, pla, plb, pha, phb)
    assert rc == 0
    return out


BIORTS = ['near_sym_a', 'antonini', 'legall', 'near_sym_b']
QSHIFTS = ['qshift_a', 'qshift_b', 'qshift_c', 'qshift_d', 'qshift_32']


@pytest.mark.parametrize('bn', BIORTS)
# the fused kernels are only used for levels >= 40 samples wide (one-bounce reflection)
@pytest.mark.parametrize('shape', [(64, 128), (41, 47), (70, 40), (40, 44)])
def test_emu_level1_forward_inverse(emu, bn, shape):
    rs = np.random.RandomState(3)
    X = rs.standard_normal((2,) + shape).astype(np.float32)
    b = biort(bn)
    lolo, yh = emu_fwd1(emu, X, b[0], b[2])
    t = o.Transform2d(b, qshift('qshift_a'))
    for i in range(2):
        p = t.forward(X[i], nlevels=1)
        assert rel(lolo[i], p.lowpass) < TOL
        assert rel(yh[i], p.highpasses[0]) < TOL
    gain = np.array([1.0, 0.5, 0.0, 2.0, 1.5, 0.7])
    Z = emu_inv1(emu, lolo, yh, b[1], b[3], gain)
    for i in range(2):
        want = t.inverse(o.Pyramid(lolo[i], (yh[i],)), gain.reshape(6, 1))
        assert rel(Z[i], want) < TOL


def _flipped(qn):
    """qshift_a with the signs of g0b, g1b, h0b and h1b flipped: sum(ha hb) of every pair changes sign, so the level
    >= 2 tile programs take their run-time filter phases instead of the compile-time ones of the shipped sets"""
    q = [np.array(v, dtype=np.float64) for v in qshift(qn.split(':')[0])]
    for k in (1, 3, 5, 7):
        q[k] = -q[k]
    return tuple(q)


@pytest.mark.parametrize('qn', QSHIFTS + ['qshift_a:flipped'])
@pytest.mark.parametrize('shape', [(64, 64), (44, 52), (42, 74), (40, 40), (130, 66)])
def test_emu_level2_forward_inverse(emu, qn, shape):
    rs = np.random.RandomState(5)
    X = rs.standard_normal((2,) + shape).astype(np.float32)
    q = _flipped(qn) if ':' in qn else qshift(qn)
    lolo, yh = emu_fwd2(emu, X, q)
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = q[:8]
    rows = lambda fn, A, *h: fn(A.T, *h).T
    gain = np.array([1.0, 0.5, 0.0, 2.0, 1.5, 0.7])
    padR, padC = int(shape[0] % 4 != 0), int(shape[1] % 4 != 0)
    Zi = emu_inv2(emu, lolo, yh, q, gain, padR, padC)
    Zl = emu_inv2(emu, lolo, yh, q, gain, padR, padC, large=True)
    for i in range(2):
        L = X[i]
        if padR:
            L = np.concatenate((L[:1], L, L[-1:]), 0)
        if padC:
            L = np.concatenate((L[:, :1], L, L[:, -1:]), 1)
        Lo = o.coldfilt(L, h0b, h0a)
        Hi = o.coldfilt(L, h1b, h1a)
        ll = rows(o.coldfilt, Lo, h0b, h0a)
        want = np.zeros((ll.shape[0] >> 1, ll.shape[1] >> 1, 6), np.complex64)
        want[:, :, 0:6:5] = o.q2c(rows(o.coldfilt, Hi, h0b, h0a))
        want[:, :, 2:4:1] = o.q2c(rows(o.coldfilt, Lo, h1b, h1a))
        want[:, :, 1:5:3] = o.q2c(rows(o.coldfilt, Hi, h1b, h1a))
        assert rel(lolo[i], ll) < TOL
        assert rel(yh[i], want) < TOL
        # inverse of this level (transform2d.py:242-268)
        lh = o.c2q(want[:, :, [0, 5]], gain[[0, 5]])
        hl = o.c2q(want[:, :, [2, 3]], gain[[2, 3]])
        hh = o.c2q(want[:, :, [1, 4]], gain[[1, 4]])
        y1 = o.colifilt(ll, g0b, g0a) + o.colifilt(lh, g1b, g1a)
        y2 = o.colifilt(hl, g0b, g0a) + o.colifilt(hh, g1b, g1a)
        Z = rows(o.colifilt, y1, g0b, g0a) + rows(o.colifilt, y2, g1b, g1a)
        if padR:
            Z = Z[1:-1]
        if padC:
            Z = Z[:, 1:-1]
        assert Z.shape == Zi[i].shape
        assert rel(Zi[i], Z) < 4 * TOL and rel(Zl[i], Z) < 4 * TOL


# ------------------------------------------------------------------------------ band-pass sets
@pytest.mark.parametrize('shape', [(64, 128), (41, 47), (70, 40), (44, 52)])
def test_emu_bandpass_levels(emu, shape):
    """near_sym_b_bp / qshift_b_bp: the third filter of the diagonal subbands in all four tile
    programs, against the oracle's two-level transform of the same wavelets."""
    rs = np.random.RandomState(8)
    X = rs.standard_normal((2,) + shape).astype(np.float32)
    b, q = biort('near_sym_b_bp'), qshift('qshift_b_bp')
    t = o.Transform2d(b, q)
    B, r, c = X.shape
    R, C = r + (r & 1), c + (c & 1)
    lolo = np.full((B, R, C), np.nan, np.float32)
    yh0 = np.full((B, R // 2, C // 2, 12), np.nan, np.float32)
    d = lambda a: _d(a)
    (h0, p0), (h1, p1), (h2, p2) = d(b[0]), d(b[2]), d(b[4])
    assert emu.emu_fwd1_bp(len(h0), len(h1), len(h2), _f(X), _f(lolo), _f(yh0), B, r, c, p0, p1, p2) == 0
    LR, LC = R + (2 if R % 4 else 0), C + (2 if C % 4 else 0)
    lolo2 = np.full((B, LR // 2, LC // 2), np.nan, np.float32)
    yh1 = np.full((B, LR // 4, LC // 4, 12), np.nan, np.float32)
    taps = [d(q[1]), d(q[0]), d(q[5]), d(q[4]), d(q[9]), d(q[8])]       # (h0b,h0a) (h1b,h1a) (h2b,h2a)
    assert emu.emu_fwd2_bp(len(taps[0][0]), _f(lolo), _f(lolo2), _f(yh1), B, R, C, *[x[1] for x in taps]) == 0
    gain = np.array([[1.0, 0.8], [0.5, 1.2], [0.0, 1.0], [2.0, 0.3], [1.5, 1.0], [0.7, 0.9]])
    gtaps = [d(q[3]), d(q[2]), d(q[7]), d(q[6]), d(q[11]), d(q[10])]    # (g0b,g0a) (g1b,g1a) (g2b,g2a)
    padR, padC = int(R % 4 != 0), int(C % 4 != 0)
    z1 = np.full((B, R, C), np.nan, np.float32)
    g1, pg1 = _d(gain[:, 1])
    assert emu.emu_inv2_bp(len(gtaps[0][0]), _f(lolo2), _f(yh1), _f(z1), B, LR // 2, LC // 2, padR, padC, pg1,
                           *[x[1] for x in gtaps]) == 0
    z0 = np.full((B, R, C), np.nan, np.float32)
    (g0o, q0), (g1o, q1), (g2o, q2) = d(b[1]), d(b[3]), d(b[5])
    g0, pg0 = _d(gain[:, 0])
    assert emu.emu_inv1_bp(len(g0o), len(g1o), len(g2o), _f(z1), _f(yh0), _f(z0), B, R, C, pg0, q0, q1, q2) == 0
    for i in range(2):
        p = t.forward(X[i], nlevels=2, include_scale=True)
        assert rel(lolo[i], p.scales[0]) < TOL and rel(yh0[i].view(np.complex64), p.highpasses[0]) < TOL
        assert rel(lolo2[i], p.lowpass) < TOL and rel(yh1[i].view(np.complex64), p.highpasses[1]) < 2 * TOL
        want = t.inverse(o.Pyramid(lolo2[i], (yh0[i].view(np.complex64), yh1[i].view(np.complex64))), gain)
        assert want.shape == z0[i].shape and rel(z0[i], want) < 4 * TOL


# ------------------------------------------------------------------------------ 3-D
def emu_fwd3_l1(emu, X, h0o, h1o, chunk):
    n0, n1, n2 = X.shape
    LLL = np.full(X.shape, np.nan, np.float32)
    Yh = np.full((n0 // 2, n1 // 2, n2 // 2, 56), np.nan, np.float32)
    h0, p0 = _d(h0o)
    h1, p1 = _d(h1o)
    rc = emu.emu_fwd3_l1(len(h0), len(h1), _f(X), _f(LLL), _f(Yh), n0, n1, n2, chunk, p0, p1)
    assert rc == 0
    return LLL, Yh.view(np.complex64)


@pytest.mark.parametrize('shape,chunk', [((8, 8, 8), 8), ((12, 20, 70), 4), ((10, 34, 130), 6), ((16, 16, 64), 16), ((12, 32, 128), 6)])
@pytest.mark.parametrize('bname', ['near_sym_a', 'antonini', 'legall'])
def test_fwd3_level1_tiles(emu, shape, chunk, bname):
    X = np.random.RandomState(11).standard_normal(shape).astype(np.float32)
    b = biort(bname)
    LLL, Yh = emu_fwd3_l1(emu, X, b[0], b[2], chunk)
    want = o.Transform3d(b, qshift('qshift_a')).forward(X.astype(np.float64), nlevels=1)
    assert rel(LLL, want.lowpass) < TOL
    assert Yh.shape == want.highpasses[0].shape
    assert rel(Yh, want.highpasses[0]) < TOL


def _oracle_level2(X, q, ext):
    t = o.Transform3d(biort('near_sym_a'), q, ext_mode=ext)
    h0a, h0b, h1a, h1b = [np.asarray(v, dtype=np.float64) for v in (q[0], q[1], q[4], q[5])]
    return t._level2_xfm(X, h0a, h0b, h1a, h1b)


@pytest.mark.parametrize('shape,ext', [((8, 40, 44), 4), ((14, 42, 50), 4), ((20, 44, 60), 8), ((4, 48, 40), 8), ((12, 36, 38), 4)])
@pytest.mark.parametrize('qname', ['qshift_a', 'qshift_b', 'qshift_d'])
def test_fwd3_level2_tiles(emu, shape, ext, qname):
    """Level >= 2 (two passes) against the oracle's level 2 given the same input volume."""
    X = np.random.RandomState(13).standard_normal(shape).astype(np.float32)
    q = qshift(qname)
    h0a, h0b, h1a, h1b = q[0], q[1], q[4], q[5]
    mult, npad = (4, 1) if ext == 4 else (8, 2)
    pads = [npad if s % mult else 0 for s in shape]
    L = [s + 2 * p for s, p in zip(shape, pads)]
    O = [l // 2 for l in L]
    planes = np.full((4, shape[0], O[1], O[2]), np.nan, np.float32)
    LLL = np.full(O, np.nan, np.float32)
    Yh = np.full((O[0] // 2, O[1] // 2, O[2] // 2, 56), np.nan, np.float32)
    t = [_d(h) for h in (h0b, h0a, h1b, h1a)]
    rc = emu.emu_fwd3_l2(len(t[0][0]), _f(X), _f(planes), _f(LLL), _f(Yh), shape[0], shape[1], shape[2],
                         pads[0], pads[1], pads[2], t[0][1], t[1][1], t[2][1], t[3][1])
    assert rc == 0
    lo, hi = _oracle_level2(X.astype(np.float64), q, ext)
    assert LLL.shape == lo.shape and rel(LLL, lo) < TOL
    Yc = Yh.view(np.complex64)
    assert Yc.shape == hi.shape and rel(Yc, hi) < TOL


def _rand_level(shape, seed):
    """(lowpass volume, highpass records) of one 3-D level with the lowpass *shape*."""
    rs = np.random.RandomState(seed)
    Yl = rs.standard_normal(shape).astype(np.float32)
    hs = tuple(s // 2 for s in shape) + (28,)
    Yh = (rs.standard_normal(hs) + 1j * rs.standard_normal(hs)).astype(np.complex64)
    return Yl, Yh


@pytest.mark.parametrize('shape,chunk', [((12, 40, 44), 4), ((12, 42, 70), 3), ((14, 40, 130), 64), ((12, 18, 20), 2),
                                         ((12, 18, 262), 3), ((12, 16, 512), 6)])   # rows of more than 128 cells: k tiles with halo cells
@pytest.mark.parametrize('bname', ['near_sym_a', 'antonini', 'legall'])
def test_inv3_level1_tiles(emu, shape, chunk, bname):
    Yl, Yh = _rand_level(shape, 21)
    b = biort(bname)
    g0, p0 = _d(b[1])
    g1, p1 = _d(b[3])
    planes = np.full((2,) + shape, np.nan, np.float32)       # Q[a1]: axes 0 and 2 merged
    Z = np.full(shape, np.nan, np.float32)
    rc = emu.emu_inv3_l1(len(g0), len(g1), _f(Yl), _f(Yh), _f(planes), _f(Z), shape[0], shape[1], shape[2], chunk,
                         p0, p1)
    assert rc == 0
    want = o.Transform3d(b, qshift('qshift_a')).inverse(o.Pyramid(Yl.astype(np.float64), (Yh.astype(np.complex128),)))
    assert Z.shape == want.shape and rel(Z, want) < TOL


@pytest.mark.parametrize('shape,crops,chunk', [((12, 40, 44), (0, 0, 0), 4), ((12, 42, 70), (1, 1, 0), 3),
                                               ((14, 40, 64), (2, 0, 2), 64), ((16, 44, 40), (1, 2, 1), 5),
                                               ((12, 28, 30), (1, 0, 1), 2)])
@pytest.mark.parametrize('qname', ['qshift_a', 'qshift_b'])
def test_inv3_level2_tiles(emu, shape, crops, chunk, qname):
    Yl, Yh = _rand_level(shape, 22)
    q = qshift(qname)
    g0a, g0b, g1a, g1b = q[2], q[3], q[6], q[7]
    t = [_d(h) for h in (g0b, g0a, g1b, g1a)]
    S = 2 * shape[0] - 2 * crops[0]
    oshape = tuple(2 * s - 2 * c for s, c in zip(shape, crops))
    planes = np.full((4, S, shape[1], shape[2]), np.nan, np.float32)
    Z = np.full(oshape, np.nan, np.float32)
    rc = emu.emu_inv3_l2(len(t[0][0]), _f(Yl), _f(Yh), _f(planes), _f(Z), shape[0], shape[1], shape[2],
                         crops[0], crops[1], crops[2], chunk, t[0][1], t[1][1], t[2][1], t[3][1])
    assert rc == 0
    full = o.Transform3d._merge(Yl.astype(np.float64), Yh.astype(np.complex128), o.colifilt,
                                (np.asarray(g0b), np.asarray(g0a)), (np.asarray(g1b), np.asarray(g1a)))
    sl = tuple(slice(c, full.shape[a] - c) for a, c in enumerate(crops))
    want = full[sl]
    assert Z.shape == want.shape and rel(Z, want) < TOL


@pytest.mark.parametrize('B,R,nstrip,band', [(1, 4096, 18, 40), (1, 4096, 18, 148), (4, 4096, 18, 148), (64, 2048, 9, 256),
                                             (64, 1024, 5, 176), (3, 128, 2, 8), (1, 64, 1, 24), (2, 520, 14, 36), (5, 44, 13, 12)])
def test_march_job_order_is_a_bijection(emu, B, R, nstrip, band):
    """The workgroup -> (strip, band, image) map of the marching launches (dtm_job, march2d.hpp): every job exactly once,
    surplus workgroups leave, and the strips of one (image, band) sit on at most `ngrp` XCDs (workgroup w runs on XCD
    w % 8) in contiguous runs -- what lets neighbouring strips share their halo columns through one L2."""
    nband = -(-R // band)
    cap = 8 * (B * nband * 2 + 8) * (nstrip + 2)
    out = (ctypes.c_int * (3 * cap))()
    emu.emu_march_jobs.restype = ctypes.c_int
    grid = emu.emu_march_jobs(B, R, nstrip, band, out, cap)
    assert 0 < grid <= cap
    jobs = np.frombuffer(out, dtype=np.int32)[:3 * grid].reshape(grid, 3)
    live = jobs[jobs[:, 0] >= 0]
    assert len(live) == B * nband * nstrip
    keys = live[:, 2].astype(np.int64) * nband * nstrip + live[:, 1] * nstrip + live[:, 0]
    assert len(np.unique(keys)) == len(keys) and keys.min() == 0 and keys.max() == B * nband * nstrip - 1
    assert grid < 2 * len(live) + 8 * (nstrip + 1)          # the padding is bounded
    w = np.nonzero(jobs[:, 0] >= 0)[0]
    xcd = w % 8
    ngrp = 2 if nstrip >= 14 else 1
    for b in range(min(B, 2)):
        for bd in (0, nband - 1):
            sel = (live[:, 2] == b) & (live[:, 1] == bd)
            assert len(np.unique(xcd[sel])) <= ngrp


def test_march_level2_taps_reproduce_coldfilt(emu):
    """dtm_pack_qshift (march2d.hpp): with the taps laid out by window offset, A = sum_t ta[t] w[2t] and
    B = sum_t tb[t] w[2t + 1] over the window of 2M samples starting at 4i - M + 2 are rows 2i and 2i + 1 of
    coldfilt(X, ha, hb) (dtcwt/numpy/lowlevel.py:82-154; the order of the pair by the sign of sum(ha hb))."""
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = [np.asarray(v, dtype=np.float64).ravel() for v in qshift('qshift_a')]
    M = len(h0a)
    f = lambda v: np.ascontiguousarray(np.concatenate([v, np.zeros(40 - M)]), dtype=np.float32)
    la, lb, ha, hb = f(h0b), f(h0a), f(h1b), f(h1a)          # coldfilt(X, h0b, h0a), coldfilt(X, h1b, h1a)
    out = np.zeros(80, dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    emu.emu_march_pack_qshift.restype = ctypes.c_int
    assert emu.emu_march_pack_qshift(M, fp(la), fp(lb), fp(ha), fp(hb), fp(out)) == 20
    ta_lo, tb_lo, ta_hi, tb_hi = (out[20 * k:20 * k + M].astype(np.float64) for k in range(4))
    rs = np.random.RandomState(3)
    X = rs.standard_normal((48, 3))
    r = X.shape[0]
    ext = lambda u: X[np.where(u < 0, -1 - u, np.where(u >= r, 2 * r - 1 - u, u))]          # half-sample symmetric
    for (ta, tb, fa, fb) in ((ta_lo, tb_lo, h0b, h0a), (ta_hi, tb_hi, h1b, h1a)):
        want = o.coldfilt(X, fa, fb)
        a_first = np.dot(fa, fb) > 0
        for i in range(r // 4):
            w = ext(np.arange(4 * i - M + 2, 4 * i + M + 2))
            A = (ta[:, None] * w[0::2]).sum(0); Bv = (tb[:, None] * w[1::2]).sum(0)
            first, second = (A, Bv) if a_first else (Bv, A)
            assert np.abs(first - want[2 * i]).max() < 1e-6 and np.abs(second - want[2 * i + 1]).max() < 1e-6
