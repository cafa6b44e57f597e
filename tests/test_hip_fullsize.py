"""Full-size parity of the BASELINE single-GPU configurations against the oracle, and the
single-process multi-GPU entry (dtcwt_hip_mgpu_*).

The reference's pattern is whole-array comparison of the accelerated backend with gold
(/root/reference tests/test_openclxfm2.py:25-90): here every level and every subband of the
4096^2 nlevels=4 pyramid (C2), the inverse with a random gain mask, images of one GPU's share
of the 512 x 2048^2 batch (C5) and the contiguous batch split of examples/register_video.py:125-156.
"""
import os

import numpy as np
import pytest

from oracle import dtcwt_oracle as o
from dtcwt_amd.coeffs import biort, qshift
from dtcwt_amd.hip import Transform2d, Context, default_context, DeviceArray
from tests._hip import assert_close, as_f64, XFM_TOL, INV_TOL

pytestmark = pytest.mark.gpu

B, Q = 'near_sym_a', 'qshift_a'


def _oracle():
    return o.Transform2d(biort(B), qshift(Q))


def test_c2_whole_pyramid_vs_oracle_4096():
    """BASELINE config[1]: 4096 x 4096 float32, nlevels=4 -- the WHOLE pyramid (every level, every
    subband, interior tiles and the XCD-order tile remap at this grid size included) against the
    oracle evaluated in float64 on the same samples, then the inverse with a gain mask."""
    rs = np.random.RandomState(4096)
    X = rs.standard_normal((4096, 4096)).astype(np.float32)
    t, to = Transform2d(B, Q), _oracle()
    p = t.forward(X, nlevels=4, include_scale=False)
    want = to.forward(as_f64(X), nlevels=4)
    assert_close(p.lowpass, want.lowpass, XFM_TOL, 'Yl')
    for l in range(4):
        assert p.highpasses[l].shape == want.highpasses[l].shape
        assert_close(p.highpasses[l], want.highpasses[l], XFM_TOL, 'Yh[%d]' % l)
        # per subband as well: a small subband must not hide behind a large one
        for d in range(6):
            assert_close(p.highpasses[l][:, :, d], want.highpasses[l][:, :, d], 2 * XFM_TOL, 'Yh[%d][..., %d]' % (l, d))
    gm = rs.uniform(0.3, 1.4, size=(6, 4)) * (rs.uniform(size=(6, 4)) > 0.2)
    z = t.inverse(p, gm)
    assert_close(z, to.inverse(want, gm), INV_TOL, 'inverse with gain mask')
    z1 = t.inverse(p)
    assert_close(z1, as_f64(X), INV_TOL, 'perfect reconstruction')


def test_c2_fused_levels_whole_pyramid_4096(monkeypatch):
    """The opt-in one-launch level-1+2 forward (DTCWT_HIP_FUSE12=1) at the headline size against the oracle."""
    monkeypatch.setenv('DTCWT_HIP_FUSE12', '1')
    rs = np.random.RandomState(4097)
    X = rs.standard_normal((4096, 4096)).astype(np.float32)
    t = Transform2d(B, Q)
    assert t.plan(1, 4096, 4096, 4).fused12
    p = t.forward(X, nlevels=4)
    want = _oracle().forward(as_f64(X), nlevels=4)
    assert_close(p.lowpass, want.lowpass, XFM_TOL, 'Yl')
    for l in range(4):
        assert_close(p.highpasses[l], want.highpasses[l], XFM_TOL, 'Yh[%d]' % l)


def test_c2_fused_levels_equal_one_launch_per_level(monkeypatch):
    """Levels 1+2 in one launch (DTCWT_HIP_FUSE12=1) give the same pyramid as one launch per level,
    include_scale (LoLo1 written out as well) included, on a size whose edge tiles hang over the image."""
    rs = np.random.RandomState(5)
    X = rs.standard_normal((1160, 1416)).astype(np.float32)
    monkeypatch.setenv('DTCWT_HIP_FUSE12', '1')
    a = Transform2d(B, Q)
    assert a.plan(1, 1160, 1416, 3).fused12
    pa = a.forward(X, nlevels=3, include_scale=True)
    monkeypatch.setenv('DTCWT_HIP_FUSE12', '0')
    b = Transform2d(B, Q)
    assert not b.plan(1, 1160, 1416, 3).fused12
    pb = b.forward(X, nlevels=3, include_scale=True)
    assert_close(pa.lowpass, pb.lowpass, 5e-7)
    for l in range(3):
        assert_close(pa.highpasses[l], pb.highpasses[l], 5e-7)
        assert_close(pa.scales[l], pb.scales[l], 5e-7)
    want = _oracle().forward(as_f64(X), nlevels=3, include_scale=True)
    assert_close(pa.scales[0], want.scales[0], XFM_TOL)
    assert_close(pa.highpasses[1], want.highpasses[1], XFM_TOL)


@pytest.mark.parametrize('shape', [(64, 64), (68, 132), (127, 256), (200, 64), (1023, 516)])
@pytest.mark.parametrize('bn', ['near_sym_a', 'antonini', 'legall'])
def test_fused_levels_small_and_odd(shape, bn, monkeypatch):
    """The one-launch level-1+2 path on small / odd (bottom row replicated) images whose tiles mostly
    hang over the edge; sizes whose extension is not a multiple of 4 keep one launch per level."""
    monkeypatch.setenv('DTCWT_HIP_FUSE12', '1')
    B = bn
    rs = np.random.RandomState(sum(shape))
    X = rs.standard_normal(shape).astype(np.float32)
    t = Transform2d(B, Q)
    R, C = shape[0] + (shape[0] & 1), shape[1] + (shape[1] & 1)
    assert t.plan(1, shape[0], shape[1], 2).fused12 == (R % 4 == 0 and C % 4 == 0)
    p = t.forward(X, nlevels=2)
    want = o.Transform2d(biort(B), qshift(Q)).forward(as_f64(X), nlevels=2)
    assert_close(p.lowpass, want.lowpass, XFM_TOL)
    for l in range(2):
        assert_close(p.highpasses[l], want.highpasses[l], XFM_TOL)


def test_c5_one_gpu_share_64x2048():
    """BASELINE config[4], one GPU's share: 64 images 2048 x 2048, nlevels=4, seed 3 + 1000*rank
    (SURVEY 8(d)).  Two images against the oracle; all 64 against the same image transformed alone."""
    rank = 0
    rs = np.random.RandomState(3 + 1000 * rank)
    Xb = rs.standard_normal((64, 2048, 2048)).astype(np.float32)
    t, to = Transform2d(B, Q), _oracle()
    ctx = default_context()
    pb = t.forward_channels(ctx.to_device(Xb), 'nhw', nlevels=4)
    assert pb.lowpass.shape == (64, 256, 256)
    for i in (0, 41):
        want = to.forward(as_f64(Xb[i]), nlevels=4)
        assert_close(pb.lowpass[i], want.lowpass, XFM_TOL, 'image %d Yl' % i)
        for l in range(4):
            assert_close(pb.highpasses[l][i], want.highpasses[l], XFM_TOL, 'image %d Yh[%d]' % (i, l))
    low = pb.lowpass
    high = [pb.highpasses[l] for l in (0, 3)]
    for i in range(64):
        single = t.forward(Xb[i], nlevels=4)
        assert np.array_equal(low[i], single.lowpass), i
        assert np.array_equal(high[0][i], single.highpasses[0]), i
        assert np.array_equal(high[1][i], single.highpasses[3]), i
    z = t.inverse_channels(pb, 'nhw')
    assert np.abs(z - Xb).max() < 2e-5 * np.abs(Xb).max()


@pytest.mark.parametrize('devices,bcast', [([0], False), ([0, 0], False), ([0, 0, 0], True)])
def test_mgpu_batch_split_matches_single_plan(devices, bcast):
    """dtcwt_hip_mgpu_*: contiguous batch split over shards (several shards on device 0 here; the
    same code drives 8 devices), one host thread + context + plan per shard; forward, inverse with a
    gain mask and gather equal the unsharded batch bit for bit.  With bcast the tap table travels
    through one RCCL broadcast first."""
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    rs = np.random.RandomState(len(devices))
    nb = 7
    X = rs.standard_normal((nb, 264, 328)).astype(np.float32)
    m = MultiGPUTransform2d(B, Q, devices=devices, batch=nb, rows=264, cols=328, nlevels=3, broadcast_taps=bcast)
    assert m.taps_broadcast == bcast
    assert [s[1:] for s in m.shards] == [(a, b - a) for a, b in
                                         [__import__('dtcwt_amd.hip.sharding', fromlist=['x']).shard_range(nb, d, len(devices))
                                          for d in range(len(devices))]]
    bufs = m.forward(X)
    low, high = m.gather_pyramid(bufs)
    t = Transform2d(B, Q)
    ref = t.forward_channels(X, 'nhw', nlevels=3)
    assert np.array_equal(low, ref.lowpass)
    for l in range(3):
        assert np.array_equal(high[l], ref.highpasses[l])
    want = _oracle().forward(as_f64(X[nb - 1]), nlevels=3)
    assert_close(high[2][nb - 1], want.highpasses[2], XFM_TOL)
    gm = rs.uniform(0.5, 1.5, size=(6, 3))
    z = m.inverse(bufs, gm)
    zr = t.inverse_channels(ref, 'nhw', gain_mask=gm)
    assert np.array_equal(z, zr)
    m.sync()


def test_mgpu_more_shards_than_images():
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    X = np.random.RandomState(9).standard_normal((2, 128, 128)).astype(np.float32)
    m = MultiGPUTransform2d(B, Q, devices=[0, 0, 0], batch=2, rows=128, cols=128, nlevels=2)
    assert [s[2] for s in m.shards] == [1, 1, 0]
    z = m.inverse(m.forward(X))
    assert np.abs(z - X).max() < 1e-5


def test_bench_self_spawns_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N ranks itself; with fewer devices than N it
    refuses loudly instead of silently running one rank (round-1 behaviour)."""
    import subprocess
    import sys
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from dtcwt_amd.hip import device_count
    ndev = device_count()
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(ndev + 1), '--steps', '2'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and 'visible' in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                        '--settle-ms', '0', '--no-cpu-baseline', '--rows', '512', '--cols', '512'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['config']['buffer_sets'] == 4 and line['recon_max_abs_err'] < 1e-4
