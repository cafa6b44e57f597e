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
    assert np.abs(z - Xb).max() < 1e-6 * np.abs(Xb).max()


def _image_of(a, i):
    """Image *i* of a batched device array as a device view (no copy)."""
    per = int(np.prod(a.shape[1:])) * a.dtype.itemsize
    return DeviceArray(a.ctx, (1,) + tuple(a.shape[1:]), a.dtype, ptr=a.ptr + i * per, owner=a)


def _device_memory_gb():
    import ctypes
    from dtcwt_amd.hip import _lib
    cus, mem = ctypes.c_int(0), ctypes.c_size_t(0)
    name = ctypes.create_string_buffer(256)
    _lib.load_library().dtcwt_hip_device_info(0, name, ctypes.byref(cus), ctypes.byref(mem))
    return mem.value / 2 ** 30


@pytest.mark.parametrize('nb', [172, 512])
def test_more_than_2_31_elements_on_one_gpu(nb):
    """nb = 512: the WHOLE C5 batch (BASELINE configs[4], 512 x 2048^2, what `bench.py --config c5full` times) on one
    GPU, ~65 GB of buffers; skipped on a device with less than 128 GB.
    172 x 2048^2 float32, nlevels=4 on ONE GPU: Yh[0] holds 172 x 1024^2 x 12 = 2.16e9 floats, beyond the 32-bit
    record addressing of the fast paths -- the transform must take the 64-bit branches (fwd1s_rows_flush's general
    index algebra, the 64-bit decode of the generic kernels) and still be right.  The reference has no size limit
    (dtcwt/numpy/transform2d.py:40-188 takes any array).  First and last image against the oracle, a middle one
    bit for bit against the same image transformed alone, then the inverse of the whole batch."""
    R, C, nl = 2048, 2048, 4
    if nb > 172 and _device_memory_gb() < 128:
        pytest.skip('needs ~65 GB of device memory')
    ctx = default_context()
    rs = np.random.RandomState(77)
    base = rs.standard_normal((8, R, C)).astype(np.float32)
    image = lambda i: (base[i % 8] * np.float32(1.0 + 0.01 * i)).astype(np.float32)     # every image different
    X = DeviceArray(ctx, (nb, R, C), np.float32)
    for i in range(nb):
        _image_of(X, i).set(image(i)[None])
    t, to = Transform2d(B, Q), _oracle()
    pb = t.forward_channels(X, 'nhw', nlevels=nl)
    yl, yh = pb.hip_lowpass, pb.hip_highpasses
    assert 2 * yh[0].size > 2 ** 31 and yh[0].shape == (nb, R // 2, C // 2, 6)      # floats (complex64 pairs)
    for i in (0, nb - 1):
        want = to.forward(as_f64(image(i)), nlevels=nl)
        assert_close(_image_of(yl, i).get()[0], want.lowpass, XFM_TOL, 'image %d Yl' % i)
        for l in range(nl):
            assert_close(_image_of(yh[l], i).get()[0], want.highpasses[l], XFM_TOL, 'image %d Yh[%d]' % (i, l))
    mid = nb // 2 + 11
    single = t.forward(image(mid), nlevels=nl)
    assert np.array_equal(_image_of(yl, mid).get()[0], single.lowpass)
    for l in range(nl):
        assert np.array_equal(_image_of(yh[l], mid).get()[0], single.highpasses[l]), l
    Z = t.inverse_channels(pb, 'nhw', device_output=True)
    assert Z.shape == (nb, R, C)
    for i in (0, mid, nb - 1):
        z = _image_of(Z, i).get()[0]
        assert np.abs(z - image(i)).max() < 1e-6 * np.abs(image(i)).max(), i
    # ... and against the inverse of that image alone (a single image takes the small coarse-level tiles, the
    # batch the large ones: the same sums formed by different kernel instantiations, equal to the last bit or two)
    alone = t.inverse(single)
    assert np.abs(_image_of(Z, mid).get()[0] - alone).max() <= 1e-6 * np.abs(alone).max()
    del X, pb, Z
    ctx.trim()


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


@pytest.mark.parametrize('lanes,partition,shape', [(4, None, (512, 464)), (4, True, (512, 464)), (2, None, (2048, 2048)), (3, None, (256, 328))])
def test_mgpu_lanes_in_flight(lanes, partition, shape):
    """dtcwt_hip_mgpu_create_lane / MultiGPUTransform2d(lanes=K): K batches in flight per device, lane k's shard contexts on
    share k of K of their device where the library's rule (or the caller) says so -- the engine of the one-process-per-GPU
    path.  Different batches on different lanes at the same time; every result equals the unsharded batch of the same
    program bit for bit."""
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    rs = np.random.RandomState(lanes)
    nb = 3
    m = MultiGPUTransform2d(B, Q, devices=[0, 0], batch=nb, rows=shape[0], cols=shape[1], nlevels=3, lanes=lanes, partition=partition)
    px = shape[0] * shape[1]
    want_shares = lanes if (partition or (partition is None and ((lanes == 4 and px >= 1024 * 1024) or (lanes == 2 and px >= 2048 * 2048)))) else 1
    assert m.shares == want_shares and m.lanes == lanes
    sets = [m.alloc() for _ in range(2 * lanes)]
    assert [s.lane for s in sets] == [k % lanes for k in range(2 * lanes)]
    Xs = [rs.standard_normal((nb,) + shape).astype(np.float32) for _ in sets]
    for s, X in zip(sets, Xs):            # everything enqueued before anything is read back
        m.scatter(X, s.X)
        m.forward_into(s)
        m.inverse_into(s)
    m.sync()
    t = Transform2d(B, Q, program='march')
    tt = Transform2d(B, Q, program='tiles')
    for s, X in zip(sets, Xs):
        low, high = m.gather_pyramid(s)
        # which program a lane's plan picked depends on its share of the device and the hint: compare with that program
        ref_m, ref_t = t.forward_channels(X, 'nhw', nlevels=3), tt.forward_channels(X, 'nhw', nlevels=3)
        ref = ref_m if np.array_equal(low, ref_m.lowpass) else ref_t
        assert np.array_equal(low, ref.lowpass)
        for l in range(3):
            assert np.array_equal(high[l], ref.highpasses[l]), (s.lane, l)
        z = m.gather(s.Z, m.ext, np.float32)
        assert np.abs(z - X).max() < 4e-6 * np.abs(X).max()


def test_mgpu_async_scatter_gather():
    """dtcwt_hip_mgpu_scatter_async / gather_async: page-locked host arrays, every shard's upload on its own stream and
    its download on its copy stream, complete at sync() -- the same numbers as the blocking pageable copies."""
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    rs = np.random.RandomState(23)
    nb = 6
    X = rs.standard_normal((nb, 512, 512)).astype(np.float32)
    m = MultiGPUTransform2d(B, Q, devices=[0, 0, 0], batch=nb, rows=512, cols=512, nlevels=3)
    bufs = m.alloc()
    Xp = m.pinned_empty((512, 512), np.float32)
    Xp[...] = X
    m.scatter_async(Xp, bufs.X)
    m.forward_into(bufs)
    m.inverse_into(bufs)
    Zp = m.gather_async(bufs.Z, m.pinned_empty((512, 512), np.float32))
    Yp = m.gather_async([bufs.Yh[d][0] for d in range(m.ndev)], m.pinned_empty((256, 256, 6), np.complex64))
    m.sync()
    ref = Transform2d(B, Q)
    p = ref.forward_channels(X, 'nhw', nlevels=3)
    assert np.array_equal(Yp, p.highpasses[0])
    assert np.array_equal(Zp, ref.inverse_channels(p, 'nhw'))
    assert_close(Zp, X, INV_TOL, 'reconstruction')


def test_mgpu_include_scale():
    """dtcwt_hip_mgpu_forward2d_scales: the per-level lowpass images (include_scale, transform2d.py:96-99, :160-163)
    of a sharded batch equal those of the unsharded batch bit for bit and the oracle's within tolerance."""
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    X = np.random.RandomState(21).standard_normal((5, 200, 264)).astype(np.float32)
    m = MultiGPUTransform2d(B, Q, devices=[0, 0], batch=5, rows=200, cols=264, nlevels=3)
    bufs = m.forward(X, include_scale=True)
    scales = m.gather_scales(bufs)
    low, _ = m.gather_pyramid(bufs)
    ref = Transform2d(B, Q).forward_channels(X, 'nhw', nlevels=3, include_scale=True)
    assert np.array_equal(low, ref.lowpass)
    for l in range(3):
        assert np.array_equal(scales[l], ref.scales[l]), l
    want = _oracle().forward(as_f64(X[4]), nlevels=3, include_scale=True)
    for l in range(3):
        assert_close(scales[l][4], want.scales[l], XFM_TOL)
    m.sync()


def test_mgpu_create_validates_tap_lengths():
    """dtcwt_hip_mgpu_create copies the tap tables before any plan sees them: lengths are checked first (a negative or
    oversized length from a C caller was undefined behaviour)."""
    import ctypes
    from dtcwt_amd.hip import _lib
    from dtcwt_amd.utils import flat_taps
    L = _lib.lib()
    pd = ctypes.POINTER(ctypes.c_double)
    taps = [flat_taps(h) for h in biort(B)] + [flat_taps(h) for h in qshift(Q)]
    bp = (pd * 4)(*[a.ctypes.data_as(pd) for a in taps[:4]])
    qp = (pd * 8)(*[a.ctypes.data_as(pd) for a in taps[4:]])
    dv = (ctypes.c_int * 1)(0)
    for bad in (-1, 0, 41):
        bl = (ctypes.c_int * 4)(bad, *[a.shape[0] for a in taps[1:4]])
        ql = (ctypes.c_int * 8)(*[a.shape[0] for a in taps[4:]])
        h = ctypes.c_void_p()
        rc = L.dtcwt_hip_mgpu_create(1, dv, 2, 64, 64, 2, bp, bl, qp, ql, 0, ctypes.byref(h))
        assert rc == -1 and b'out of range' in L.dtcwt_hip_last_error()


def test_mgpu_more_shards_than_images():
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    X = np.random.RandomState(9).standard_normal((2, 128, 128)).astype(np.float32)
    m = MultiGPUTransform2d(B, Q, devices=[0, 0, 0], batch=2, rows=128, cols=128, nlevels=2)
    assert [s[2] for s in m.shards] == [1, 1, 0]
    z = m.inverse(m.forward(X))
    assert np.abs(z - X).max() < 2e-6


def test_bench_default_line_carries_the_streaming_probe():
    """The default configuration's line runs tools/kbench/step_probe (built by __graft_entry__.build()) in the same call: a
    trivial float4 streaming program with the bytes and launches of a step, in the three protocols, and the transform's
    figures over it."""
    import subprocess
    import sys
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, 'tools', 'kbench', 'step_probe')):
        pytest.skip('tools/kbench/step_probe is not built')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '1',
                        '--settle-ms', '0', '--no-cpu-baseline', '--no-other-configs'], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    sp = line['streaming_probe']
    assert sp and 'error' not in sp, sp
    for k in ('one_stream', 'four_plain_streams', 'four_streams_on_quarters'):
        assert 0.05 < sp[k]['ms_per_step_200'] < 1.0 and 0.05 < sp[k]['ms_per_step_20'] < 1.0
        assert 0.2 < sp['probe_frac_of_8TBs'][k] < 1.0          # a streaming program cannot beat the spec peak
    assert sp['transform_over_probe']['one_stream'] > 0.8         # the transform does not beat a program without arithmetic by much
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--settle-ms', '0',
                        '--no-cpu-baseline', '--no-other-configs', '--no-probe'], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'streaming_probe' not in json.loads(r.stdout.strip().splitlines()[-1])


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
                        '--settle-ms', '0', '--no-cpu-baseline', '--no-other-configs', '--rows', '2048', '--cols', '2048'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['config']['buffer_sets'] == 8 and line['config']['streams'] == 4
    assert line['config']['cu_partition'].startswith('4 contexts')          # each stream on a quarter of the CUs
    assert line['roofline']['in_flight']['streams'] == 4 and line['roofline']['in_flight']['kernel_ms'] > 0
    assert line['recon_max_abs_err'] < 1e-4 and line['roofline']['launches']['fwd_levels_1_2_one_launch']


def test_bench_config_c4_line():
    """`bench.py --config c4` (3-D forward + inverse of one volume, here 64^3) prints the one JSON line with the roofline
    object of k_fwd3_l1 and a reconstruction that matches the oracle's."""
    import subprocess
    import sys
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', 'c4', '--rows', '64', '--steps', '3',
                        '--warmup', '1', '--settle-ms', '0', '--sets', '2', '--streams', '2'],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['unit'] == 'Mvoxels/s' and line['n_gpus'] == 1 and line['config']['buffer_sets'] == 2
    assert line['config']['streams'] == 2 and line['ms_per_step_one_stream'] > 0
    assert line['roofline']['kernel'].startswith('k_fwd3m_l1') and 0 < line['roofline']['frac'] < 1
    assert line['recon_max_abs_err'] < 1e-4 and line['gpu_vs_cpu_recon_max_abs_diff'] < 1e-4
    assert line['cpu_baseline']['kind'] == 'port'
    # "qbgn-style" (near_sym_b / qshift_b: other_configs.c4_qbgn of the default run): level 1 as two launches (fused3d_long.hpp: axis 0, then k_fwd3l_slices)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', 'c4', '--rows', '64', '--steps', '3',
                        '--warmup', '1', '--settle-ms', '0', '--sets', '2', '--streams', '2', '--biort', 'near_sym_b',
                        '--qshift', 'qshift_b', '--no-cpu-baseline'], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert 'near_sym_b/qshift_b' in line['config']['workload'] and 'k_fwd3l_slices' in line['roofline']['kernel']
    assert line['recon_max_abs_err'] < 1e-4 and line['roofline']['traffic'] is None
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', 'c4', '--gpus', '2', '--steps', '1'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and 'does not shard' in (r.stderr + r.stdout)


def test_scale_run_script_produces_one_line_per_mode(tmp_path):
    """tools/scale_run.sh is what the first run on an 8-GPU node will be started with: here with the devices this box
    has (normally one), a few steps per line -- every (config, N, mode) it runs must leave one well-formed bench line
    and the efficiency table at the end must print."""
    import json
    import os
    import subprocess
    from dtcwt_amd.hip import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'scale')
    env = dict(os.environ, SCALE_RUN_FAST='1')
    r = subprocess.run(['bash', os.path.join(root, 'tools', 'scale_run.sh'), out, '10'], capture_output=True, text=True,
                       timeout=800, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(l) for l in open(os.path.join(out, 'scale.jsonl')) if l.strip().startswith('{')]
    ndev = _lib.device_count()
    want = [(c, n, m) for c in ('c2', 'c5') for n in (1, 2, 4, 8) if n <= ndev for m in ('ranks', 'mgpu')]
    assert len(rows) == len(want), (len(rows), want, r.stderr[-2000:])
    for row, (c, n, m) in zip(rows, want):
        assert row['n_gpus'] == n and row['value'] > 0 and row['ms_per_step'] > 0 and row['unit'] == 'Mpixels/s'
        if m == 'ranks':
            assert row['rank_ms_per_step']['max'] >= row['rank_ms_per_step']['min'] > 0
        assert ('mgpu' in row['launch']) == (m == 'mgpu')
        assert ('4096' in row['metric']) == (c == 'c2')
    assert r.stdout.count('efficiency') == len(want)


@pytest.mark.parametrize('partition', [False, True])
def test_mgpu_lanes_beside_a_thread_on_the_null_stream(partition):
    """Multi-GPU readiness on one device: MultiGPUTransform2d(lanes=4) keeps four batches in flight while ANOTHER
    THREAD enqueues work on the legacy NULL stream of the same device, as a framework's default stream would on an 8-GPU node.
    Plain lanes (non-blocking streams) must overlap with it -- asserted by timing: the lanes' steps take no longer than 1.35 x their
    time alone although the NULL stream is busy throughout.  Lanes on shares of the compute units own BLOCKING streams
    (hipExtStreamCreateWithCUMask takes no flags; include/dtcwt_hip.h: dtcwt_hip_ctx_create_partition): every NULL-stream operation
    is a barrier across them, the documented caveat -- there only the results are asserted and the slowdown is printed."""
    import ctypes
    import threading
    import time
    from dtcwt_amd.hip.multigpu import MultiGPUTransform2d
    default_context()
    path = next((ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64.so' in ln), None)
    if path is None:
        pytest.skip('no libamdhip64 mapped: cannot enqueue NULL-stream work')
    hip = ctypes.CDLL(path)
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    rs = np.random.RandomState(9)
    R = C = 2048
    m = MultiGPUTransform2d('near_sym_a', 'qshift_a', devices=[0], batch=1, rows=R, cols=C, nlevels=4, lanes=4, partition=partition)
    assert m.shares == (4 if partition else 1)
    sets = [m.alloc() for _ in range(8)]
    X = [rs.standard_normal((1, R, C)).astype(np.float32) for _ in range(8)]
    for s, x in zip(sets, X):
        m.scatter(x, s.X)
    m.sync()

    def steps(n):
        t0 = time.perf_counter()
        for k in range(n):
            m.forward_into(sets[k % 8]); m.inverse_into(sets[k % 8])
        m.sync()
        return time.perf_counter() - t0
    steps(16)
    alone = min(steps(80) for _ in range(3))
    d = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(1 << 16)) == 0
    stop, count = threading.Event(), [0]

    def null_stream_worker():           # small operations: they cost the memory system nothing, only their ORDER matters
        while not stop.is_set():
            for _ in range(16):
                hip.hipMemsetAsync(d, count[0] & 255, ctypes.c_size_t(1 << 16), None)
                count[0] += 1
            hip.hipStreamSynchronize(None)
    th = threading.Thread(target=null_stream_worker)
    th.start()
    try:
        time.sleep(0.02)
        n0 = count[0]
        beside = min(steps(80) for _ in range(3))
        issued = count[0] - n0
    finally:
        stop.set(); th.join()
        hip.hipDeviceSynchronize(); hip.hipFree(d)
    print('lanes=4 partition=%s: 80 steps alone %.3f ms, beside %d NULL-stream operations %.3f ms (x %.2f)'
          % (partition, alone * 1e3, issued, beside * 1e3, beside / alone))
    assert issued > 0
    for s, x in zip(sets, X):           # right either way
        z = m.gather(s.Z, m.ext, np.float32)
        assert np.abs(z[:, :R, :C] - x).max() < 4e-6 * np.abs(x).max()
    if not partition:
        assert beside < 1.35 * alone, (alone, beside)
