"""The N > 1 path on CPU: two gloo processes exercise what bench.py does at N GPUs --
tap broadcast from rank 0, batch sharding by image, max-over-ranks timing reduction -- and
the scatter / transform / gather of a sharded batch: every rank transforms ITS images with the
tile programs of the fused kernels stepped on the host (tests/emu, the same code the GPU runs)
using the taps it RECEIVED, rank 0 gathers the pyramids and compares them with the unsharded
batch (shape of the reference's examples/register_video.py:125-156)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_batch():
    from dtcwt_amd.hip.sharding import shard_range
    for n in (0, 1, 7, 8, 64, 512, 513):
        for world in (1, 2, 3, 8):
            got = [shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from dtcwt_amd.coeffs import biort, qshift
    from dtcwt_amd.hip.sharding import broadcast_taps, shard_range
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        b, qs = biort('near_sym_a'), qshift('qshift_a')
        if rank != 0:      # only rank 0 holds the real values
            b = tuple(np.full_like(h, np.nan) for h in b)
            qs = tuple(np.full_like(h, np.nan) for h in qs)
        b2, q2 = broadcast_taps(b, qs, dist, device=None, src=0)
        ok = all(np.array_equal(x.reshape(-1), np.asarray(y).reshape(-1)) for x, y in
                 zip(b2 + q2, biort('near_sym_a') + qshift('qshift_a')))
        lo, hi = shard_range(9, rank, world)
        t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        q.put((rank, ok, (lo, hi), float(t.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_tap_broadcast_and_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [(0, 5), (5, 9)]
    assert all(abs(r[3] - 0.2) < 1e-12 for r in res)


def _emu_levels12(emu, X, b, q):
    """Levels 1 + 2 of a batch through the host emulator of the fused tile program."""
    import ctypes
    B, r, c = X.shape
    yh0 = np.full((B, r // 2, c // 2, 12), np.nan, np.float32)
    lolo2 = np.full((B, r // 2, c // 2), np.nan, np.float32)
    yh1 = np.full((B, r // 4, c // 4, 12), np.nan, np.float32)
    d = lambda a: np.ascontiguousarray(np.asarray(a, np.float64).reshape(-1))
    f = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h0a, h0b, g0a, g0b, h1a, h1b, g1a, g1b = q[:8]
    taps = [d(b[0]), d(b[2]), d(h0b), d(h0a), d(h1b), d(h1a)]
    rc = emu.emu_fwd12(len(taps[0]), len(taps[1]), len(taps[2]), f(X), None, f(yh0), f(lolo2), f(yh1), B, r, c,
                       *[f(t) for t in taps])
    assert rc == 0
    return lolo2, yh0.view(np.complex64), yh1.view(np.complex64)


def _shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes
    import torch
    import torch.distributed as dist
    from dtcwt_amd.coeffs import biort, qshift
    from dtcwt_amd.hip.sharding import broadcast_taps, shard_range
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        emu = ctypes.CDLL(os.path.join(ROOT, 'tests', 'emu', 'libdtcwt_emu.so'))
        b, qs = biort('near_sym_a'), qshift('qshift_a')
        if rank != 0:      # only rank 0 holds the real values: a rank that skipped the broadcast computes NaNs
            b = tuple(np.full_like(h, np.nan) for h in b)
            qs = tuple(np.full_like(h, np.nan) for h in qs)
        b, qs = broadcast_taps(b, qs, dist, device=None, src=0)
        nb = 5
        batch = np.random.RandomState(77).standard_normal((nb, 64, 128)).astype(np.float32)    # same on every rank
        lo, hi = shard_range(nb, rank, world)
        mine = _emu_levels12(emu, np.ascontiguousarray(batch[lo:hi]), b, qs)
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, hi, mine))
        ok = None
        if rank == 0:
            whole = _emu_levels12(emu, batch, b, qs)
            parts = sorted(gathered, key=lambda g: g[0])
            assert [(g[0], g[1]) for g in parts] == [shard_range(nb, r, world) for r in range(world)]
            ok = all(np.array_equal(np.concatenate([g[2][k] for g in parts], axis=0), whole[k]) for k in range(3))
            ok = ok and not any(np.isnan(w).any() for w in whole)
        dist.barrier()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_transform_gather():
    from tests.test_emu_tiles import _build
    _build()
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] is True
