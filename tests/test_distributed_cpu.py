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
        # bench.py's arrangement: the taps travel on a group of their own (RCCL there, gloo here) that is torn
        # down right after the broadcast; barriers and reductions stay on the default group
        g = dist.new_group(backend='gloo')
        b2, q2 = broadcast_taps(b, qs, dist, device=None, src=0, group=g)
        assert dist.get_world_size(group=g) == world
        dist.destroy_process_group(g)
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
    """Levels 1 + 2 of a batch through the host emulator of the fused per-level tile programs (k_fwd1, k_fwd2:
    the kernels the plan launches)."""
    from tests.test_emu_tiles import emu_fwd1, emu_fwd2
    lolo1, yh0 = emu_fwd1(emu, X, b[0], b[2])
    lolo2, yh1 = emu_fwd2(emu, lolo1, q)
    return lolo2, yh0, yh1


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


def _bench_shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cfg = bench.CONFIGS['c5']
        plan, global_batch = bench.shard_plan(cfg, world)
        seed, images = plan[rank]
        head = np.random.RandomState(seed).standard_normal(8).astype(np.float32)     # start of this rank's image 0
        dt = torch.tensor([1e-3 * (1 + rank)], dtype=torch.float64)                  # rank r "took" (1 + r) ms
        every = [torch.zeros_like(dt) for _ in range(world)]
        dist.all_gather(every, dt)
        mx = dt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        got = [None] * world
        dist.all_gather_object(got, (seed, images, head.tolist()))
        px = float(images) * cfg['rows'] * cfg['cols']
        q.put((rank, got, global_batch, [float(t) for t in every], bench.aggregate_value(world, px, 10, float(mx))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_four_rank_gloo_bench_sharding_c5():
    """bench.py's N-GPU arithmetic for BASELINE config[4] on four gloo ranks: rank r transforms 64 images of
    seed 3 + 1000 r per step (weak scaling, 4 x 64 images in flight), the per-rank times are gathered for the
    min / max of the JSON line, and `value` is all ranks' pixels over the SLOWEST rank's time."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_bench_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, global_batch, every, value in res:
        assert [g[0] for g in got] == [3, 1003, 2003, 3003]
        assert [g[1] for g in got] == [64] * 4 and global_batch == 256
        heads = [tuple(g[2]) for g in got]
        assert len(set(heads)) == 4                      # every rank's images differ
        assert every == pytest.approx([1e-3, 2e-3, 3e-3, 4e-3])
        # 4 ranks x 64 x 2048^2 px x 10 steps in 4 ms (the slowest rank)
        assert value == pytest.approx(4 * 64 * 2048 * 2048 * 10 / 4e-3 / 1e6)
