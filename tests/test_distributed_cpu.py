"""The N > 1 path on CPU: two gloo processes exercise what bench.py does at N GPUs --
tap broadcast from rank 0, batch sharding by image, max-over-ranks timing reduction.
(No GPU compute here; the per-rank transform is the single-GPU path the GPU suite covers.)"""
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
