"""Multi-GPU plumbing of the ``hip`` backend: the path shards by image.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL on ROCm); a batch of
independent images is split contiguously across ranks and each rank transforms its own
images with its own plan -- no data-path collective, exactly the shape of the reference's
only parallel code, the MPI frame scatter of examples/register_video.py:125-156.  The one
collective is a broadcast of the packed filter-tap table from rank 0 at set-up (SURVEY.md
section 8(e)).  torch is imported lazily: nothing here is needed on a single GPU.
"""
import numpy as np

__all__ = ['shard_range', 'pack_taps', 'unpack_taps', 'broadcast_taps']


def shard_range(n_items, rank, world):
    """Contiguous [start, stop) of *n_items* owned by *rank*; sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError('bad rank/world %r/%r' % (rank, world))
    base, rem = divmod(int(n_items), world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_taps(biort, qshift):
    """(flat float64 vector, lengths) of all tap vectors, biort first."""
    vecs = [np.asarray(h, np.float64).reshape(-1) for h in tuple(biort) + tuple(qshift)]
    return np.concatenate(vecs), [len(v) for v in vecs]


def unpack_taps(flat, lengths, n_biort):
    parts, off = [], 0
    for n in lengths:
        parts.append(np.array(flat[off:off + n], dtype=np.float64))
        off += n
    return tuple(parts[:n_biort]), tuple(parts[n_biort:])


def broadcast_taps(biort, qshift, dist, device=None, src=0, group=None):
    """Broadcast rank *src*'s tap table to every rank of the default process group, or of *group*
    (``dist`` = ``torch.distributed``).  With ``device`` set the buffer lives on that GPU
    (RCCL over xGMI); with ``None`` on the host (gloo, used by the CPU tests).  Every rank
    must pass tap tuples of the same lengths (the wavelet *names* are part of the job
    description; the *values* come from rank *src*)."""
    import torch
    flat, lengths = pack_taps(biort, qshift)
    if dist.get_rank() != src:
        flat = np.zeros_like(flat)
    buf = torch.from_numpy(flat)
    if device is not None:
        buf = buf.to(device)
    dist.broadcast(buf, src=src, group=group)
    return unpack_taps(buf.cpu().numpy(), lengths, len(biort))
