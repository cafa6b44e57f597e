"""Batches of images over the GPUs of one node, from one process.

``MultiGPUTransform2d`` is the host side of ``dtcwt_hip_mgpu_*`` (include/dtcwt_hip.h): a batch
of B equally sized images is split contiguously over the devices (:func:`shard_range`), every
shard has its own context, stream, fused plan and host worker thread inside the library, and
the pyramids stay in the HBM of the device that computed them.  Nothing is exchanged between
shards -- the shape of the reference's only parallel code, the MPI scatter / transform / gather of
frame groups in examples/register_video.py:125-156.  With ``broadcast_taps=True`` the filter-tap
table travels to the other devices by one RCCL broadcast over xGMI at set-up and each shard's plan
is built from the copy that arrived on its device.

    m = MultiGPUTransform2d('near_sym_a', 'qshift_a', devices=range(8), batch=512, rows=2048,
                            cols=2048, nlevels=4)
    pyr = m.forward(X)            # X: host array [512, 2048, 2048]; pyramids stay on the devices
    Z = m.inverse(pyr)            # host array [512, 2048, 2048]

``lanes=K`` keeps K batches in flight per device -- the engine of the one-process-per-GPU path: K native objects on the
same devices (``dtcwt_hip_mgpu_create_lane``), lane k's shard contexts on share k of K of their device's compute units
where that measured faster (four lanes: images of 1024 x 1024 and more; two lanes: 2048 x 2048 and more), plain contexts with the concurrency hint
otherwise.  ``alloc()`` hands out buffer sets lane by lane; ``forward_into`` / ``inverse_into`` run on the lane their
buffers belong to, so consecutive batches overlap:

    m = MultiGPUTransform2d('near_sym_a', 'qshift_a', devices=range(8), batch=8, rows=4096, cols=4096, nlevels=4, lanes=4)
    sets = [m.alloc() for _ in range(8)]          # lanes 0, 1, 2, 3, 0, 1, 2, 3
    for k, frames in enumerate(groups_of_8_frames):
        s = sets[k % 8]; m.scatter(frames, s.X); m.forward_into(s); ...

The one-process-per-GPU form (``torch.distributed``, what ``bench.py`` runs under a launcher) is in
:mod:`dtcwt_amd.hip.sharding`.
"""
import ctypes

import numpy as np

from dtcwt_amd.coeffs import biort as _biort, qshift as _qshift
from dtcwt_amd.utils import flat_taps
from dtcwt_amd.hip import _lib
from dtcwt_amd.hip._lib import check
from dtcwt_amd.hip.sharding import shard_range

__all__ = ['MultiGPUTransform2d', 'ShardedBuffers']

_vp = ctypes.c_void_p
_pd = ctypes.POINTER(ctypes.c_double)
BCAST_TAPS = 1          # DTCWT_HIP_MGPU_BCAST_TAPS
PARTITION, NO_PARTITION = 2, 4      # DTCWT_HIP_MGPU_PARTITION / _NO_PARTITION


class _ShardCtx(object):
    """Borrowed view of the context the library created for a shard (never destroyed from here)."""

    def __init__(self, handle, device, lib):
        self._h = _vp(handle)
        self.device = device
        self._lib = lib

    @property
    def handle(self):
        return self._h


class ShardedBuffers(object):
    """Device buffers of one batch: per shard X, Yl, Yh[level], Z (``DeviceArray``; None for empty shards)."""

    def __init__(self, X, Yl, Yh, Z, Ys=None, lane=0):
        self.X, self.Yl, self.Yh, self.Z, self.Ys, self.lane = X, Yl, Yh, Z, Ys, lane


class MultiGPUTransform2d(object):
    def __init__(self, biort, qshift, devices, batch, rows, cols, nlevels, broadcast_taps=False, lanes=1, partition=None):
        try:
            biort = _biort(biort)
        except TypeError:
            pass
        try:
            qshift = _qshift(qshift)
        except TypeError:
            pass
        if len(biort) != 4 or len(qshift) != 8:
            raise ValueError('the multi-GPU plan takes 4-vector biort and 8-vector q-shift sets')
        lanes = int(lanes)
        if not 1 <= lanes <= 16:
            raise ValueError('lanes: 1 .. 16')
        L = _lib.lib()
        if _lib.device_count() < 1:
            raise _lib.NoHIPPresentError('no HIP device visible to libdtcwt_hip.so')
        devices = [int(d) for d in devices]
        self._keep = [flat_taps(h) for h in biort] + [flat_taps(h) for h in qshift]
        bp = (_pd * 4)(*[a.ctypes.data_as(_pd) for a in self._keep[:4]])
        bl = (ctypes.c_int * 4)(*[a.shape[0] for a in self._keep[:4]])
        qp = (_pd * 8)(*[a.ctypes.data_as(_pd) for a in self._keep[4:]])
        ql = (ctypes.c_int * 8)(*[a.shape[0] for a in self._keep[4:]])
        dv = (ctypes.c_int * len(devices))(*devices)
        self._lib, self._hs = L, []
        self.lanes = lanes
        pflag = 0 if partition is None else (PARTITION if partition else NO_PARTITION)
        for lane in range(lanes):
            h = _vp()
            # the one collective of the path, the RCCL broadcast of the tap table, happens once: lane 0 builds its plans
            # from the copies that arrived on the devices, the other lanes of the same process from the same host table
            flags = (BCAST_TAPS if (broadcast_taps and lane == 0) else 0) | pflag
            rc = L.dtcwt_hip_mgpu_create_lane(len(devices), dv, batch, rows, cols, nlevels, bp, bl, qp, ql, flags,
                                              lane, lanes, ctypes.byref(h))
            if rc == -3:
                raise NotImplementedError(L.dtcwt_hip_last_error().decode())
            check(rc)
            self._hs.append(h)
        h = self._h = self._hs[0]
        self.devices, self.batch, self.rows, self.cols, self.nlevels = devices, batch, rows, cols, nlevels
        self.ndev = len(devices)
        s = (ctypes.c_int * (4 + 4 * nlevels))()
        check(L.dtcwt_hip_mgpu_shapes(h, s))
        self.ext, self.low = (s[0], s[1]), (s[2], s[3])
        self.high = [(s[4 + 4 * l], s[5 + 4 * l]) for l in range(nlevels)]
        self.scale = [(s[6 + 4 * l], s[7 + 4 * l]) for l in range(nlevels)]
        self.shards = []
        for d in range(self.ndev):
            dev, st, cnt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(L.dtcwt_hip_mgpu_shard(h, d, ctypes.byref(dev), ctypes.byref(st), ctypes.byref(cnt)))
            assert (st.value, st.value + cnt.value) == shard_range(batch, d, self.ndev)
            self.shards.append((dev.value, st.value, cnt.value))
        self.lane_ctxs = [[_ShardCtx(L.dtcwt_hip_mgpu_ctx(hl, d), self.shards[d][0], L) for d in range(self.ndev)] for hl in self._hs]
        self.ctxs = self.lane_ctxs[0]
        self.shares = int(L.dtcwt_hip_mgpu_shares(h))      # > 1: every lane's contexts run on 1 / shares of their device
        self._next_lane = 0

    @property
    def taps_broadcast(self):
        return bool(self._lib.dtcwt_hip_mgpu_taps_broadcast(self._h))

    # ---- buffers ---------------------------------------------------------------------
    def _per_shard(self, shape_tail, dtype, lane=0):
        out = []
        for d, (_, _, cnt) in enumerate(self.shards):
            out.append(_lib.DeviceArray(self.lane_ctxs[lane][d], (cnt,) + tuple(shape_tail), dtype) if cnt else None)
        return out

    def alloc(self, with_input=True, include_scale=False, lane=None):
        """Device buffers of one batch on lane *lane* (default: the lanes in turn)."""
        if lane is None:
            lane = self._next_lane
            self._next_lane = (self._next_lane + 1) % self.lanes
        X = self._per_shard((self.rows, self.cols), np.float32, lane) if with_input else None
        Yl = self._per_shard(self.low, np.float32, lane)
        Yh = [[None] * self.nlevels for _ in range(self.ndev)]
        for l in range(self.nlevels):
            col = self._per_shard(self.high[l] + (6,), np.complex64, lane)
            for d in range(self.ndev):
                Yh[d][l] = col[d]
        Z = self._per_shard(self.ext, np.float32, lane)
        bufs = ShardedBuffers(X, Yl, Yh, Z, lane=lane)
        bufs.Ys = None
        if include_scale:       # the lowpass image of every level (transform2d.py:96-99, :160-163)
            bufs.Ys = [[None] * self.nlevels for _ in range(self.ndev)]
            for l in range(self.nlevels):
                col = self._per_shard(self.scale[l], np.float32, lane)
                for d in range(self.ndev):
                    bufs.Ys[d][l] = col[d]
        return bufs

    def _lane_of(self, dev):
        """The lane a list of per-shard device arrays was allocated on (its contexts)."""
        for a in dev:
            if a is not None:
                for k in range(self.lanes):
                    if any(a.ctx is c for c in self.lane_ctxs[k]):
                        return k
        return 0

    @staticmethod
    def _ptrs(arrs):
        return (_vp * len(arrs))(*[(a.ptr if a is not None else None) for a in arrs])

    def scatter(self, host, dev):
        """Host batch [batch, ...] -> the shards' buffers *dev* (list of DeviceArray per shard)."""
        host = np.ascontiguousarray(host)
        assert host.shape[0] == self.batch
        per = host.nbytes // self.batch
        check(self._lib.dtcwt_hip_mgpu_scatter(self._hs[self._lane_of(dev)], host.ctypes.data_as(_vp), per, self._ptrs(dev)))

    def gather(self, dev, shape_tail, dtype):
        out = np.empty((self.batch,) + tuple(shape_tail), dtype=dtype)
        per = out.nbytes // self.batch
        check(self._lib.dtcwt_hip_mgpu_gather(self._hs[self._lane_of(dev)], self._ptrs(dev), per, out.ctypes.data_as(_vp)))
        return out

    def pinned_empty(self, shape_tail, dtype):
        """A page-locked host array [batch, ...] for scatter_async / gather_async (from the pool of pinned buffers)."""
        from dtcwt_amd.hip._lib import host_pool
        return host_pool.empty((self.batch,) + tuple(shape_tail), dtype)

    def scatter_async(self, host, dev):
        """As :meth:`scatter`, enqueue only: *host* must be page-locked (:meth:`pinned_empty`) and stay untouched
        until :meth:`sync`; every shard's upload runs on its own stream ahead of the transforms issued after it."""
        assert host.flags.c_contiguous and host.shape[0] == self.batch
        per = host.nbytes // self.batch
        check(self._lib.dtcwt_hip_mgpu_scatter_async(self._hs[self._lane_of(dev)], host.ctypes.data_as(_vp), per, self._ptrs(dev)))

    def gather_async(self, dev, out):
        """As :meth:`gather` into the page-locked array *out*, enqueue only: the downloads run on the shards' copy
        streams behind the kernels issued so far; *out* is valid after :meth:`sync`."""
        assert out.flags.c_contiguous and out.shape[0] == self.batch
        per = out.nbytes // self.batch
        check(self._lib.dtcwt_hip_mgpu_gather_async(self._hs[self._lane_of(dev)], self._ptrs(dev), per, out.ctypes.data_as(_vp)))
        return out

    # ---- transforms ------------------------------------------------------------------
    def forward_into(self, bufs):
        flat = [bufs.Yh[d][l] for d in range(self.ndev) for l in range(self.nlevels)]
        Ys = getattr(bufs, 'Ys', None)
        if Ys is None:
            check(self._lib.dtcwt_hip_mgpu_forward2d(self._hs[bufs.lane], self._ptrs(bufs.X), self._ptrs(bufs.Yl), self._ptrs(flat)))
        else:
            flat_s = [Ys[d][l] for d in range(self.ndev) for l in range(self.nlevels)]
            check(self._lib.dtcwt_hip_mgpu_forward2d_scales(self._hs[bufs.lane], self._ptrs(bufs.X), self._ptrs(bufs.Yl),
                                                            self._ptrs(flat), self._ptrs(flat_s)))

    def inverse_into(self, bufs, gain_mask=None):
        flat = [bufs.Yh[d][l] for d in range(self.ndev) for l in range(self.nlevels)]
        gp = None
        if gain_mask is not None:
            g = np.ascontiguousarray(np.asarray(gain_mask, dtype=np.float64).reshape(6, self.nlevels))
            gp = g.ctypes.data_as(_pd)
        check(self._lib.dtcwt_hip_mgpu_inverse2d(self._hs[bufs.lane], self._ptrs(bufs.Yl), self._ptrs(flat), gp, self._ptrs(bufs.Z)))

    def sync(self):
        for h in self._hs:
            check(self._lib.dtcwt_hip_mgpu_sync(h))

    def forward(self, X, include_scale=False):
        """Host batch [batch, rows, cols] float32 -> :class:`ShardedBuffers` (pyramids resident per device);
        *include_scale*: also keep the lowpass image of every level (``bufs.Ys[shard][level]``)."""
        X = np.asarray(X)
        if X.shape != (self.batch, self.rows, self.cols):
            raise ValueError('expected a batch of shape %r' % ((self.batch, self.rows, self.cols),))
        bufs = self.alloc(include_scale=include_scale)
        self.scatter(X.astype(np.float32, copy=False), bufs.X)
        self.forward_into(bufs)
        return bufs

    def inverse(self, bufs, gain_mask=None):
        self.inverse_into(bufs, gain_mask)
        return self.gather(bufs.Z, self.ext, np.float32)

    def gather_pyramid(self, bufs):
        """(lowpass [B, ...], [highpasses[l] [B, hr, hc, 6]]) on the host."""
        low = self.gather(bufs.Yl, self.low, np.float32)
        high = [self.gather([bufs.Yh[d][l] for d in range(self.ndev)], self.high[l] + (6,), np.complex64)
                for l in range(self.nlevels)]
        return low, high

    def gather_scales(self, bufs):
        """[scales[l] [B, lo_r, lo_c]] on the host (forward(..., include_scale=True))."""
        return [self.gather([bufs.Ys[d][l] for d in range(self.ndev)], self.scale[l], np.float32)
                for l in range(self.nlevels)]

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                for lane in self.lane_ctxs:         # buffers that outlive the plan must not touch its contexts
                    for c in lane:
                        c._h = None
                for h in self._hs:
                    self._lib.dtcwt_hip_mgpu_destroy(h)
                self._h, self._hs = None, []
        except Exception:
            pass
