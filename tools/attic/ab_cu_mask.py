#!/usr/bin/env python
"""CU-masked streams under the bench protocol (experiment): does giving each of the S streams of
independent images its own part of the chip (hipExtStreamCreateWithCUMask) beat letting the dispatcher mix their
wavefronts on every CU?  4096 x 4096 float32, nlevels=4, forward + inverse per step, eight rotating buffer sets.

    python tools/ab_cu_mask.py [steps=200] [rounds=3] [variant list 1|2] [indices, e.g. 0,1]

Every CU-masked stream is a hardware queue of its own: with some twenty of them alive in one process the queue scheduler
time-slices them and every variant slows down (0.27-0.37 ms), so compare few variants per process.

The library takes the caller's stream (dtcwt_hip_ctx_create(device, stream)), so nothing in it changes: the masked streams
are made here.  Each variant is timed `rounds` times, the variants alternating, after one settle phase.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dtcwt_amd.hip                                     # noqa: E402
from dtcwt_amd.coeffs import biort, qshift               # noqa: E402
from dtcwt_amd.hip import Context, DeviceArray           # noqa: E402

NCU = 256
_hip = []


def hip_runtime():
    """The HIP runtime instance libdtcwt_hip.so runs on: looked up only AFTER the library has loaded it (dlopen by
    name before that brought a second instance into the process, which then sees no device)."""
    if not _hip:
        Context(0)
        _hip.append(ctypes.CDLL('libamdhip64.so.7'))
    return _hip[0]


def masked_stream(bits):
    words = [0] * (NCU // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    hip = hip_runtime()
    hip.hipSetDevice(0)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask -> %d' % rc)
    return s.value


def variants(which):
    allb = range(NCU)
    first = [
        ('4 streams, no mask', 4, None, None),
        ('4 streams, bits 64s..64s+63', 4, lambda s: [b for b in allb if b // 64 == s], None),
        ('4 streams, bits b % 4 == s', 4, lambda s: [b for b in allb if b % 4 == s], None),
        ('4 streams, bits b % 8 in (2s, 2s+1)', 4, lambda s: [b for b in allb if (b % 8) // 2 == s], None),
        ('2 streams, no mask', 2, None, None),
        ('2 streams, bits 128s..128s+127', 2, lambda s: [b for b in allb if b // 128 == s], None),
        ('2 streams, bits b % 2 == s', 2, lambda s: [b for b in allb if b % 2 == s], None),
        ('2 streams, bits b % 8 in (4s..4s+3)', 2, lambda s: [b for b in allb if (b % 8) // 4 == s], None),
    ]
    second = [
        ('4 streams, no mask', 4, None, None),
        ('4 streams, bits 64s..64s+63', 4, lambda s: [b for b in allb if b // 64 == s], None),
        ('4 x 64 bits, bands as for 2 in flight', 4, lambda s: [b for b in allb if b // 64 == s], 2),
        ('4 x 64 bits, bands as for 8 in flight', 4, lambda s: [b for b in allb if b // 64 == s], 8),
        ('8 streams, bits 32s..32s+31', 8, lambda s: [b for b in allb if b // 32 == s], None),
        ('8 streams, no mask', 8, None, None),
        ('3 streams, bits 85s..85s+84', 3, lambda s: [b for b in allb if b // 85 == s and b < 255], None),
        ('4 streams, 128 bits from 64s (overlapping)', 4, lambda s: [(64 * s + i) % NCU for i in range(128)], None),
        ('4 streams, bits (b // 8) % 4 == s', 4, lambda s: [b for b in allb if (b // 8) % 4 == s], None),
        ('4 streams, bits (b // 16) % 4 == s', 4, lambda s: [b for b in allb if (b // 16) % 4 == s], None),
    ]
    return second if which == 2 else first


class Rig(object):
    def __init__(self, name, nstreams, maskfn, hint=None):
        self.name = name
        R = C = 4096
        NL = 4
        nsets = 8 if nstreams != 3 else 9
        bt, qt = biort('near_sym_a'), qshift('qshift_a')
        self.ctxs = [Context(0, stream=masked_stream(maskfn(s))) if maskfn else Context(0) for s in range(nstreams)]
        self.t2s = [dtcwt_amd.hip.Transform2d(tuple(bt), tuple(qt), ctx=c) for c in self.ctxs]
        self.plans = [t.plan(1, R, C, NL) for t in self.t2s]
        for pl in self.plans:
            pl.set_concurrency(hint if hint else nstreams)
        rs = np.random.RandomState(5)
        self.sets = []
        for k in range(nsets):
            c = self.ctxs[k % nstreams]
            X = c.to_device(rs.standard_normal((1, R, C)).astype(np.float32))
            pl = self.plans[0]
            Yl = DeviceArray(c, (1,) + pl.low, np.float32)
            Yh = [DeviceArray(c, (1,) + pl.high[l] + (6,), np.complex64) for l in range(NL)]
            Z = DeviceArray(c, (1,) + pl.ext, np.float32)
            self.sets.append((X, Yl, Yh, Z))
        self.n = nstreams

    def run(self, steps):
        for k in range(steps):
            X, Yl, Yh, Z = self.sets[k % len(self.sets)]
            pl = self.plans[k % self.n]
            pl.forward_into(X, Yl, Yh)
            pl.inverse_into(Yl, Yh, None, Z)

    def timed(self, steps):
        self.run(16)
        self.ctxs[0].device_sync()
        t0 = time.perf_counter()
        self.run(steps)
        self.ctxs[0].device_sync()
        return (time.perf_counter() - t0) / steps * 1e3

    def check(self):
        X, Yl, Yh, Z = self.sets[0]
        self.ctxs[0].device_sync()
        return float(np.abs(Z.get() - X.get()).max())


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rigs = []
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    only = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else None
    for i, (name, n, fn, hint) in enumerate(variants(which)):
        if only is not None and i not in only:
            continue
        try:
            rigs.append(Rig(name, n, fn, hint))
        except Exception as exc:      # a mask the runtime refuses is a result too
            print('%-42s not created: %s' % (name, exc))
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        rigs[0].run(32)
    rigs[0].ctxs[0].device_sync()
    res = {r.name: [] for r in rigs}
    for _ in range(rounds):
        for r in rigs:
            res[r.name].append(r.timed(steps))
    if only is None:
        print('4096 x 4096 f32 nlevels=4 forward + inverse, %d steps per measurement, ms per step (%d rounds, variants alternating)' % (steps, rounds))
    for r in rigs:
        print('%-42s %s   recon err %.2e' % (r.name, '  '.join('%.4f' % v for v in res[r.name]), r.check()))


if __name__ == '__main__':
    main()
