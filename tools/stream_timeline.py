#!/usr/bin/env python
"""When does each of the four image streams finish its share of a K-step timed region?  (experiment: how much of the gap
between 20-step and 200-step runs of bench.py is drain imbalance)  4096 x 4096 f32 nlevels=4, partitioned contexts.

    python tools/stream_timeline.py [steps=20] [reps=6] [partition 1|0] [mode]
mode 1: the steps are issued to the streams in REVERSE order (stream 3 first); mode 2: stream s runs on share 3 - s
mode 3: the host issues the forwards of four consecutive steps, then their inverses
(round 6: is the stream that finishes first the one that starts first, or the one on share 0?)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dtcwt_amd.hip                                     # noqa: E402
from dtcwt_amd.hip import Context, DeviceArray           # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    part = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    S, R, NL = 4, 4096, 4
    ctxs = [Context(0, partition=((S - 1 - s) if mode == 2 else s, S)) if part else Context(0) for s in range(S)]
    t2s = [dtcwt_amd.hip.Transform2d(ctx=c) for c in ctxs]
    plans = [t.plan(1, R, R, NL) for t in t2s]
    if not part:
        for pl in plans:
            pl.set_concurrency(S)
    rs = np.random.RandomState(5)
    sets = []
    for k in range(8):
        c = ctxs[k % S]
        pl = plans[0]
        sets.append((c.to_device(rs.standard_normal((1, R, R)).astype(np.float32)), DeviceArray(c, (1,) + pl.low, np.float32),
                     [DeviceArray(c, (1,) + pl.high[l] + (6,), np.complex64) for l in range(NL)], DeviceArray(c, (1,) + pl.ext, np.float32)))

    def step(k):
        if mode == 1:
            k = (k // S) * S + (S - 1 - k % S)
        X, Yl, Yh, Z = sets[k % 8]
        pl = plans[k % S]
        pl.forward_into(X, Yl, Yh)
        pl.inverse_into(Yl, Yh, None, Z)

    t_end = time.perf_counter() + 0.4
    k = 0
    while time.perf_counter() < t_end:
        step(k); k += 1
    ctxs[0].device_sync()
    ev0 = [c.event() for c in ctxs]
    ev1 = [c.event() for c in ctxs]
    print('mode %d;' % mode, end=' ')
    print('%d steps over %d streams (%s); per stream: ms from its first launch to its last kernel; wall = host clock' % (steps, S, 'CU partition' if part else 'plain streams'))
    for rep in range(reps):
        for i in range(5):
            step(i)
        ctxs[0].device_sync()
        t0 = time.perf_counter()
        for c, e in zip(ctxs, ev0):
            e.record()
        if mode == 3:       # the forwards of S consecutive steps first, then their inverses: every stream has work after S calls
            for i0 in range(0, steps, S):
                for i in range(i0, min(i0 + S, steps)):
                    X, Yl, Yh, Z = sets[i % 8]
                    plans[i % S].forward_into(X, Yl, Yh)
                for i in range(i0, min(i0 + S, steps)):
                    X, Yl, Yh, Z = sets[i % 8]
                    plans[i % S].inverse_into(Yl, Yh, None, Z)
        else:
            for i in range(steps):
                step(i)
        t_issue = time.perf_counter() - t0
        for c, e in zip(ctxs, ev1):
            e.record()
        ctxs[0].device_sync()
        wall = time.perf_counter() - t0
        per = [a.elapsed_ms(b) for a, b in zip(ev0, ev1)]
        print('  wall %.3f ms (%.4f per step), host done issuing after %.3f ms; streams %s; ideal if balanced %.4f per step' %
              (wall * 1e3, wall * 1e3 / steps, t_issue * 1e3, ' '.join('%.3f' % p for p in per), sum(per) / S / steps))


if __name__ == '__main__':
    main()
