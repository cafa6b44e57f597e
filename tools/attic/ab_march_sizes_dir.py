#!/usr/bin/env python
"""One transform at a time, by image size and DIRECTION: the marching launches against the per-level tile programs
(DTCWT_HIP_MARCH=1 / =0), forward alone and inverse alone (the inverse's levels 2 + 1 run as a marching pair up to 4096^2).
us per call on one stream, device-resident buffers.

    python tools/ab_march_sizes_dir.py [reps=300]
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ctx = Context(0)
    t = dtcwt_amd.hip.Transform2d(ctx=ctx)
    rs = np.random.RandomState(1)
    print('%-22s %8s %8s %6s | %8s %8s %6s' % ('batch x rows x cols, nl', 'fwd m', 'fwd t', 'm/t', 'inv m', 'inv t', 'm/t'))
    cases = [(1, 512, 512, 3), (1, 768, 768, 4), (1, 1024, 1024, 4), (1, 1280, 1280, 4), (1, 1536, 1536, 4), (1, 1792, 1792, 4), (1, 2048, 2048, 4),
             (1, 1080, 1920, 4), (1, 720, 1280, 4), (2, 1024, 1024, 4), (4, 512, 512, 3), (4, 1024, 1024, 4), (1, 3072, 3072, 4)]
    for B, n, m_, nl in cases:
        pl = t.plan(B, n, m_, nl)
        X = ctx.to_device(rs.standard_normal((B, n, m_)).astype(np.float32))
        Yl = DeviceArray(ctx, (B,) + pl.low, np.float32)
        Yh = [DeviceArray(ctx, (B,) + pl.high[l] + (6,), np.complex64) for l in range(nl)]
        Z = DeviceArray(ctx, (B,) + pl.ext, np.float32)
        pl.forward_into(X, Yl, Yh)
        res = {}
        for rnd in range(3):
            for mode in ('1', '0'):
                os.environ['DTCWT_HIP_MARCH'] = mode
                for name, fn in (('f', lambda: pl.forward_into(X, Yl, Yh)), ('i', lambda: pl.inverse_into(Yl, Yh, None, Z))):
                    for _ in range(20):
                        fn()
                    ctx.sync()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    ctx.sync()
                    res.setdefault((name, mode), []).append((time.perf_counter() - t0) / reps * 1e6)
        os.environ.pop('DTCWT_HIP_MARCH')
        fm, ft, im, it = (min(res[k]) for k in (('f', '1'), ('f', '0'), ('i', '1'), ('i', '0')))
        print('%-22s %8.1f %8.1f %6.2f | %8.1f %8.1f %6.2f   %s %.1f Mpx' % ('%d x %d x %d, %d' % (B, n, m_, nl), fm, ft, fm / ft, im, it, im / it, pl.launches(), B * n * m_ / 1e6))


if __name__ == '__main__':
    main()
