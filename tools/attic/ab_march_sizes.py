#!/usr/bin/env python
"""One transform at a time, by image size: the marching launches (levels 1 + 2 in one launch) against the per-level tile
programs (DTCWT_HIP_MARCH=1 / =0, read per call), and what the plan picks itself (last column).  us per forward + inverse on one stream, device-resident buffers.

    python tools/ab_march_sizes.py [reps=300] [cases, e.g. 1x512x512x3,4x1024x1024x4]
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
    print('%-22s %10s %10s %10s' % ('batch x rows x cols, nl', 'march', 'tiles', 'march/tiles'))
    cases = [(1, 128, 128, 3), (1, 256, 256, 3), (1, 512, 512, 3), (1, 512, 512, 4), (1, 768, 768, 4), (1, 1024, 1024, 4), (1, 1536, 1536, 4),
             (1, 1792, 1792, 4), (1, 2048, 2048, 4), (1, 4096, 4096, 4), (4, 256, 256, 3), (4, 512, 512, 3), (16, 256, 256, 3), (16, 512, 512, 4),
             (8, 1024, 1024, 4), (1, 1024, 2048, 4), (1, 2048, 1024, 4), (1, 1080, 1920, 4), (2, 1536, 1536, 4), (2, 1024, 1024, 4),
             (4, 1024, 1024, 4), (32, 512, 512, 4), (64, 512, 512, 4), (256, 256, 256, 3), (1, 1024, 4096, 4), (1, 4096, 1024, 4),
             (3, 720, 1280, 4), (1, 1440, 2560, 4)]
    if len(sys.argv) > 2:
        cases = [tuple(int(v) for v in c.split('x')) for c in sys.argv[2].split(',')]
    for B, n, m_, nl in cases:
        pl = t.plan(B, n, m_, nl)
        X = ctx.to_device(rs.standard_normal((B, n, m_)).astype(np.float32))
        Yl = DeviceArray(ctx, (B,) + pl.low, np.float32)
        Yh = [DeviceArray(ctx, (B,) + pl.high[l] + (6,), np.complex64) for l in range(nl)]
        Z = DeviceArray(ctx, (B,) + pl.ext, np.float32)
        res = {}
        for rnd in range(3):
            for mode in ('1', '0'):
                os.environ['DTCWT_HIP_MARCH'] = mode
                for _ in range(20):
                    pl.forward_into(X, Yl, Yh); pl.inverse_into(Yl, Yh, None, Z)
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    pl.forward_into(X, Yl, Yh); pl.inverse_into(Yl, Yh, None, Z)
                ctx.sync()
                res.setdefault(mode, []).append((time.perf_counter() - t0) / reps * 1e6)
        os.environ.pop('DTCWT_HIP_MARCH')        # what the plan chooses when left alone
        m, tl = min(res['1']), min(res['0'])
        print('%-22s %10.1f %10.1f %10.2f   %s  %.1f Mpx' % ('%d x %d x %d, %d' % (B, n, m_, nl), m, tl, m / tl, pl.launches(), B * n * m_ / 1e6))


if __name__ == '__main__':
    main()
