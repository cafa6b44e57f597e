"""Experiment: four transforms in flight (hipGraph replay, so that the host is out of it), marching launches forced /
tile programs / the plan's choice, by image size (profiles/r04/hint_sizes.txt).

    python tools/kbench/hint_sizes.py [part]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import dtcwt_amd.hip
from dtcwt_amd.hip import Context, DeviceArray
# plain contexts with the concurrency hint (or, argv[1] = 'part', contexts on quarters of the CUs), graphs to take the host out of it
PART = len(sys.argv) > 1 and sys.argv[1] == 'part'
ctxs = [Context(0, partition=(p, 4)) if PART else Context(0) for p in range(4)]
t2s = [dtcwt_amd.hip.Transform2d(ctx=c) for c in ctxs]
rs = np.random.RandomState(5)
print('%s, hipGraph replay; us per image, four in flight: march / tiles / auto' % ('contexts on quarters' if PART else 'plain streams + hint 4'))
for n, m, nl in ((256, 256, 3), (512, 512, 3), (768, 768, 4), (1024, 1024, 4), (1080, 1920, 4), (1536, 1536, 4), (1792, 1792, 4), (2048, 2048, 4)):
    out = []
    for mode in ('1', '0', None):
        if mode is None: os.environ.pop('DTCWT_HIP_MARCH', None)
        else: os.environ['DTCWT_HIP_MARCH'] = mode
        plans = [t.plan(1, n, m, nl) for t in t2s]
        if not PART:
            for pl in plans: pl.set_concurrency(4)
        sets = []
        for k in range(8):
            c = ctxs[k % 4]; pl = plans[0]
            sets.append((c.to_device(rs.standard_normal((1, n, m)).astype(np.float32)), DeviceArray(c, (1,) + pl.low, np.float32),
                         [DeviceArray(c, (1,) + pl.high[l] + (6,), np.complex64) for l in range(nl)], DeviceArray(c, (1,) + pl.ext, np.float32)))
        graphs = [plans[k % 4].capture(*s[:3], s[3]) for k, s in enumerate(sets)]
        best = 1e9
        for rep in range(3):
            for k in range(40): graphs[k % 8].launch()
            ctxs[0].device_sync()
            t0 = time.perf_counter()
            for k in range(400): graphs[k % 8].launch()
            ctxs[0].device_sync()
            best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
        out.append(best)
        del graphs
    print('%-14s %8.1f %8.1f %8.1f %s' % ('%dx%d, %d' % (n, m, nl), out[0], out[1], out[2], plans[0].launches()))
