"""Per-kernel medians and fabric traffic of ONE profiled bench command, as JSON.

    python tools/roofline_from_trace.py <outdir of tools/profile_round.sh> [--write profiles/traffic.json]

Reads the rocprofv3 outputs that tools/profile_round.sh leaves under <outdir>:
  trace/      --kernel-trace --stats        -> duration of every dispatch (the driver's command: four streams,
              so kernels of two images overlap in time)
  trace1/     the same with --streams 1     -> durations of kernels that run alone (what bench.py's hipEvent
              pairs measure)
  pmc_fetch/  --kernel-trace --pmc FETCH_SIZE
  pmc_write/  --kernel-trace --pmc WRITE_SIZE
Dispatches are grouped by (kernel, grid size); the untimed settle phase of bench.py issues the same
kernels, so every statistic is taken over the LAST `--last` dispatches of a group (default 60: the
one-stream phases at the end of a `--steps 20` run -- the resident / one-stream re-runs and the steps
with per-kernel event pairs -- in which kernels do not overlap) -- the clock ramp at the start of the
process and the multi-stream phases, where kernels of several images share the chip, are excluded.  Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE
counts 128-byte read requests as 64 B on gfx950 (MI355X_MICROARCH.md, HBM section).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

import numpy as np


def short_name(k):
    m = re.search(r'(k_\w+)', k)
    return m.group(1) if m else k[:40]


def groups_from_trace(path, last):
    by = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
        for r in rows:
            g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
            by[(short_name(r['Kernel_Name']), g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out = {}
    for key, v in by.items():
        v = np.asarray(v[-last:])
        out[key] = dict(n=int(v.size), median_us=float(np.median(v)), min_us=float(v.min()), mean_us=float(v.mean()))
    return out


def counters(path, last):
    by = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)):
        for r in csv.DictReader(open(f)):
            g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
            by[(short_name(r['Kernel_Name']), g)][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: float(np.mean(v[-last:])) for c, v in d.items()} for k, d in by.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('outdir')
    ap.add_argument('--last', type=int, default=60)
    ap.add_argument('--write', default=None, help='also write the bench.py side file (profiles/traffic.json)')
    ap.add_argument('--source', default=None, help='what to name as the source in the side file')
    a = ap.parse_args()
    tr = groups_from_trace(os.path.join(a.outdir, 'trace'), a.last)
    tr1 = groups_from_trace(os.path.join(a.outdir, 'trace1'), a.last) if os.path.isdir(os.path.join(a.outdir, 'trace1')) else {}
    fe = counters(os.path.join(a.outdir, 'pmc_fetch'), a.last)
    wr = counters(os.path.join(a.outdir, 'pmc_write'), a.last)
    # levels of a kernel family: largest grid = finest level
    fam = defaultdict(list)
    for (name, g) in tr:
        if name.startswith('k_'):
            fam[name].append(g)
    table = {}
    for name, grids in fam.items():
        for lvl, g in enumerate(sorted(set(grids), reverse=True)):
            key = (name, g)
            row = dict(tr[key], grid=g)
            if key in tr1:
                row.update(one_stream_median_us=tr1[key]['median_us'], one_stream_min_us=tr1[key]['min_us'])
            f = fe.get(key, {}).get('FETCH_SIZE')
            w = wr.get(key, {}).get('WRITE_SIZE')
            if f is not None and w is not None:
                row.update(fetch_bytes=2 * f * 1024, write_bytes=w * 1024, traffic_bytes=(2 * f + w) * 1024)
            table['%s#%d' % (name, lvl)] = row
    print(json.dumps(table, indent=1, sort_keys=True))
    if a.write:
        px = 4096 * 4096
        side = {'_method': 'tools/profile_round.sh: rocprofv3 --kernel-trace --stats, and --pmc FETCH_SIZE / --pmc WRITE_SIZE in '
                           'separate passes, over the driver\'s command `python bench.py --gpus 1 --steps 20 --warmup 5` (two '
                           'streams); statistics over the last %d dispatches of each kernel = the one-stream phases at the end of the run '
                           '(settle phase and overlapped multi-stream steps excluded); bytes per '
                           'launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md). '
                           'Counts fabric requests, Infinity-Cache hits included.' % a.last,
                '_algorithmic_bytes_per_launch': 20 * px, 'source': a.source or a.outdir, 'rocprof_median_us': {},
                'rocprof_median_us_one_stream': {}}
        for fam_name, short in (('k_fwd1#0', 'k_fwd1'), ('k_inv1#0', 'k_inv1'), ('k_fwd12m#0', 'k_fwd12m'), ('k_inv21m#0', 'k_inv21m')):
            if fam_name in table:
                side['rocprof_median_us'][short] = round(table[fam_name]['median_us'], 2)
                if 'one_stream_median_us' in table[fam_name]:
                    side['rocprof_median_us_one_stream'][short] = round(table[fam_name]['one_stream_median_us'], 2)
                if 'traffic_bytes' in table[fam_name]:
                    side[short] = int(table[fam_name]['traffic_bytes'])
        json.dump(side, open(a.write, 'w'), indent=1)


if __name__ == '__main__':
    main()
