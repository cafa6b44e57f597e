"""profiles/rNN/roofline.json + profiles/traffic.json from the per-config outputs of tools/profile_round.sh (written to <src>/c2) and
tools/profile_configs.sh (<src>/c3, c5, c4, c4q).

    python tools/merge_roofline.py gpurun_out/r06prof profiles/r06

Per config (c2, c3, c5, c4) and kernel: rocprofv3 median durations (alone = one transform in flight on the whole device;
in flight = the config's default streams), fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) KiB, the algorithmic
bytes of the launch and the two ratios that matter: traffic / algorithmic, algorithmic bytes / alone duration / 8 TB/s."""
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
PX = {'c2': 4096 * 4096, 'c3': 64 * 1024 * 1024, 'c5': 64 * 2048 * 2048, 'c4': 256 ** 3, 'c4q': 256 ** 3}
ALG = {  # algorithmic bytes per unit of the launches DESIGN section 3 prices
    'k_fwd12m#0': 20.0, 'k_inv21m#0': 20.0, 'k_inv21p#0': 20.0, 'k_fwd1#0': 20.0, 'k_inv1#0': 20.0,
    'k_fwd3m_l1#0': 36.0, 'k_fwd3_l1#0': 36.0,
}
out = {}
side = json.load(open('profiles/traffic.json')) if os.path.exists('profiles/traffic.json') else {}
for cfg in ('c2', 'c3', 'c5', 'c4', 'c4q'):
    f = os.path.join(src, cfg, 'roofline.json')
    if not os.path.exists(f):
        continue
    t = json.load(open(f))
    if cfg != 'c2':
        # the default run of a batch config may launch other grids (contexts on shares of the CUs) than the one-at-a-time
        # run the counters were collected on: key everything by the ALONE run (trace1 + counters), and take the in-flight
        # median of a kernel family from the default run's largest grid
        import importlib.util
        spec = importlib.util.spec_from_file_location('rft', os.path.join(os.path.dirname(__file__), 'roofline_from_trace.py'))
        rft = importlib.util.module_from_spec(spec); spec.loader.exec_module(rft)
        d = os.path.join(src, cfg)
        tr1, tr = rft.groups_from_trace(os.path.join(d, 'trace1'), 60), rft.groups_from_trace(os.path.join(d, 'trace'), 60)
        fe, wr = rft.counters(os.path.join(d, 'pmc_fetch'), 60), rft.counters(os.path.join(d, 'pmc_write'), 60)
        fam = {}
        for (name, g) in tr1:
            if name.startswith('k_'):
                fam.setdefault(name, []).append(g)
        t = {}
        for name, grids in fam.items():
            infl = sorted([g for (n2, g) in tr if n2 == name], reverse=True)
            for lvl, g in enumerate(sorted(set(grids), reverse=True)):
                row = dict(tr1[(name, g)], grid=g)
                row['one_stream_median_us'] = row['median_us']
                if lvl < len(infl):
                    row['median_us'] = tr[(name, infl[lvl])]['median_us']; row['in_flight_grid'] = infl[lvl]
                fv, wv = fe.get((name, g), {}).get('FETCH_SIZE'), wr.get((name, g), {}).get('WRITE_SIZE')
                if fv is not None and wv is not None:
                    row.update(fetch_bytes=2 * fv * 1024, write_bytes=wv * 1024, traffic_bytes=(2 * fv + wv) * 1024)
                t['%s#%d' % (name, lvl)] = row
    # keep, per kernel family, the entries of the largest grids only (levels), and add the ratios
    for k, v in t.items():
        fam = k.split('#')[0]
        alg = None
        for key, bpp in ALG.items():
            if key.split('#')[0] == fam and k.endswith('#0'):
                alg = bpp * PX[cfg]
        if alg:
            v['algorithmic_bytes'] = alg
            if 'traffic_bytes' in v:
                v['traffic_over_algorithmic'] = round(v['traffic_bytes'] / alg, 4)
            us = v.get('one_stream_median_us', v['median_us'])
            v['alone_frac_of_8TBs'] = round(alg / (us * 1e-6) / 8e12, 4)
            v['alone_frac_of_copy_ceiling'] = round(alg / (us * 1e-6) / 6.29e12, 4)
    out[cfg] = t
    sec = {'source': os.path.join(dst, 'roofline.json').replace(os.sep, '/') + ' [%s]' % cfg, 'dominant': 'k_inv21', 'rocprof_median_us': {}, 'rocprof_median_us_one_stream': {}}
    for k, v in t.items():
        if not k.endswith('#0'):
            continue
        fam = k.split('#')[0]
        sec['rocprof_median_us'][fam] = round(v['median_us'], 2)
        if 'one_stream_median_us' in v:
            sec['rocprof_median_us_one_stream'][fam] = round(v['one_stream_median_us'], 2)
        if 'traffic_bytes' in v:
            sec[fam] = int(v['traffic_bytes'])
    if cfg == 'c2':
        keep = {k: side[k] for k in ('_method',) if k in side}
        side = dict(keep, **{k: v for k, v in side.items() if k in ('c3', 'c5', 'c4', 'c4_qbgn')})
        side.update(sec)
        side['_algorithmic_bytes_per_launch'] = 20 * PX['c2']
    else:
        if cfg == 'c4q':
            # the level-1 forward of the long filters is two launches: the axis-0 pair filter + k_fwd3l_slices (bench.py prices them together)
            pair = [t.get('k_g2_fwd_p1#0'), t.get('k_fwd3l_slices#0')]
            if all(pair):
                if all('traffic_bytes' in v for v in pair):
                    sec['level1_forward'] = int(sum(v['traffic_bytes'] for v in pair))
                sec['rocprof_median_us_one_stream']['level1_forward'] = round(sum(v.get('one_stream_median_us', v['median_us']) for v in pair), 2)
            side['c4_qbgn'] = sec
            continue
        if cfg == 'c4' and 'k_fwd3m_l1' in sec:
            sec['k_fwd3_l1'] = sec['k_fwd3m_l1']          # the name bench.py's c4 roofline object was written for
            sec['rocprof_median_us_one_stream']['k_fwd3_l1'] = sec['rocprof_median_us_one_stream'].get('k_fwd3m_l1', sec['rocprof_median_us'].get('k_fwd3m_l1'))
        side[cfg] = sec
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, 'roofline.json'), 'w'), indent=1, sort_keys=True)
side['_method'] = ('tools/profile_round.sh + tools/profile_configs.sh + tools/merge_roofline.py: rocprofv3 --kernel-trace --stats and separate --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                   'passes per config; bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md); '
                   'c2: the driver\'s command, statistics over the last 60 dispatches (one-stream phases); c3 / c5 / c4: counters and the alone medians '
                   'from --streams 1 --cu-partition off runs.  Counts fabric requests, Infinity-Cache hits included.')
json.dump(side, open('profiles/traffic.json', 'w'), indent=1)
for cfg, t in out.items():
    for k, v in sorted(t.items()):
        if 'algorithmic_bytes' in v:
            print(cfg, k, 'alone %.1f us' % v.get('one_stream_median_us', v['median_us']), 'in flight %.1f us' % v['median_us'],
                  'traffic/alg', v.get('traffic_over_algorithmic'), 'frac', v['alone_frac_of_8TBs'])
