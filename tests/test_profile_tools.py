"""tools/roofline_from_trace.py turns the rocprofv3 CSVs of tools/profile_round.sh into profiles/rNN/roofline.json and
profiles/traffic.json (what bench.py quotes as `traffic` / `rocprof_kernel_ms`): checked here on a synthetic trace."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)


def test_roofline_from_trace_medians_and_traffic(tmp_path):
    out = tmp_path / 'prof'
    k1 = 'void (anonymous namespace)::k_inv1<dt2d::Inv1RCfg<16, 120, 8, 7, 5, 0> >(dt2d::Inv1Params)'
    k2 = 'void (anonymous namespace)::k_inv2s<dt2d::Inv2RCfg<16, 56, 2, 10, false>, true>(dt2d::Inv2Params)'
    trace, t = [], 0
    for i in range(100):                        # 40 slow "settle" dispatches first, then 60 at the sustained rate
        for name, grid, us in ((k1, 2293760, 90.0 if i < 40 else 60.0 + (i % 3)), (k2, 1212416, 33.0), (k2, 311296, 12.0)):
            trace.append({'Kernel_Name': name, 'Grid_Size': grid, 'Start_Timestamp': t, 'End_Timestamp': t + int(us * 1000)})
            t += int(us * 1000) + 1500
    _write(str(out / 'trace' / 'x' / 'bench_kernel_trace.csv'), trace)
    _write(str(out / 'trace1' / 'x' / 'bench_kernel_trace.csv'), [dict(r) for r in trace])
    _write(str(out / 'pmc_fetch' / 'x' / 'p_counter_collection.csv'),
           [{'Kernel_Name': k1, 'Grid_Size': 2293760, 'Counter_Name': 'FETCH_SIZE', 'Counter_Value': 132900.0} for _ in range(70)])
    _write(str(out / 'pmc_write' / 'x' / 'p_counter_collection.csv'),
           [{'Kernel_Name': k1, 'Grid_Size': 2293760, 'Counter_Name': 'WRITE_SIZE', 'Counter_Value': 68900.0} for _ in range(70)])
    side = tmp_path / 'traffic.json'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'roofline_from_trace.py'), str(out), '--write', str(side)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    table = json.loads(r.stdout)
    assert set(table) == {'k_inv1#0', 'k_inv2s#0', 'k_inv2s#1'}             # levels of a family by grid size, finest first
    assert table['k_inv1#0']['n'] == 60 and 60.0 <= table['k_inv1#0']['median_us'] <= 62.0     # settle phase excluded
    assert table['k_inv2s#0']['grid'] == 1212416 and table['k_inv2s#1']['median_us'] == 12.0
    # bytes per launch = (2 FETCH + WRITE) * 1024: the gfx950 FETCH_SIZE correction
    assert table['k_inv1#0']['traffic_bytes'] == (2 * 132900.0 + 68900.0) * 1024
    s = json.load(open(side))
    assert s['k_inv1'] == int((2 * 132900.0 + 68900.0) * 1024)
    assert 60.0 <= s['rocprof_median_us']['k_inv1'] <= 62.0 and 'k_inv1' in s['rocprof_median_us_one_stream']
