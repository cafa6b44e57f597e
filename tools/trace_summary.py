"""Per-dispatch kernel durations of a rocprofv3 kernel_trace.csv, in dispatch order, grouped by
consecutive repeats: `python tools/trace_summary.py trace.csv [every]`."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    m = re.search(r'(k_\w+)(<[^>]*>)?', n)
    out.append(((m.group(1) + (m.group(2) or '')) if m else n[:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('LDS_Block_Size', '')))
for name, us, grid, lds in out:
    print('%-34s %9.1f us  grid %s lds %s' % (name, us, grid, lds))
