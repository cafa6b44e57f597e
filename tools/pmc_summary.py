"""Summarise rocprofv3 --pmc CSVs: per kernel (largest grid of each name), mean counter value
per dispatch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values
grid = {}
for f in sorted(glob.glob(os.path.join(out, '*', '**', '*counter_collection.csv'), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get('Kernel_Name', '')
            m = re.search(r'(k_\w+)<.*?<([\d, ]+)>(.*?)>\(', k)
            if not m:
                m2 = re.search(r'(k_\w+)<([^>]*)>', k)
            short = (m.group(1) + '<' + m.group(2).replace(' ', '') + '>' + m.group(3).replace(' ', '')) if m else (m2.group(0).replace(' ', '') if m2 else k[:40])
            g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
            key = (short, g)
            rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
# keep, for each kernel name, the largest grid only (the 4096^2 level) and the next ones
names = sorted(rows, key=lambda k: (k[0], -k[1]))
for key in names:
    print('%s grid=%d' % key)
    for c in sorted(rows[key]):
        v = rows[key][c]
        print('    %-28s n=%-4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
