#!/bin/bash
# GPU box: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the float64 Transform2d (4096^2, nlevels = 4, near_sym_a / qshift_a): tools/prof_f64.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/${1:-gpurun_out/prof_f64}; mkdir -p $O
cat > /tmp/f64run.py <<PY
import os, sys, numpy as np
sys.path.insert(0, "$R")
from dtcwt_amd.hip import Context, Transform2d
ctx = Context(0)
X = ctx.to_device(np.random.RandomState(0).standard_normal((4096, 4096)))
t = Transform2d(ctx=ctx)
for _ in range(12):
    p = t.forward(X, nlevels=4); z = t.inverse(p, device_output=True)
ctx.device_sync()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o f --output-format csv -- python /tmp/f64run.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python /tmp/f64run.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python /tmp/f64run.py > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections
st = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(st)):
    print("%-90s calls %4s avg %8.1f us" % (r["Name"].replace("void ","").replace("(anonymous namespace)::","")[:88], r["Calls"], float(r["AverageNs"])/1e3))
for tag, d in (("FETCH_SIZE", "pf"), ("WRITE_SIZE", "pw")):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == tag:
            acc[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(tag, k, "n=%d mean %.4g KiB" % (len(v), sum(v) / len(v)))
PY
