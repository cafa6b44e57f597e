#!/bin/bash
# per-kernel durations of the C4 inverse under the caller's environment: tools/trace_3d_inv.sh <outdir> <tag>
out=$GRAFT_REPO_ROOT/$1; tag=$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
REPS=30 rocprofv3 --kernel-trace --stats -d $out/$tag -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof3d_inv.py > $out/$tag.log 2>&1
python - "$out/$tag/t_kernel_trace.csv" "$tag" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']; n = n[n.find('k_'):][:44] if 'k_' in n else n[:44]
    d[(n, r['Grid_Size_X'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0
for k, v in sorted(d.items()):
    v.sort(); med = v[len(v) // 2]; tot += med if len(v) >= 20 else 0
    if len(v) >= 20: print('%-8s %-46s grid %9s n=%3d  median %7.1f us  min %7.1f' % (sys.argv[2], k[0], k[1], len(v), med / 1e3, v[0] / 1e3))
print('%-8s sum of medians %.1f us' % (sys.argv[2], tot / 1e3))
PY
