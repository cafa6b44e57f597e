#!/bin/bash
# level-1 kernel alone + whole C4 forward / inverse with each library build: tools/ab_3d.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2; do
  for lib in "$@"; do
    echo "== $(basename $lib .so) ($i)"
    DTCWT_HIP_LIBRARY=$R/$lib python $R/tools/bench_fwd3_l1.py
    DTCWT_HIP_LIBRARY=$R/$lib REPS=30 python $R/tools/prof3d_fwd.py
    DTCWT_HIP_LIBRARY=$R/$lib REPS=30 python $R/tools/prof3d_inv.py 2>&1 | tail -2
  done
done
