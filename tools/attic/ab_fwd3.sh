#!/bin/bash
# tools/bench_fwd3_l1.py with each of the given library builds: tools/ab_fwd3.sh "<knocks>" lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
K=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    echo "== $(basename $lib .so) ($i)"
    DTCWT_HIP_LIBRARY=$R/$lib KNOCKS=$K python $R/tools/bench_fwd3_l1.py
  done
done
