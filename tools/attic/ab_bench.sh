#!/bin/bash
# A/B of library builds inside ONE gpurun call (run-to-run noise between calls is +-3 %):
# tools/ab_bench.sh <outdir> <rounds> <lib1.so> <lib2.so> ...   (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1; ROUNDS=$2; shift 2
mkdir -p "$OUT"
for i in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    DTCWT_HIP_LIBRARY=$R/$lib python $R/bench.py --steps 300 --no-cpu-baseline --streams 1 > "$OUT/${name}_$i.json" 2>> "$OUT/err.txt"
    python -c "
import json;d=json.load(open('$OUT/${name}_$i.json'));print('%-28s %d  %.5f' % ('$name', $i, d['ms_per_step']), d['roofline']['fwd_kernel_ms'], d['roofline']['inv_kernel_ms'])"
  done
done
