#!/bin/bash
# tools/measure_configs.py with each of the given library builds, inside one gpurun call
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/$1; shift
mkdir -p "$OUT"
for i in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    echo "== $name ($i)"
    DTCWT_HIP_LIBRARY=$R/$lib python $R/tools/measure_configs.py 2>/dev/null | tee "$OUT/${name}_$i.txt" | cut -c1-175
  done
done
