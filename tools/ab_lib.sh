#!/bin/bash
# A/B of two builds of the library under the bench protocol, alternating in one call so that both arms see the same box:
#   tools/ab_lib.sh <libA.so> <libB.so> [rounds] [extra bench args]      (paths relative to the repo root)
A=$1; B=$2; N=${3:-3}; shift 3 || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq $N); do
  for arm in "$A" "$B"; do
    echo "$arm: $(DTCWT_HIP_LIBRARY=$R/$arm python $R/bench.py --no-cpu-baseline --no-other-configs --no-probe "$@" 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("ms/step", d["ms_per_step"], "one-stream", d["one_stream_ms_per_step"], "fwd", r["fwd_kernel_ms"], "inv", r["inv_kernel_ms"], "recon", d["recon_max_abs_err"])')"
  done
done
