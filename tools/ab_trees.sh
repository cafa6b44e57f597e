#!/bin/bash
# A/B of two TREES (this one against a copy of another commit under gpurun_ab/<name>/: its own bench.py, package and library) under the
# driver's bench command, alternating in one gpurun call so that both see the same box:
#   tools/ab_trees.sh <name under gpurun_ab> [rounds] [extra bench args]
N=$1; R=${2:-3}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq $R); do
  for arm in $N HEAD; do
    B=$ROOT/bench.py; [ $arm != HEAD ] && B=$ROOT/gpurun_ab/$arm/bench.py
    echo "$arm: $(python $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-clocks "$@" 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; f=r.get("in_flight") or {}; p=(d.get("streaming_probe") or {}).get("transform_over_probe")
print("ms/step", d["ms_per_step"], "sustained", d.get("sustained_ms_per_step"), "one", d["one_stream_ms_per_step"], "| alone fwd", r["fwd_kernel_ms"], "inv", r["inv_kernel_ms"], "| in flight fwd", f.get("fwd_kernel_ms"), "inv", f.get("inv_kernel_ms"), "| over probe", p)')"
  done
done
