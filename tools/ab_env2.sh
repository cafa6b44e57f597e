#!/bin/bash
# A/B of environment settings under the driver-protocol bench, alternating in one call so that every arm sees the same box:
#   tools/ab_env2.sh [rounds] "<VAR=value ...>" "<VAR=value ...>" ...        (an arm "-" = no setting)
# prints per run: ms per step (20-step window), sustained, one transform at a time, and the in-flight kernel times per level
N=$1; shift
for i in $(seq $N); do
  for arm in "$@"; do
    a=$arm; [ "$arm" = "-" ] && a="_AB_NONE=1"
    echo "$arm: $(env $a python bench.py --no-cpu-baseline --no-other-configs --no-probe --no-clocks $AB_BENCH_ARGS 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; f=r.get("in_flight") or {}
print("ms/step", d["ms_per_step"], "sustained", d.get("sustained_ms_per_step"), "one", d["one_stream_ms_per_step"], "| alone fwd", r["fwd_kernel_ms"], "inv", r["inv_kernel_ms"], "| in flight fwd", f.get("fwd_kernel_ms"), "inv", f.get("inv_kernel_ms"))')"
  done
done
