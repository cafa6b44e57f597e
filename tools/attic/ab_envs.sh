#!/bin/bash
# A/B/C... of environment settings under bench.py, alternating in one call so that every arm sees the same box:
#   tools/ab_envs.sh <rounds> "<bench args>" "<VAR=v for arm 1>" "<VAR=v for arm 2>" ...
# prints per run: ms_per_step sustained one_stream [fwd kernel ms] [inv kernel ms] (in flight: fwd12 / inv21 kernel ms per image-share)
N=$1; ARGS=$2; shift 2
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq $N); do
  for arm in "$@"; do
    echo "$arm: $(env $arm python bench.py --no-cpu-baseline --no-other-configs --no-probe $ARGS 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; f=r.get("in_flight") or {}
print(d["ms_per_step"], d.get("sustained_ms_per_step"), d.get("one_stream_ms_per_step"), r.get("fwd_kernel_ms", [None])[0], (r.get("inv_kernel_ms") or [None, None])[1], "in flight", (f.get("fwd_kernel_ms") or [None])[0], (f.get("inv_kernel_ms") or [None, None])[1], "recon", d["recon_max_abs_err"])')"
  done
done
