#!/bin/bash
# One transform at a time on the whole device, sizes 2048^2 .. 8192^2: an environment switch off / on, alternating, twice per size.
#   tools/sweep_inv_pair_sizes.sh [VAR=DTCWT_HIP_INV21_PAIR]       (k_inv21m against the same macro-steps as a marching pair)
# prints per run: ms_per_step, level-1+2 forward kernel ms, level-2+1 inverse kernel ms (event pairs, one at a time)
VAR=${1:-DTCWT_HIP_INV21_PAIR}
cd ${GRAFT_REPO_ROOT:-.}
for n in 2048 3072 4096 5120 6144 8192; do
  for rep in 1 2; do
  for arm in 0 1; do
    echo "n=$n $VAR=$arm: $(env $VAR=$arm python bench.py --no-cpu-baseline --no-other-configs --no-probe --streams 1 --rows $n --cols $n --steps 30 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(d["ms_per_step"], r["fwd_kernel_ms"][0], r["inv_kernel_ms"][1])')"
  done; done
done
