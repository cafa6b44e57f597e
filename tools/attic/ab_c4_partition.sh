#!/bin/bash
# bench.py --config c4 with four plain streams against four streams on quarters of the compute units, alternating (near_sym_a, then near_sym_b / qshift_b)
cd ${GRAFT_REPO_ROOT:-.}
for w in "" "--biort near_sym_b --qshift qshift_b"; do
for rep in 1 2 3; do
for arg in "--cu-partition off" "--cu-partition on"; do
  echo "c4 $w $arg: $(python bench.py --config c4 --no-cpu-baseline --steps 40 $w $arg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["fwd_ms_per_step"], d["inv_ms_per_step"], d["ms_per_step_one_stream"])')"
done; done; done
