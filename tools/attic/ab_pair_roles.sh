#!/bin/bash
# EXPERIMENT OF RECORD (profiles/r05/pair_roles.txt): the switches DTCWT_HIP_PAIR_ROLES / DTCWT_HIP_PAIR_B it drives lived in
# march2d_pair.hpp / march2d.hip for one measurement and are gone again (no difference); the probe half still runs.
# Which wavefront of a marching pair takes level 1: the placement probe first, then
# bench.py lines per role mode, alternating in one call.  Columns: ms_per_step, one_stream_ms_per_step, fwd kernel ms per level.
cd ${GRAFT_REPO_ROOT:-.}
tools/kbench/hwid_probe
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["ms_per_step"], d["one_stream_ms_per_step"], r.get("fwd_kernel_ms"), d["recon_max_abs_err"])'; }
for rep in 1 2; do
  for roles in 0 1 2 3 4; do
    echo "near_sym_b/qshift_b PAIR_B=1 roles=$roles: $(DTCWT_HIP_PAIR_B=1 DTCWT_HIP_PAIR_ROLES=$roles python bench.py --no-cpu-baseline --no-other-configs --biort near_sym_b --qshift qshift_b --steps 40 2>/dev/null | line)"
  done
  echo "near_sym_b/qshift_b no pair: $(python bench.py --no-cpu-baseline --no-other-configs --biort near_sym_b --qshift qshift_b --steps 40 2>/dev/null | line)"
  for roles in 0 1 2 3 4; do
    echo "near_sym_a/qshift_b roles=$roles: $(DTCWT_HIP_PAIR_ROLES=$roles python bench.py --no-cpu-baseline --no-other-configs --qshift qshift_b --steps 40 2>/dev/null | line)"
  done
done
