#!/bin/bash
# round-4 profile collection (GPU box)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
bash $R/tools/profile_round.sh gpurun_out/r04prof r04 > $O/profile_round.log 2>&1
cd $R/tools/kbench && timeout 300 ./march_bench_noslp 4096 60 > $O/march_bench.txt 2>&1
timeout 60 ./vmcnt_probe > $O/vmcnt_probe.txt 2>&1
cd $R
for c in c3 c5 c4; do python bench.py --config $c --no-cpu-baseline --steps 40 > $O/bench_$c.json 2>/dev/null; done
DTCWT_HIP_MARCH=0 python bench.py --no-cpu-baseline --no-other-configs --steps 100 > $O/bench_march_off.json 2>/dev/null
DTCWT_HIP_MARCH_INV=1 python bench.py --no-cpu-baseline --no-other-configs --steps 100 > $O/bench_inv_march_forced.json 2>/dev/null
cat $O/status.txt
